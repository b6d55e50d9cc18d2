#!/bin/bash
# tools/build_variant.sh NAME "file.hip [file2.hip ...]" "-DMACRO=VALUE ..."  ->  tools/ab/libNAME.so
# (the named sources compiled with the extra flags, everything else taken from the in-tree objects: build the default library first)
set -e
cd "$(dirname "$0")/.."
NAME=$1; FILES=$2; FLAGS=$3
PKG=segment-anything-in-nerf_amd
python -c "import sys; sys.path.insert(0,'.'); import samnerf_amd; from samnerf_amd import _lib; _lib.build()"
mkdir -p tools/ab /tmp/snf_variant_$NAME
OBJS=""
for o in $PKG/lib/obj/*.o; do
  b=$(basename $o .o); skip=0
  for f in $FILES; do [ "$b.hip" = "$f" ] && skip=1; done
  [ $skip = 0 ] && OBJS="$OBJS $o"
done
for f in $FILES; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics $FLAGS -w -I include -c $PKG/csrc/$f -o /tmp/snf_variant_$NAME/${f%.hip}.o &
done
wait
for f in $FILES; do OBJS="$OBJS /tmp/snf_variant_$NAME/${f%.hip}.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o tools/ab/lib$NAME.so
echo "tools/ab/lib$NAME.so  ($FILES: $FLAGS)"

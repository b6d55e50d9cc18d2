#!/bin/bash
# same-box A/B of a compile-time switch:  tools/ab_define.sh <source.hip> "<-DNAME=VALUE ...>" <command ...>
# runs <command> with the stock library, rebuilds <source.hip> with the defines and relinks (on the GPU box's scratch copy), runs it again.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
src="$1"; defs="$2"; shift 2
PKG=segment-anything-in-nerf_amd
python -c "import sys; sys.path.insert(0,'.'); import samnerf_amd; from samnerf_amd import _lib; _lib.build()"
echo "== stock"; "$@"
obj=$PKG/lib/obj/${src%.hip}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics $defs -I include -c $PKG/csrc/$src -o $obj || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $PKG/lib/obj/*.o -o $PKG/lib/libsamnerf_hip.so || exit 1
echo "== $defs"; "$@"

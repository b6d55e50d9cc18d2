#!/bin/bash
# Ablation of k_gemm_planes (csrc/gemm_planes.hip, GP_ABL): tools/ab/libgpabl<N>.so built beforehand with
#   for n in 0 1 2 4 8 7 15 ...; do tools/build_variant.sh gpabl$n gemm_planes.hip -DGP_ABL=$n; done
# each copied over the in-tree library in turn, tools/bench_gemm_planes.py on the forced shapes.  usage: ablate_gp.sh "gpabl0 gpabl1 gpabl7 <any other variant name>"
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
LIB=segment-anything-in-nerf_amd/lib/libsamnerf_hip.so
cp $LIB /tmp/lib_keep.so
for n in ${1:-gpabl0 gpabl1 gpabl2 gpabl4 gpabl8 gpabl7}; do
  cp tools/ab/lib$n.so $LIB
  echo "-- $n"
  GP_SHAPES=${GP_SHAPES:-2x2,1x4,2x4} GP_GEMMS=${GP_GEMMS:-lin1,lin2} python tools/bench_gemm_planes.py 2>/dev/null
done
cp /tmp/lib_keep.so $LIB

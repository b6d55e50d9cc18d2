#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s23; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu -k "grid_head or fused or render or camera or head_path" 2>&1 | tail -4 > $O/out.txt
LIB=segment-anything-in-nerf_amd/lib/libsamnerf_hip.so; cp $LIB /tmp/lib_keep.so
for r in 1 2 3; do for v in A B; do cp tools/ab/lib$v.so $LIB; echo "render $v $(python tools/bench_render.py 2>/dev/null | grep '^render' | cut -c1-60)" >> $O/out.txt; done; done
cp /tmp/lib_keep.so $LIB
cat $O/out.txt

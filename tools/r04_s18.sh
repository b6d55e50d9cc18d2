#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s18; mkdir -p $O; : > $O/out.txt
LIB=segment-anything-in-nerf_amd/lib/libsamnerf_hip.so
cp $LIB /tmp/lib_keep.so
for r in 1 2; do for v in ${VARIANTS:-A B}; do cp tools/ab/lib$v.so $LIB; echo "== $v" >> $O/out.txt; python tools/bench_attention.py 2>/dev/null | tail -5 >> $O/out.txt; python tools/bench_vit.py 2>/dev/null | head -8 >> $O/out.txt; done; done
cp /tmp/lib_keep.so $LIB
timeout 900 python -m pytest tests -x -q -m gpu -k "attention or vit or encoder" 2>&1 | tail -5 >> $O/out.txt
cat $O/out.txt

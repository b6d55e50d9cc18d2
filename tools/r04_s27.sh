#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s27; mkdir -p $O; : > $O/out.txt
for i in 1 2; do
echo "mode1 $(python tools/bench_render.py 2>/dev/null | grep '^render' | cut -c1-60)" >> $O/out.txt
echo "mode2 $(SNF_GEMM_MODE=2 python tools/bench_render.py 2>&1 | grep -E '^render|Error' | cut -c1-100)" >> $O/out.txt
done
cat $O/out.txt

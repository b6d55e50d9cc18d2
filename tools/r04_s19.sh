#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s19; mkdir -p $O
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_step_program_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "mlp64 or chain or colour or ministep or one_step or trajectory or fused or render or camera" 2>&1 | tail -4 > $O/tests.txt
VARIANTS="A B" ROUNDS=3 STEPS=60 KEYS="mlp64" bash tools/ab_bench.sh > $O/ab.txt 2>&1
LIB=segment-anything-in-nerf_amd/lib/libsamnerf_hip.so; cp $LIB /tmp/lib_keep.so
for r in 1 2; do for v in A B; do cp tools/ab/lib$v.so $LIB; echo "render $v $(python tools/bench_render.py 2>/dev/null | grep '^render' | cut -c1-60)" >> $O/ab.txt; done; done
cp /tmp/lib_keep.so $LIB
cat $O/tests.txt $O/ab.txt

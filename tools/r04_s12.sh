#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s12; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_step_program_gpu.py tests/test_fullsize_gpu.py -x -q -k "weight_gradient or wgrad or rows or ministep or trajectory or composed or schedule or rendered" 2>&1 | tail -4 > $O/tests.txt
KEYS="${KEYS:-bwd_weight}" ROUNDS=2 STEPS=40 timeout 900 bash tools/ab_bench.sh > $O/ab.txt 2>&1
cat $O/tests.txt $O/ab.txt

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s4; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py tests/test_step_program_gpu.py -x -q 2>&1 | tail -8 > $O/tests.txt
KEYS="presorted_adam" ROUNDS=2 STEPS=40 timeout 900 bash tools/ab_bench.sh > $O/ab_permute.txt 2>&1
cat $O/tests.txt $O/ab_permute.txt

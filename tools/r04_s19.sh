#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s19; mkdir -p $O
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_step_program_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "mlp64 or chain or colour or ministep or one_step or trajectory or fused or hashgrid or adam or full_table" 2>&1 | tail -5 > $O/tests.txt
VARIANTS="A B C" ROUNDS=3 STEPS=60 KEYS="mlp64 adam_xp" bash tools/ab_bench.sh > $O/ab.txt 2>&1
cat $O/tests.txt $O/ab.txt

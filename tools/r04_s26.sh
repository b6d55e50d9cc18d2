#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s26; mkdir -p $O
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py tests/test_step_program_gpu.py -x -q -m gpu -k "hashgrid or sort or presorted or full_table or one_step or ministep or pair" 2>&1 | tail -4 > $O/out.txt
VARIANTS="A B" ROUNDS=3 STEPS=60 KEYS="sort_xp adam_xp" bash tools/ab_bench.sh >> $O/out.txt 2>&1
cat $O/out.txt

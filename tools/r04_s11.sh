#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s11; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_step_program_gpu.py tests/test_fullsize_gpu.py -x -q -k "mlp64 or colour_net or ministep or trajectory or composed or schedule or nerfacto" 2>&1 | tail -12 > $O/tests.txt
KEYS="mlp64" ROUNDS=2 STEPS=40 timeout 900 bash tools/ab_bench.sh > $O/ab_dg3.txt 2>&1
cat $O/tests.txt $O/ab_dg3.txt

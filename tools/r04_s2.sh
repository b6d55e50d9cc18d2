#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s2; mkdir -p $O
timeout 300 python -m pytest tests/test_step_program_gpu.py -x -q 2>&1 | tail -3 > $O/tests.txt
SNF_SAM_WGRAD_ON_CLIPSEG=1 timeout 300 python -m pytest tests/test_step_program_gpu.py -x -q 2>&1 | tail -3 >> $O/tests.txt
timeout 900 bash tools/ab_env.sh SNF_SAM_WGRAD_ON_CLIPSEG=1 > $O/ab_bal.txt 2>&1
cat $O/tests.txt $O/ab_bal.txt

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s3; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "dense_coarse or x_pair" 2>&1 | tail -25 > $O/tests.txt
timeout 900 python -m pytest tests/test_step_program_gpu.py tests/test_fullsize_gpu.py -x -q 2>&1 | tail -5 >> $O/tests.txt
timeout 900 bash tools/ab_env.sh SNF_HG_DENSE_COARSE=0 snf_hashgrid_sort_xp/L16 snf_hashgrid_sort_xp/L11 snf_hashgrid_bwd_presorted_adam_xp/F2L16 snf_hashgrid_bwd_presorted_adam_xp/F2L11tp snf_hashgrid_bwd_dense/F2L5 snf_adam_step_rows > $O/ab_dense.txt 2>&1
cat $O/tests.txt $O/ab_dense.txt

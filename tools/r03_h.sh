#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$(pwd)
out=$ROOT/gpurun_out/${1:-r03h}; mkdir -p $out
python -m pytest tests/test_ops_gpu.py -q -x -k "reachable or pair or fused_adam or both_grids" 2>&1 | tail -2
SP=1 CASES=f8a POS=real REPS=20 python tools/microbench_hgadam.py 2>/dev/null | grep "^f8a" | cut -c1-200
tools/ab_env.sh SNF_HG_SPARSE_LEVELS=0 snf_hashgrid_bwd_presorted_adam_pair/F8L12+12 snf_adam_step_rows 2>&1 | cut -c1-300 | tee $out/ab.txt
SNF_PARITY_VERBOSE=1 python -m pytest tests -m gpu -x -q -s > $out/tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/tests.log | cut -c1-300
grep "grad_parity" $out/tests.log | cut -c1-2500 > $out/parity_reports.txt

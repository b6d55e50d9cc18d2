#!/bin/bash
# round 3: reachable-row levels over compact rows -- tests, microbenchmarks (alone), step A/B on one box
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out/${1:-r03b}; mkdir -p $out
python -m pytest tests/test_ops_gpu.py -q -x -k "reachable or pair or fused_adam or fixed_point or both_grids or record_limit or hashgrid" > $out/tests_hg.log 2>&1; echo "hg tests rc=$?"; tail -3 $out/tests_hg.log
for c in f8a f2 f2p; do
  ROWS_ADAM=1 CASES=$c POS=real python tools/microbench_hgadam.py 2>/dev/null | grep "^$c" | sed 's/^/rows_adam /'
  SP=1 CASES=$c POS=real python tools/microbench_hgadam.py 2>/dev/null | grep "^$c" | sed 's/^/sparse    /'
done | tee $out/microbench.txt
tools/ab_env.sh SNF_HG_SPARSE_LEVELS=0 snf_hashgrid_bwd_presorted_adam_pair/F8L12+12 snf_hashgrid_bwd_presorted_adam_sp/F2L16 snf_hashgrid_bwd_presorted_adam/F2L16 snf_hashgrid_bwd_presorted_adam_sp/F2L5 snf_adam_step_rows 2>&1 | tee $out/ab.txt

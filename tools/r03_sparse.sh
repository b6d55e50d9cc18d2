#!/bin/bash
# round 3: reachable-row levels over compact rows -- tests, microbenchmarks (alone), step A/B on one box; render profile
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$(pwd)
out=$ROOT/gpurun_out/${1:-r03b}; mkdir -p $out
python -m pytest tests/test_ops_gpu.py -q -x -k "reachable or pair or fused_adam or fixed_point or both_grids or record_limit or hashgrid" > $out/tests_hg.log 2>&1; echo "hg tests rc=$?"; tail -3 $out/tests_hg.log
for c in f8a; do
  ROWS_ADAM=1 CASES=$c POS=real python tools/microbench_hgadam.py 2>/dev/null | grep "^$c\|reachable" | sed 's/^/rows_adam /'
  SP=1 CASES=$c POS=real python tools/microbench_hgadam.py 2>/dev/null | grep "^$c\|reachable" | sed 's/^/sparse    /'
done | tee $out/microbench.txt
tools/ab_env.sh SNF_HG_SPARSE_LEVELS=0 snf_hashgrid_bwd_presorted_adam_pair/F8L12+12 snf_adam_step_rows 2>&1 | tee $out/ab.txt
python -m pytest tests/test_step_program_gpu.py tests/test_fullsize_gpu.py -q -x > $out/tests_sp.log 2>&1; echo "schedule tests rc=$?"; tail -3 $out/tests_sp.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/render_stats -o render -- python $ROOT/tools/bench_render.py > $out/render.txt 2> $out/render.err
tail -2 $out/render.txt
f=$(find $out/render_stats -name "*kernel_stats.csv" | head -1); head -40 $f

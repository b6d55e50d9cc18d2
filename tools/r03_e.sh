#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$(pwd)
out=$ROOT/gpurun_out/${1:-r03e}; mkdir -p $out
python tools/debug_fullsize.py 2>/dev/null | cut -c1-400 | tee $out/debug_fullsize.txt
python -m pytest tests/test_ops_gpu.py -q -x -k "reachable or pair" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for frac in 0.4 0.3; do
  SNF_SPARSE_MAX_FRACTION=$frac SP=1 CASES=f8a POS=real REPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $out/mb_SP$frac -o mb -- python $ROOT/tools/microbench_hgadam.py > $out/mb_SP$frac.txt 2>&1
  echo "== SP frac $frac"; grep "^f8a\|reachable" $out/mb_SP$frac.txt | cut -c1-200; python $ROOT/tools/kstats.py $out/mb_SP$frac 12 | grep "k_hg_reduce\|absmax\|k_adam"
done
cd $ROOT
for frac in 0.4 0.3; do
SNF_SPARSE_MAX_FRACTION=$frac tools/ab_env.sh SNF_HG_SPARSE_LEVELS=0 snf_hashgrid_bwd_presorted_adam_pair/F8L12+12 snf_adam_step_rows 2>&1 | cut -c1-300 | tee $out/ab$frac.txt
done
python -m pytest tests/test_model_gpu.py -q -x -k "render or eval" 2>&1 | tail -3 | cut -c1-300
python tools/bench_render.py 2>/dev/null | tail -1 | cut -c1-200
SNF_STATIC_RENDER=0 python tools/bench_render.py 2>/dev/null | tail -1 | cut -c1-200
RES=1024 python tools/bench_render.py 2>/dev/null | tail -1 | cut -c1-200

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$(pwd)
out=$ROOT/gpurun_out/${1:-r03d}; mkdir -p $out
SNF_PARITY_VERBOSE=1 python -m pytest tests/test_step_program_gpu.py tests/test_fullsize_gpu.py tests/test_model_gpu.py -q -x -s > $out/tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/tests.log | cut -c1-300
grep "grad_parity" $out/tests.log | cut -c1-1800 > $out/parity_reports.txt
python tools/bench_render.py 2>/dev/null | tail -1 | cut -c1-200
cd /tmp && export TMPDIR=/tmp
for mode in SP ROWS_ADAM; do
  env $mode=1 CASES=f8a POS=real REPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $out/mb_$mode -o mb -- python $ROOT/tools/microbench_hgadam.py > $out/mb_$mode.txt 2>&1
  echo "== $mode"; grep "^f8a" $out/mb_$mode.txt | cut -c1-200; python $ROOT/tools/kstats.py $out/mb_$mode 12 | grep -v "at::native\|rocclr"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/render_stats -o render -- python $ROOT/tools/bench_render.py > $out/render.txt 2> $out/render.err
python $ROOT/tools/kstats.py $out/render_stats 14

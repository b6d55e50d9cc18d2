#!/bin/bash
# HBM bytes of every kernel per STEADY-STATE train step: two PMC passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only) at two step
# counts; the per-kernel difference / (steps_b - steps_a) cancels table initialisation, first-step schedule building and
# everything else that is not a train step.      usage on the GPU box: bash tools/pmc_steady.sh <tag> [workload]
set -u
TAG=${1:-steady}; WL=${2:-distill_4096x128}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for S in 24 64; do for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/steady_${S}_$C -o pmc -- python $ROOT/tools/steady_steps.py $S $WL > $OUT/steady_${S}_$C.log 2>&1
  rm -f $OUT/steady_${S}_$C/pmc_kernel_trace.csv
done; done
python $ROOT/tools/pmc_steady_summary.py $OUT 24 64 > $OUT/hbm_bytes_per_kernel_steady.txt
find $OUT -size +8M -delete
head -40 $OUT/hbm_bytes_per_kernel_steady.txt

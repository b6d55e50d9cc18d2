#!/bin/bash
# HBM bytes of EVERY kernel of the step: two PMC passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only) over a short bench run.
# usage on the GPU box: bash tools/pmc_all.sh <tag>     ->  gpurun_out/<tag>/pmcall_{FETCH,WRITE}_SIZE/
set -u
TAG=${1:-pmcall}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmcall_$C -o pmc -- \
      python $ROOT/bench.py --steps 6 --warmup 3 --cpu-baseline-seconds 0 --other-workloads none --steady-steps 0 > $OUT/pmcall_$C.json 2> $OUT/pmcall_$C.err
  rm -f $OUT/pmcall_$C/pmc_kernel_trace.csv
done
ls -la $OUT/pmcall_*/

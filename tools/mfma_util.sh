#!/bin/bash
# Matrix-core utilisation of the step's GEMM-shaped kernels from hardware counters (north_star: "MFMA utilisation against
# gfx950 peak"): one rocprofv3 PMC pass (kernel-trace only) over a short bench run.
# usage on the GPU box:  bash tools/mfma_util.sh <tag>   ->  gpurun_out/<tag>/mfma/ , then  python tools/mfma_util.py gpurun_out/<tag>/mfma
set -u
TAG=${1:-mfma}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG/mfma
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES \
    --kernel-trace --output-format csv -d $OUT -o pmc -- \
    python $ROOT/bench.py --steps 6 --warmup 3 --cpu-baseline-seconds 0 --other-workloads none > $OUT/bench.json 2> $OUT/bench.err
ls -la $OUT

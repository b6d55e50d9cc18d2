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
PMC="SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES"
rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/step -o pmc -- \
    python $ROOT/bench.py --steps 6 --warmup 3 --cpu-baseline-seconds 0 --other-workloads none --steady-steps 0 > $OUT/bench.json 2> $OUT/bench.err
# config #5: the render pass and the encoder (their matrix kernels are not part of the train step)
rocprofv3 --pmc $PMC --kernel-trace --kernel-include-regex "k_grid_head_fused|k_gemm|k_mlp_chain" --output-format csv -d $OUT/render -o pmc -- \
    python $ROOT/tools/bench_render.py > /dev/null 2> $OUT/render.err
rocprofv3 --pmc $PMC --kernel-trace --kernel-include-regex "k_gemm_planes|k_attention|k_gemm_rows" --output-format csv -d $OUT/vit -o pmc -- \
    python $ROOT/tools/bench_vit.py > /dev/null 2> $OUT/vit.err
cd $ROOT && python tools/mfma_util.py $ROOT/gpurun_out/$TAG/mfma_util.json $OUT/step $OUT/render $OUT/vit > $ROOT/gpurun_out/$TAG/mfma_util.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete
ls -la $OUT

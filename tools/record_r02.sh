#!/bin/bash
# Round-2 evidence on one GPU box: bench line, rocprofv3 kernel stats, PMC traffic of the dominant kernel, HBM bytes of every
# kernel, matrix-core utilisation, timelines, host enqueue time, image-encoder and render timings.
# usage (GPU box, repo root): bash tools/record_r02.sh <tag>
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
# matrix-core counters first: bench.py quotes them (profiles/r02_mfma_util.json -> roofline_other_kernels[*].mfma_busy)
bash tools/mfma_util.sh $TAG > /dev/null 2>&1
python tools/mfma_util.py $OUT/mfma $OUT/mfma_util.json > $OUT/mfma_util.txt 2>&1 && cp $OUT/mfma_util.json profiles/r02_mfma_util.json
bash tools/gpu_record.sh $TAG > $OUT/gpu_record.log 2>&1
python tools/rocpd_stats.py $(ls $OUT/stats/*/*_results.db 2>/dev/null | head -1) > $OUT/kernel_stats_from_db.csv 2>/dev/null
bash tools/pmc_all.sh $TAG > /dev/null 2>&1
python tools/pmc_all_summary.py $OUT > $OUT/hbm_bytes_per_kernel.txt 2>&1
python tools/eager_timeline.py > $OUT/timeline_distill.txt 2>&1
WORKLOAD=no_distill_4096x128 python tools/eager_timeline.py > $OUT/timeline_no_distill.txt 2>&1
python tools/host_vs_gpu.py > $OUT/host_vs_gpu.txt 2>&1
python tools/bench_vit.py > $OUT/vit.txt 2>&1
python tools/bench_render.py > $OUT/render_512.txt 2>&1
RES=1024 python tools/bench_render.py > $OUT/render_1024.txt 2>&1
rm -rf $OUT/pmcall_*/pmc_kernel_trace.csv $OUT/mfma/pmc_kernel_trace.csv
find $OUT -size +8M -delete
ls -la $OUT | head -40
tail -c 1200 $OUT/bench.json

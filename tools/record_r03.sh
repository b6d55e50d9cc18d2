#!/bin/bash
# Round-3 evidence on ONE GPU box: PMC traffic of the dominant kernel, default bench line, rocprofv3 kernel stats of the same
# command, HBM bytes of every kernel, step timeline, per-kernel ablation of the concurrent step, render-path kernel stats,
# ViT-H timing, cache / TA counters of the F = 2 grid kernels.      usage (GPU box, repo root): bash tools/record_r03.sh <tag>
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
bash tools/gpu_record.sh $TAG > $OUT/gpu_record.log 2>&1
bash tools/pmc_all.sh $TAG > /dev/null 2>&1
python tools/pmc_all_summary.py $OUT > $OUT/hbm_bytes_per_kernel.txt 2>&1
python tools/eager_timeline.py 2>/dev/null | cut -c1-200 > $OUT/timeline_distill.txt
python tools/host_vs_gpu.py > $OUT/host_vs_gpu.txt 2>&1
tools/ablate_step.sh $OUT/ablation.txt snf_hashgrid_bwd_presorted_adam_pair snf_hashgrid_bwd_presorted_adam_xp/F2L16 snf_hashgrid_sort_xp/L16 snf_mlp64_bwd_fused/31x64x64x3 snf_mlp64_bwd_fused/32x64x16 snf_mlp64_fwd snf_hashgrid_fwd/F2L16 snf_hashgrid_fwd/F8L12 snf_linear_bwd_weight_rows snf_linear_bwd_data_rows snf_linear_fwd_mean 2304x256 snf_adam_step_rows snf_linear_ snf_mlp64_ > /dev/null 2>&1
python tools/bench_vit.py 2>/dev/null | cut -c1-200 > $OUT/vit.txt
( python tools/bench_render.py; RES=1024 python tools/bench_render.py ) 2>/dev/null | grep "^render" | cut -c1-400 > $OUT/render.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/render_stats -o render -- python $ROOT/tools/bench_render.py > /dev/null 2>&1
cd $ROOT
python tools/kstats.py $OUT/render_stats 24 > $OUT/render_kernel_stats.txt 2>&1
python tools/kstats.py $OUT/stats 40 > $OUT/kernel_stats_short.txt 2>&1
rm -rf $OUT/pmcall_*/pmc_kernel_trace.csv $OUT/render_stats/*kernel_trace.csv $OUT/stats/*kernel_trace.csv $OUT/stats/*/*kernel_trace.csv
find $OUT -size +6M -delete
ls $OUT | head -40
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print({k:d[k] for k in ("ms_per_step","value","step_frac_of_hbm_peak")}, "roofline", d["roofline"]["frac"], "serial", d["roofline"]["serial"]["frac"], "traffic", d["roofline"]["traffic"])
print(json.dumps({k:(v.get("ms_per_step") or v.get("ms_per_image")) for k,v in d["other_workloads"].items()}))
PY
cat $OUT/ablation.txt

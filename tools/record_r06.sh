#!/bin/bash
# Round-6 evidence on ONE GPU box (same set as record_r04.sh; the per-kernel HBM bytes by the steady-state difference of two step counts):
# matrix-core busy fractions, PMC traffic of the dominant kernel and of the field grid's reduce, the default bench line, rocprofv3 kernel
# stats of the same command, HBM bytes per kernel and steady-state step, step timeline per stream, per-kernel ablation, wave-stall counters,
# render / encoder timings.      usage (GPU box, repo root): bash tools/record_r06.sh <tag>
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
bash tools/mfma_util.sh $TAG > $OUT/mfma_util.log 2>&1
[ -s $OUT/mfma_util.json ] && cp $OUT/mfma_util.json profiles/r06_mfma_util.json
bash tools/gpu_record.sh $TAG > $OUT/gpu_record.log 2>&1
python tools/pmc_traffic.py $OUT --regex "k_hg_reduce_fx<2, ?true, ?1, ?true>" --key snf_hashgrid_bwd_presorted_adam_xp/F2L16 \
    --out $OUT/pmc_traffic_fx.json > $OUT/pmc_traffic_fx.log 2>&1 && cp $OUT/pmc_traffic_fx.json profiles/pmc_traffic_fx.json
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 --other-workloads none > $OUT/bench_driver_args.json 2>/dev/null
bash tools/pmc_steady.sh $TAG > /dev/null 2>&1
PER_STREAM=1 python tools/eager_timeline.py 2>/dev/null | cut -c1-200 > $OUT/timeline_distill.txt
python tools/host_vs_gpu.py > $OUT/host_vs_gpu.txt 2>&1
tools/ablate_step.sh $OUT/ablation.txt snf_hashgrid_bwd_presorted_adam_pair "snf_hashgrid_bwd_presorted_adam_pair,snf_hashgrid_sort/L12" snf_hashgrid_bwd_presorted_adam_xp/F2L16 "snf_hashgrid_bwd_presorted_adam_xp,snf_hashgrid_sort_xp" snf_mlp64_bwd_fused_sh/31x64x64x3 snf_mlp64_bwd_fused/32x64x16 snf_mlp64_fwd snf_hashgrid_fwd/F2L16 snf_hashgrid_fwd/F8L12 snf_linear_bwd_weight_rows snf_linear_bwd_data_rows snf_linear_fwd_mean 2304x256 snf_patch "256x256r,256x192r" "snf_composite,snf_rowmse,snf_trunc_exp,snf_weights,snf_distortion,snf_interlevel,snf_add_scaled,snf_nerf_loss_summary" "snf_mlp_tiny,snf_hashgrid_fwd/F2L5,snf_prop_density,snf_hashgrid_bwd_presorted_adam/F2L5,snf_hashgrid_sort/L5" snf_adam_step snf_guard snf_linear_ snf_mlp64_ snf_hashgrid_fwd snf_hashgrid_ "snf_linear_,snf_mlp64_,snf_patch" > /dev/null 2>&1
bash tools/step_counters.sh $TAG/counters > /dev/null 2>&1
python tools/bench_vit.py 2>/dev/null | cut -c1-200 > $OUT/vit.txt
( python tools/bench_render.py; SNF_RENDER_REUSE_PASS1=0 python tools/bench_render.py; RES=1024 python tools/bench_render.py ) 2>/dev/null | grep "^render" | cut -c1-400 > $OUT/render.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/render_stats -o render -- python $ROOT/tools/bench_render.py > /dev/null 2>&1
cd $ROOT
python tools/kstats.py $OUT/render_stats 24 > $OUT/render_kernel_stats.txt 2>&1
python tools/kstats.py $OUT/stats 40 > $OUT/kernel_stats_short.txt 2>&1
rm -rf $OUT/steady_*/pmc_kernel_trace.csv $OUT/render_stats/*kernel_trace.csv $OUT/stats/*kernel_trace.csv $OUT/stats/*/*kernel_trace.csv
find $OUT -size +6M -delete
ls $OUT | head -60
python tools/benchsum.py $OUT/bench.json
cat $OUT/ablation.txt $OUT/render.txt $OUT/vit.txt

#!/bin/bash
# Copies the judged subset of a round-6 recording (gpurun_out/<tag>/, made by tools/record_r06.sh) into profiles/ as r06_*.
# usage (build container, repo root): bash tools/install_r06.sh <tag>
set -eu
TAG=$1; G=gpurun_out/$TAG; P=profiles
[ -s $G/bench.json ] || { echo "no $G/bench.json"; exit 1; }
cp $G/bench.json $P/r06_bench.json
cp $G/bench_driver_args.json $P/r06_bench_driver_args.json
cp $G/bench_under_rocprof.json $P/r06_bench_under_rocprof.json
cp $(ls $G/stats/*/*kernel_stats.csv $G/stats/*kernel_stats.csv 2>/dev/null | head -1) $P/r06_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  cp $(ls $G/pmc_$C/*/*counter_collection.csv $G/pmc_$C/*counter_collection.csv 2>/dev/null | head -1) $P/r06_pmc_hg_reduce_$C.csv
done
cp $G/pmc_traffic.json $P/pmc_traffic.json
cp $G/pmc_traffic_fx.json $P/pmc_traffic_fx.json
cp $G/mfma_util.json $P/r06_mfma_util.json
cp $G/mfma_util.txt $P/r06_mfma_util.txt
cp $G/hbm_bytes_per_kernel_steady.txt $P/r06_hbm_bytes_per_kernel_steady.txt
cp $G/host_vs_gpu.txt $P/r06_host_vs_gpu.txt
cp $G/timeline_distill.txt $P/r06_timeline_distill.txt
cp $G/ablation.txt $P/r06_ablation.txt
cp $G/render.txt $P/r06_render.txt
cp $G/render_kernel_stats.txt $P/r06_render_kernel_stats.txt
cp $G/vit.txt $P/r06_vit.txt
[ -s $G/counters/step_counters.txt ] && cp $G/counters/step_counters.txt $P/r06_step_counters.txt || true
ls -la $P/r06_* $P/pmc_traffic*.json

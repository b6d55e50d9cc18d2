#!/bin/bash
# Copies the judged subset of a recording (gpurun_out/<tag>/, made by tools/record_r02.sh + tools/pmc_steady.sh) into profiles/
# and drops the previous recording's files.     usage (build container, repo root): bash tools/install_profiles.sh <tag> [old-tag]
set -eu
TAG=$1; OLD=${2:-}
G=gpurun_out/$TAG
P=profiles
[ -s $G/bench.json ] || { echo "no $G/bench.json"; exit 1; }
[ -n "$OLD" ] && git rm -q --ignore-unmatch $P/${OLD}_* && rm -f $P/${OLD}_*
cp $G/bench.json $P/${TAG}_bench.json
cp $G/bench_under_rocprof.json $P/${TAG}_bench_under_rocprof.json
cp $(ls $G/stats/*/*kernel_stats.csv $G/stats/*kernel_stats.csv 2>/dev/null | head -1) $P/${TAG}_kernel_stats.csv 2>/dev/null || cp $G/kernel_stats_from_db.csv $P/${TAG}_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  cp $(ls $G/pmc_$C/*/*counter_collection.csv $G/pmc_$C/*counter_collection.csv 2>/dev/null | head -1) $P/${TAG}_pmc_hg_reduce_$C.csv
done
cp $G/pmc_traffic.json $P/pmc_traffic.json
cp $G/mfma_util.json $P/r02_mfma_util.json
cp $G/hbm_bytes_per_kernel.txt $P/${TAG}_hbm_bytes_per_kernel.txt
cp $G/hbm_bytes_per_kernel_steady.txt $P/${TAG}_hbm_bytes_per_kernel_steady.txt
cp $G/host_vs_gpu.txt $P/${TAG}_host_vs_gpu.txt
cp $G/timeline_distill.txt $P/${TAG}_timeline_distill.txt
cp $G/timeline_no_distill.txt $P/${TAG}_timeline_no_distill.txt
cp $G/vit.txt $P/r02_vit.txt
cat $G/render_512.txt $G/render_1024.txt > $P/r02_render.txt
ls -la $P/${TAG}_* $P/pmc_traffic.json $P/r02_*

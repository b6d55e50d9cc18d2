#!/bin/bash
# Records the round's judged evidence on the GPU box: default bench line, rocprofv3 kernel stats of the same command,
# and the two PMC passes (FETCH_SIZE, WRITE_SIZE -- separate runs, kernel-trace only) for the dominant kernel.
# usage (from the repo root, on the GPU box): bash tools/gpu_record.sh <tag> [kernel-regex]
set -u
TAG=${1:-r01}
KREGEX=${2:-k_hg_reduce}   # kernels the PMC passes collect (the dominant one: the fused backward + Adam reduce pass)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# counter passes first: the bench line below quotes their results (profiles/pmc_traffic.json -> roofline.traffic)
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --kernel-include-regex $KREGEX --output-format csv -d $OUT/pmc_$C -o pmc -- \
      python $ROOT/bench.py --steps 4 --warmup 2 --cpu-baseline-seconds 0 --other-workloads none --steady-steps 0 > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
done
python $ROOT/tools/pmc_traffic.py $OUT --out $OUT/pmc_traffic.json > $OUT/pmc_traffic.log 2>&1 && cp $OUT/pmc_traffic.json $ROOT/profiles/pmc_traffic.json
python $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python $ROOT/bench.py --other-workloads none > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
find $OUT -name "*.csv" | head -20

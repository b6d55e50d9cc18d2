#!/bin/bash
# Cache / fabric counters of the F = 2 grid kernels: is the 8-byte-gather forward (k_hashgrid_fwd<2>) and the
# fixed-point reduce (k_hg_reduce_fx<2>) bound by the L2 -> L1 line fills?  One rocprofv3 --pmc pass per counter group (kernel-trace
# only), summed per kernel.  usage: tools/f2_counters.sh <tag>
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$(pwd)
out=$ROOT/gpurun_out/${1:-r03_f2}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "\b\(TCP\|TCC\|TA\|TD\|SQ\)_[A-Z0-9_]*\b" | sort -u > $out/avail_counters.txt
wc -l $out/avail_counters.txt
i=0
for grp in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  for which in fwd bwd; do
    if [ $which = fwd ]; then cmd="python $ROOT/tools/microbench_hgfwd.py"; env="CASES=f2,f8 REPS=5"; else cmd="python $ROOT/tools/microbench_hgadam.py"; env="CASES=f2 REPS=5"; fi
    env $env rocprofv3 --pmc $grp --kernel-trace --kernel-include-regex "k_hashgrid_fwd|k_hg_reduce_fx" --output-format csv -d $out/pmc_${which}_$i -o pmc -- $cmd > $out/pmc_${which}_$i.log 2>&1
  done
done
python $ROOT/tools/pmc_table.py $out | tee $out/f2_counter_table.txt

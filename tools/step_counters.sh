#!/bin/bash
# Wave-stall split, instruction mix and LDS conflicts of the train step's matrix / chain kernels (two rocprofv3 PMC passes over a
# short bench run, kernel-trace only).   usage on the GPU box: bash tools/step_counters.sh <tag> [kernel-regex]
TAG=${1:-stepc}
KRE=${2:-"k_mlp_chain|k_gemm_ws_b3|k_wgrad_full|k_gemm_rows_b3|k_gemm_wgrad|k_hg_reduce|k_hg_scatter|k_hashgrid_fwd"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD=${COUNTERS_CMD:-"python $ROOT/bench.py --steps 4 --warmup 3 --cpu-baseline-seconds 0 --other-workloads none --steady-steps 0"}  # COUNTERS_CMD: another workload (tools/bench_render.py, tools/bench_vit.py)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
    --kernel-trace --kernel-include-regex "$KRE" --output-format csv -d $OUT/p1 -o pmc -- $CMD > /dev/null 2> $OUT/p1.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES \
    --kernel-trace --kernel-include-regex "$KRE" --output-format csv -d $OUT/p2 -o pmc -- $CMD > /dev/null 2> $OUT/p2.err
python - <<PY > $OUT/step_counters.txt
import csv, glob, collections, re
for p in ("p1", "p2"):
    cc = glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True)
    kt = glob.glob("$OUT/%s/**/*kernel_trace.csv" % p, recursive=True)
    if not cc: print(p, "no counters"); continue
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[0]))}
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    first = None
    for r in csv.DictReader(open(cc[0])):
        first = first or r["Counter_Name"]
        key = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void snf::", "")[:70]
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == first: agg[key]["n"] += 1; agg[key]["ns"] += dur.get(r["Dispatch_Id"], 0)
    for key, a in sorted(agg.items(), key=lambda kv: -kv[1]["ns"]):
        n = max(a["n"], 1)
        print(p, key, "launches", int(n), "avg_us", round(a["ns"] / n / 1e3, 1), {k: round(v / n) for k, v in a.items() if k not in ("n", "ns")})
PY
rm -rf $OUT/p1 $OUT/p2
cut -c1-600 $OUT/step_counters.txt

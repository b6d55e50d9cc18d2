#!/bin/bash
# counters of the split-operand GEMM on the encoder's four shapes: matrix-pipe busy fraction, wave stall split, effective clock
# usage on the GPU box: bash tools/gemm_planes_counters.sh <tag> -> gpurun_out/<tag>/gpc.txt
TAG=${1:-gpc}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $OUT/p1 -o pmc -- python $ROOT/tools/bench_gemm_planes.py > $OUT/b1.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD \
    --kernel-trace --output-format csv -d $OUT/p2 -o pmc -- python $ROOT/tools/bench_gemm_planes.py > $OUT/b2.txt 2>&1
python - <<PY > $OUT/gpc.txt
import csv, glob, collections, re
for p in ("p1", "p2"):
    cc = glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True)
    kt = glob.glob("$OUT/%s/**/*kernel_trace.csv" % p, recursive=True)
    if not cc: print(p, "no counters"); continue
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[0]))}
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(cc[0])):
        if "k_gemm_planes" not in r["Kernel_Name"]: continue
        key = (re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void snf::", ""), r["Grid_Size"])
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        agg[key]["n:" + r["Counter_Name"]] += 1
        if r["Counter_Name"] in ("SQ_BUSY_CYCLES", "SQ_INSTS_LDS"): agg[key]["ns"] += dur.get(r["Dispatch_Id"], 0)
    for key, a in sorted(agg.items()):
        n = max(a.get("n:SQ_BUSY_CYCLES", 0), a.get("n:SQ_INSTS_LDS", 0), 1)
        print(key, "launches", int(n), "avg_us", round(a["ns"] / n / 1e3, 1), {k: round(v / n) for k, v in a.items() if not k.startswith("n:") and k != "ns"})
PY
cat $OUT/gpc.txt | cut -c1-600

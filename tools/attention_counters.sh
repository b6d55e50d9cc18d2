#!/bin/bash
# wave-stall split and instruction mix of the attention kernel on the encoder's two shapes (tools/bench_attention.py)
TAG=${1:-attc}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/tools/bench_attention.py 2>&1 | grep -v amdgpu > $OUT/plain.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $OUT/p1 -o pmc -- python $ROOT/tools/bench_attention.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES \
    --kernel-trace --output-format csv -d $OUT/p2 -o pmc -- python $ROOT/tools/bench_attention.py > /dev/null 2>&1
python - <<PY > $OUT/attc.txt
import csv, glob, collections, re
for p in ("p1", "p2"):
    cc = glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True)
    kt = glob.glob("$OUT/%s/**/*kernel_trace.csv" % p, recursive=True)
    if not cc: print(p, "no counters"); continue
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[0]))}
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    first = None
    for r in csv.DictReader(open(cc[0])):
        if "k_attention" not in r["Kernel_Name"]: continue
        first = first or r["Counter_Name"]
        key = (re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void snf::", ""), r["Grid_Size"])
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == first: agg[key]["n"] += 1; agg[key]["ns"] += dur.get(r["Dispatch_Id"], 0)
    for key, a in sorted(agg.items()):
        n = max(a["n"], 1)
        print(key, "launches", int(n), "avg_us", round(a["ns"] / n / 1e3, 1), {k: round(v / n) for k, v in a.items() if k not in ("n", "ns")})
PY
cat $OUT/plain.txt $OUT/attc.txt | cut -c1-700

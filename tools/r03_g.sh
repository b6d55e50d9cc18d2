#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$(pwd)
out=$ROOT/gpurun_out/${1:-r03g}; mkdir -p $out
SNF_PARITY_VERBOSE=1 python -m pytest tests -m gpu -x -q -s > $out/tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/tests.log | cut -c1-300
grep "grad_parity" $out/tests.log | cut -c1-2500 > $out/parity_reports.txt
tools/ab_env.sh SNF_HG_SPARSE_LEVELS=0 snf_hashgrid_bwd_presorted_adam_pair/F8L12+12 snf_adam_step_rows 2>&1 | cut -c1-300 | tee $out/ab.txt
python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python - <<'PY'
import json,sys
d=json.load(open(sys.argv[1] if len(sys.argv)>1 else "gpurun_out/r03g/bench.json"))
print({k:d[k] for k in ("ms_per_step","value","step_frac_of_hbm_peak")}, d["roofline"]["frac"], d["roofline"]["serial"]["frac"])
print(json.dumps(d["other_workloads"])[:1500])
print([ (o["kernel"],o["frac"],o.get("peak")) for o in d["roofline_other_kernels"]])
PY

#!/bin/bash
# What is each kernel's time worth INSIDE the concurrent step?  The step is timed with one launch key left out at a time
# (results are garbage, the timing is not): step(all) - step(without k) = the marginal cost of k in the three-stream schedule,
# to be compared with its stand-alone duration.   usage: tools/ablate_step.sh <outfile> key [key ...]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=$1; shift
one() {
  env SNF_ABLATE_SKIP="$1" python bench.py --allow-ablation --steps 40 --warmup 8 --cpu-baseline-seconds 0 --other-workloads none --steady-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('${1:-none}'.ljust(48), 'step', round(d['ms_per_step'],3))"
}
{ one ""; for k in "$@"; do one "$k"; done; one ""; } | tee $out

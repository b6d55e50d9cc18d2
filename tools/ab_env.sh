#!/bin/bash
# same-box A/B of an environment toggle:  tools/ab_env.sh VAR=VALUE [kernel-key ...]   (runs off,on,off,on)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
tog="$1"; shift
keys="$*"
one() {
  env $1 python bench.py --steps 30 --warmup 5 --cpu-baseline-seconds 0 --other-workloads none --steady-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step_serial']
print('$1'.ljust(24), 'step', round(d['ms_per_step'],3), 'fb', round(d['fwd_bwd_only']['ms_per_step'],3), {q: k.get(q) for q in '$keys'.split()})"
}
one "_SNF_AB=off"; one "$tog"; one "_SNF_AB=off"; one "$tog"

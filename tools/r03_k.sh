#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py -q -x -k "hashgrid" 2>&1 | tail -2 | cut -c1-200
CASES=f2,f2p POS=real python tools/microbench_hgfwd.py 2>/dev/null | grep "^f" | cut -c1-200
python tools/bench_render.py 2>/dev/null | tail -1 | cut -c1-120
python bench.py --steps 40 --warmup 8 --cpu-baseline-seconds 0 --other-workloads none 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step_serial']; print('step', round(d['ms_per_step'],3), {q:k.get(q) for q in ('snf_hashgrid_fwd/F2L16','snf_hashgrid_fwd/F2L5')})"

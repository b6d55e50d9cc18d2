#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py -q -x -k "hashgrid" 2>&1 | tail -2 | cut -c1-200
CASES=f2,f2p POS=real python tools/microbench_hgfwd.py 2>/dev/null | grep "^f" | cut -c1-200
CASES=f2,f2p python tools/microbench_hgfwd.py 2>/dev/null | grep "^f" | cut -c1-200
python tools/bench_render.py 2>/dev/null | tail -1 | cut -c1-120

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out/${1:-r03j}; mkdir -p $out
python -m pytest tests/test_vit_gpu.py tests/test_model_gpu.py -q -x -k "full_depth or preprocess or eight_ranks or bench_default" 2>&1 | tail -4 | cut -c1-300
CASES=f2,f8,f2p POS=real python tools/microbench_hgfwd.py 2>/dev/null | grep "^f" | cut -c1-200
tools/f2_counters.sh ${1:-r03j}_f2 2>&1 | tail -60 | cut -c1-200

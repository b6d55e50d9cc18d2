#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
python -m pytest tests/test_ops_gpu.py -q -x -k "fused_through_lds or accumulation_renderer" 2>&1 | tail -3 | cut -c1-300
python -m pytest tests/test_model_gpu.py -q -x -k "render or eval" 2>&1 | tail -3 | cut -c1-300
python tools/bench_render.py 2>/dev/null | tail -1 | cut -c1-120
RES=1024 python tools/bench_render.py 2>/dev/null | tail -1 | cut -c1-120

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s22; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu -k "attention or vit or encoder or conv or patch or sam_utils or embed" 2>&1 | tail -8 > $O/tests.txt
python tools/bench_vit.py 2>/dev/null | head -12 >> $O/tests.txt
cat $O/tests.txt

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s22; mkdir -p $O
timeout 900 python -m pytest tests/test_vit_gpu.py -x -q -m gpu -k "position_terms_of_large" 2>&1 | tail -15 > $O/tests.txt
cat $O/tests.txt

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s14; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_step_program_gpu.py -x -q -k "colour_net or mlp64 or ministep or trajectory" 2>&1 | tail -30 > $O/tests.txt


cat $O/tests.txt

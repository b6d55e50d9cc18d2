#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s15; mkdir -p $O; : > $O/tests.txt
for i in 1 2 3 4 5 6; do
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_step_program_gpu.py -x -q -k "colour_net or mlp64 or ministep or trajectory" 2>&1 | grep -E "passed|failed|Error|assert|FAILED" | head -12 >> $O/tests.txt
done
cat $O/tests.txt

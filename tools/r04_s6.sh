#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s6; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_step_program_gpu.py tests/test_fullsize_gpu.py -x -q -k "mlp_tiny or ministep or trajectory or composed or schedule" 2>&1 | tail -3 > $O/tests.txt
for i in 1 2 3; do timeout 200 python bench.py --steps 40 --warmup 10 --cpu-baseline-seconds 0 --other-workloads no_distill_4096x128 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step_serial']
print(round(d['ms_per_step'],3), 'fb', round(d['fwd_bwd_only']['ms_per_step'],3), 'serial', d['serial_step_ms'], 'nodistill', d['other_workloads']['no_distill_4096x128']['ms_per_step'], {n:v for n,v in k.items() if 'tiny' in n})" >> $O/bench.txt; done
cat $O/tests.txt $O/bench.txt

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s5; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "mlp64 or chain or nerfacto" 2>&1 | tail -3 > $O/tests.txt
for i in 1 2; do timeout 200 python bench.py --steps 40 --warmup 10 --cpu-baseline-seconds 0 --other-workloads none 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step_serial']
print(round(d['ms_per_step'],3), 'fb', round(d['fwd_bwd_only']['ms_per_step'],3), 'serial', d['serial_step_ms'], {n:v for n,v in k.items() if 'mlp64' in n})" >> $O/bench.txt; done
cat $O/tests.txt $O/bench.txt

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s13; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "colour_net or mlp64" 2>&1 | tail -2 > $O/tests.txt
( python tools/bench_render.py; python tools/bench_render.py ) 2>/dev/null | grep "^render" | cut -c1-60 > $O/render.txt
for i in 1 2; do timeout 200 python bench.py --steps 40 --warmup 10 --cpu-baseline-seconds 0 --other-workloads none 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step_serial']
print(round(d['ms_per_step'],3), 'fb', round(d['fwd_bwd_only']['ms_per_step'],3), {n:v for n,v in k.items() if 'mlp64' in n})" >> $O/bench.txt; done
timeout 200 python bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 --other-workloads none 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('driver args:', round(d['ms_per_step'],3), d['untimed_steps_before_timed_region'])" >> $O/bench.txt
cat $O/tests.txt $O/render.txt $O/bench.txt

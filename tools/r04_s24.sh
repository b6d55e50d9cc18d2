#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s24; mkdir -p $O; : > $O/out.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 >> $O/out.txt
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -4 >> $O/out.txt
for i in 1 2 3; do
python bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 --other-workloads none 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver args', round(d['ms_per_step'],3), d['roofline']['frac'])" >> $O/out.txt
python bench.py --steps 100 --warmup 10 --cpu-baseline-seconds 0 --other-workloads none 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('100 steps  ', round(d['ms_per_step'],3), d['roofline']['frac'])" >> $O/out.txt
done
cat $O/out.txt

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s25; mkdir -p $O; : > $O/out.txt
for i in 1 2 3 4; do for rs in 8 0; do
python bench.py --steps 20 --warmup 5 --resettle $rs --cpu-baseline-seconds 0 --other-workloads none 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver args resettle $rs', round(d['ms_per_step'],3))" >> $O/out.txt
done; done
cat $O/out.txt

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s7; mkdir -p $O
one() { timeout 300 python bench.py --steps $1 --warmup $2 --cpu-baseline-seconds 0 --other-workloads none 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('steps $1 warmup $2:', round(d['ms_per_step'],3), 'fb', round(d['fwd_bwd_only']['ms_per_step'],3))" >> $O/runlen.txt; }
for r in 1 2; do one 20 5; one 100 10; one 40 8; one 300 10; done
python - > $O/persteps.txt 2>/dev/null <<'PY'
import sys, time, torch
sys.path.insert(0, '.')
import bench
tr = bench.build_trainer(dict(bench.WORKLOADS["distill_4096x128"], world=1), 0, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
marks = []
for i in range(160):
    tr.train_iteration(i)
    if i % 10 == 9:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        marks.append(round((t1 - t0) / 10 * 1e3, 3))
        t0 = time.perf_counter()
print("ms per step in blocks of 10 steps (sync every 10):", marks)
PY
cat $O/runlen.txt $O/persteps.txt

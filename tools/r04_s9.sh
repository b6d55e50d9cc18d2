#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s9; mkdir -p $O
python - > $O/drift.txt 2>/dev/null <<'PY'
import sys, time, torch
sys.path.insert(0, '.')
import bench
from samnerf_amd import ops
tr = bench.build_trainer(dict(bench.WORKLOADS["distill_4096x128"], world=1), 0, 1)
def serial(step0, n=4):
    torch.cuda.synchronize()
    tr.overlap = False
    ps, ops.PRESORT_SIDE_STREAM = ops.PRESORT_SIDE_STREAM, False
    ops.enable_kernel_timing("all")
    for i in range(n):
        tr.train_iteration(step0 + i)
    br = ops.kernel_timing_summary()
    ops.enable_kernel_timing(None)
    tr.overlap = True
    ops.PRESORT_SIDE_STREAM = ps
    return {k: v["total_ms"] / n for k, v in br.items()}
step = 0
for i in range(10):
    tr.train_iteration(step); step += 1
a = serial(step); step += 4
for i in range(150):
    tr.train_iteration(step); step += 1
b = serial(step); step += 4
print("serial sum early %.3f late %.3f" % (sum(a.values()), sum(b.values())))
for k in sorted(a, key=lambda k: -(a[k] - b.get(k, 0))):
    if abs(a[k] - b.get(k, 0)) > 0.003:
        print(f"{k:55s} early {a[k]:.4f}  late {b.get(k, 0):.4f}  delta {a[k] - b.get(k, 0):+.4f}")
ps = tr.pipeline.model.proposal_sampler
print("anneal now", ps._anneal)
PY
cat $O/drift.txt

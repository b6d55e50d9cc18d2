"""N full train steps of a BASELINE workload and nothing else (no serial replay, no fwd/bwd-only phase): the process the
steady-state byte table is differenced from (tools/pmc_steady.sh).   usage: python tools/steady_steps.py <steps> [workload]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
steps = int(sys.argv[1])
w = dict(bench.WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else "distill_4096x128"], world=1)
tr = bench.build_trainer(w, 0, 1)
for i in range(steps):
    tr.train_iteration(i)
tr.synchronize()
torch.cuda.synchronize()
print("steps", steps, "static", tr._program is not None)

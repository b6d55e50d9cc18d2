"""How long does the host need to ENQUEUE a train step vs how long the GPU needs to run it?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
w = bench.WORKLOADS[os.environ.get("WL", "distill_4096x128")]
tr = bench.build_trainer(w, 0, 1)
for i in range(5):
    tr.train_iteration(i)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for i in range(n):
    tr.train_iteration(5 + i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/n:.3f} ms/step ; total {1e3*(t2-t0)/n:.3f} ms/step ; gpu drain after last enqueue {1e3*(t2-t1):.3f} ms")

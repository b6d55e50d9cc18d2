"""Feasibility probe: capture one whole train iteration (3 streams, autograd, MIOpen conv, fused Adam) in a HIP graph and
time its replay against eager enqueue.  Host scalars (lr, Adam step, anneal) are frozen in the graph -- timing only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
w = bench.WORKLOADS[os.environ.get("WORKLOAD", "distill_4096x128")]
tr = bench.build_trainer(w, 0, 1)
tr.pipeline_steps = False
for i in range(5):
    tr.train_iteration(i)
torch.cuda.synchronize()
def timeit(fn, n=30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager, joined streams   ms/step", round(timeit(lambda i: tr.train_iteration(10 + i)), 3))
tr.pipeline_steps = True
print("eager, pipelined        ms/step", round(timeit(lambda i: tr.train_iteration(50 + i)), 3))
tr.pipeline_steps = False
tr.synchronize(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
g.register_generator_state(tr.pipeline.datamanager.gen)
with torch.cuda.graph(g):
    loss, loss_dict, metrics = tr.train_iteration(100)
torch.cuda.synchronize()
print("captured; loss", float(loss))
print("graph replay            ms/step", round(timeit(lambda i: g.replay()), 3))
print("loss after replays", float(loss), {k: round(float(v), 5) for k, v in loss_dict.items()})
t0 = time.perf_counter()
for i in range(30): g.replay()
host = (time.perf_counter() - t0) / 30 * 1e3
torch.cuda.synchronize()
print("host ms per replay call", round(host, 3))

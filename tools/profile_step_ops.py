"""List the torch-level ops of one train step (shapes + device time) to spot glue kernels outside the C-ABI."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
w = bench.WORKLOADS["distill_4096x128"]
tr = bench.build_trainer(w, 0, 1)
for i in range(4):
    tr.train_iteration(i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    for i in range(4, 7):
        tr.train_iteration(i)
    torch.cuda.synchronize()
rows = prof.key_averages(group_by_input_shape=True)
sel = [r for r in rows if r.key.startswith("aten::") and r.device_time_total > 50]
sel.sort(key=lambda r: -r.device_time_total)
for r in sel[:40]:
    print(f"{r.key:28s} calls={r.count:4d} dev_us/step={r.device_time_total/3:9.1f} shapes={r.input_shapes}")
if os.environ.get("STACK"):
    pass

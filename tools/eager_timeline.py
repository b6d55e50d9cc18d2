"""Eager-mode GPU timeline of the train step from HIP events around every C-ABI launch (no profiler attached): shows
which stream runs what when, where a stream sits idle, and how many C-ABI kernels overlap."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from samnerf_amd import ops
w = bench.WORKLOADS[os.environ.get("WORKLOAD", "distill_4096x128")]
tr = bench.build_trainer(w, 0, 1)
tr.pipeline_steps = os.environ.get("PIPE", "1") == "1"
for i in range(8):
    tr.train_iteration(i)
torch.cuda.synchronize()
ops.enable_kernel_timing("all")
base = torch.cuda.Event(enable_timing=True)
NS = 4
marks = []
for i in range(NS):
    m = torch.cuda.Event(enable_timing=True); m.record(); marks.append(m)
    if i == 0: base = m
    tr.train_iteration(8 + i)
tl = ops.kernel_timeline(base)
ops.enable_kernel_timing(None)
mk = [base.elapsed_time(m) for m in marks]
print("step starts (main stream) ms:", [round(x, 3) for x in mk])
sids = sorted({t[2] for t in tl})
name = {sid: f"s{i}" for i, sid in enumerate(sids)}
lo, hi = mk[2], mk[3]          # third step window
print(f"window {lo:.3f} .. {hi:.3f} ms  ({hi - lo:.3f} ms)")
sel = [t for t in tl if t[1] > lo and t[0] < hi]
busy = collections.Counter()
for a, b, sid, key in sel:
    busy[name[sid]] += min(b, hi) - max(a, lo)
print("C-ABI busy ms per stream in window:", {k: round(v, 3) for k, v in busy.items()})
pts = []
for a, b, sid, key in sel:
    pts += [(max(a, lo), 1), (min(b, hi), -1)]
pts.sort()
lvl, last, hist = 0, lo, collections.Counter()
for t, d in pts:
    hist[lvl] += t - last; last = t; lvl += d
hist[lvl] += hi - last
print("C-ABI concurrency histogram (ms):", {k: round(v, 3) for k, v in sorted(hist.items())})
MIN_MS = float(os.environ.get("MIN_MS", "0.04"))
for a, b, sid, key in sel:
    if b - a > MIN_MS:
        print(f"{a - lo:8.3f} {b - a:7.3f} {name[sid]} {key}")
if os.environ.get("PER_STREAM") == "1":  # every launch of the window stream by stream, with the idle gap in front of it
    for sid in sids:
        print(f"--- {name[sid]}")
        prev = None
        for a, b, s2, key in sorted(t for t in sel if t[2] == sid):
            gap = (a - prev) if prev is not None else 0.0
            print(f"{a - lo:8.3f} {b - a:7.3f} gap {gap:6.3f}  {key}")
            prev = b

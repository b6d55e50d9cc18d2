"""Microbenchmark of the dense-layer kernels at the feature-head shapes (N = 65536 samples)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import samnerf_amd
from samnerf_amd import ops

N = int(os.environ.get("N", 65536))
shapes = [(192, 256), (256, 256), (256, 192), (64, 64), (32, 64), (64, 16), (2304, 256)]
ops.enable_kernel_timing("all")
for (I, O) in shapes:
    n = (4096 if I > 1024 else N) if max(I, O) >= 192 else N * 8
    x = torch.randn((n, I), device="cuda", requires_grad=True)
    w = (torch.randn((O, I), device="cuda") / I ** 0.5).requires_grad_(True)
    gy = torch.randn((n, O), device="cuda")
    for _ in range(5):
        y = ops.linear(x, w, None, ops.ACT_RELU)
        y.backward(gy)
s = ops.kernel_timing_summary()
for k, v in s.items():
    name, _, tag = k.partition("/")
    I, O = (int(t) for t in tag.split("x"))
    n = (4096 if I > 1024 else N) if max(I, O) >= 192 else N * 8
    fl = 2.0 * n * I * O
    print(f"{k:36s} {v['avg_ms']*1e3:8.1f} us  {fl / (v['avg_ms']*1e-3) / 1e12:6.1f} TFLOP/s")

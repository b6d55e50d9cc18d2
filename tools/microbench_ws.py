"""The three GEMMs of a feature head's first layer alone, at the step's shapes (R rays x 16 samples, 192 -> 256):
snf_linear_fwd_mean (level-major X, rendered epilogue), snf_linear_bwd_data_rows (mask bits, level-major dX),
snf_linear_bwd_weight_rows.  us per launch, with the HBM floor of each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import samnerf_amd
from samnerf_amd import ops as m

R, K, I, O = int(os.environ.get("R", 4096)), 16, int(os.environ.get("I", 192)), int(os.environ.get("O", 256))
N = R * K
DEV = "cuda"
m.set_gemm_mode("bf16x3")
g = torch.Generator(device=DEV).manual_seed(0)
x = torch.randn((I // 8, N, 8), device=DEV, generator=g) * 0.3
w = torch.randn((O, I), device=DEV, generator=g) * 0.1
wk = torch.rand((R, K), device=DEV, generator=g)
dyg = torch.randn((R, O), device=DEV, generator=g)
hbar = torch.empty((R, O), device=DEV)
mask = torch.zeros((N, O // 8), device=DEV, dtype=torch.uint8)
dx = torch.empty((N * I,), device=DEV)
dw = torch.zeros((O, I), device=DEV)
nb = int(m._L().snf_linear_bwd_weight_workspace_bytes(N, I, O))
ws = torch.empty((nb // 4,), device=DEV)
st = m._stream()
calls = {
    "fwd_mean": lambda: m._launch("snf_linear_fwd_mean", m._p(x), m._p(w), N, I, O, -8, m._p(wk), K, m._p(hbar), m._p(mask), None, O, st),
    "bwd_data_rows": lambda: m._launch("snf_linear_bwd_data_rows", m._p(dyg), m._p(wk), K, m._p(mask), 1, m._p(w), N, I, O, O, O // 8, -8,
                                       m.ACT_RELU, m._p(dx), st),
    "bwd_weight_rows": lambda: m._launch("snf_linear_bwd_weight_rows", m._p(dyg), m._p(wk), K, m._p(mask), 1, m._p(x), N, I, O, O, O // 8, -8,
                                         m.ACT_RELU, m._p(dw), m._p(ws), nb, st),
}
floor = {"fwd_mean": N * I * 4 + N * O / 8, "bwd_data_rows": N * I * 4 + N * O / 8, "bwd_weight_rows": N * I * 4 + N * O / 8}
for name, f in calls.items():
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"{name:18s} {us:8.1f} us   ({floor[name] / 1e6:.0f} MB: {floor[name] / us / 1e6:.2f} TB/s;  "
          f"{2.0 * N * I * O * 3 / us / 1e6:.0f} TFLOP/s bf16)")

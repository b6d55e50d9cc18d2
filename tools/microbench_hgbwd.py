"""Microbenchmark of the bucketed hash-grid backward on bench-shaped inputs (one L12/F8 grid, N = 65536)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import samnerf_amd
from samnerf_amd import ops
from oracle import samnerf_oracle as O

def run(N, L, F, T, mn, mx, ld, clustered):
    g = O.GridSpec(L, F, T, mn, mx)
    sc = g.scalings().cuda()
    gen = torch.Generator(device="cuda").manual_seed(0)
    if clustered:
        d = torch.randn((N, 3), device="cuda", generator=gen); d = d / d.norm(dim=-1, keepdim=True)
        u = (d * (1.0 + torch.rand((N, 1), device="cuda", generator=gen)) + 2.0) / 4.0
    else:
        u = torch.rand((N, 3), device="cuda", generator=gen)
    table = torch.zeros((g.rows * F,), device="cuda", requires_grad=True)
    table.main_grad = torch.zeros_like(table)
    gy = torch.randn((N, ld), device="cuda", generator=gen)
    ops.enable_kernel_timing("all")
    for it in range(6):
        out = ops.hashgrid(u, [table], ((sc, L, F, T),))
        out.backward(gy[:, : L * F].contiguous() if ld == L * F else gy[:, :L*F].contiguous())
    s = ops.kernel_timing_summary()
    return {k: round(v["avg_ms"], 4) for k, v in s.items()}

CASES = {"f8u": ("F8 uniform  ", (65536, 12, 8, 19, 128, 512, 96, False)),
         "f8c": ("F8 clustered", (65536, 12, 8, 19, 16, 128, 96, True)),
         "f2": ("F2 field    ", (524288, 16, 2, 19, 16, 2048, 32, True)),
         "f2p": ("F2 proposal ", (262144, 5, 2, 17, 16, 128, 10, True))}
for r in (16, 32, 64, 128, 256, 512, 1024, 2048):
    CASES[f"f2r{r}"] = (f"F2 L2 res {r}", (524288, 2, 2, 19, r, r, 4, True))
    CASES[f"f8r{r}"] = (f"F8 L2 res {r}", (65536, 2, 8, 19, r, r, 16, True))
if True:
  for c in os.environ.get("CASES", "f8u,f8c,f2").split(","):
    print("  ", CASES[c][0], run(*CASES[c][1]))

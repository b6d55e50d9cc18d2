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

for dbg in os.environ.get("DBGS", "0").split(","):
    os.environ["SNF_HG_LONG"] = dbg
    print("dbg", dbg, "F8 uniform  ", run(65536, 12, 8, 19, 128, 512, 96, False))
    print("dbg", dbg, "F8 clustered", run(65536, 12, 8, 19, 16, 128, 96, True))
    print("dbg", dbg, "F2 field    ", run(524288, 16, 2, 19, 16, 2048, 32, True))

"""The eval render's two hash-grid forwards per chunk of 32 768 rays (proposal grid: 2.1 M samples x 5 levels, row-major; field grid:
4.2 M samples x 16 levels, level-major), samples along random segments through the unit cube, HIP-event time of 5 launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import samnerf_amd
from samnerf_amd import ops

for name, S, L, T, lo, hi, planar in [("proposal", 64, 5, 17, 16, 128, False), ("field", 128, 16, 19, 16, 2048, True)]:
    R = 32768
    N = R * S
    g = torch.Generator(device="cuda").manual_seed(1)
    growth = (hi / lo) ** (1.0 / (L - 1))
    sc = torch.floor(lo * growth ** torch.arange(L, dtype=torch.float64)).float().cuda()
    table = (torch.rand(((L << T), 2), device="cuda", generator=g) * 2 - 1) * 0.1
    o, e = torch.rand((R, 1, 3), device="cuda", generator=g), torch.rand((R, 1, 3), device="cuda", generator=g)
    u = (o + (e - o) * torch.linspace(0, 1, S, device="cuda").view(1, S, 1)).reshape(N, 3).contiguous()
    for _ in range(2):
        ops.hashgrid_fwd_raw(u, table, sc, L, 2, T, planar)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.hashgrid_fwd_raw(u, table, sc, L, 2, T, planar)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    per = []
    for l in range(L):  # one level at a time (the kernel's grid runs level by level anyway)
        tl, sl = table[l << T:(l + 1) << T], sc[l:l + 1].contiguous()
        ops.hashgrid_fwd_raw(u, tl, sl, 1, 2, T, planar)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            ops.hashgrid_fwd_raw(u, tl, sl, 1, 2, T, planar)
        e1.record()
        torch.cuda.synchronize()
        per.append(e0.elapsed_time(e1) / 5 * 1e3)
    print(f"{name:9s} per level (resolution: us): " + "  ".join(f"{int(r)}: {t:.0f}" for r, t in zip(sc.tolist(), per)))
    print(f"{name:9s} N={N} L={L} T={T}: {us:7.1f} us  ({N * L / us:6.1f} M (sample, level) / s x 1e-6, {N * L * 6 / us / 256 / 2.1e3:5.2f} lane addresses / clock / CU at 6 per item, 2.1 GHz)")

"""Hash-grid forward alone (snf_hashgrid_fwd), bench-shaped: f2 = the field grid (16 levels, F = 2, T = 19, N = 524288, level-major
output), f8 = one feature grid 128 -> 512 (12 levels, F = 8, N = 65536).  POS=real: positions of a real train step.
usage: CASES=f2,f8 REPS=20 python tools/microbench_hgfwd.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import samnerf_amd  # noqa: F401
from samnerf_amd import ops, tcnn_compat

CASES = {"f2": (524288, 16, 2, 19, 16, 2048, True), "f8": (65536, 12, 8, 19, 128, 512, False), "f2p": (262144, 5, 2, 17, 16, 128, True)}
REPS = int(os.environ.get("REPS", "20"))
REAL = {}
if os.environ.get("POS") == "real":
    import bench
    tr = bench.build_trainer(bench.WORKLOADS["distill_4096x128"], 0, 1)
    for i in range(3):
        tr.train_iteration(i)
    tr.synchronize()
    torch.cuda.synchronize()
    REAL = {v.shape[0]: v.clone() for k, v in tr._program.bufs.items() if k in ("u0", "u1", "uk@0")}
    del tr
for name in os.environ.get("CASES", "f2,f8").split(","):
    N, L, F, T, mn, mx, clustered = CASES[name]
    growth = float(np.exp((np.log(mx) - np.log(mn)) / (L - 1)))
    enc = tcnn_compat.Encoding(3, {"otype": "HashGrid", "n_levels": L, "n_features_per_level": F, "log2_hashmap_size": T,
                                   "base_resolution": mn, "per_level_scale": growth}, device="cuda")
    gen = torch.Generator(device="cuda").manual_seed(0)
    if N in REAL:
        u = REAL[N]
    elif clustered:
        R = N // 128
        o = torch.rand((R, 1, 3), device="cuda", generator=gen) * 0.2 + 0.4
        d = torch.nn.functional.normalize(torch.randn((R, 1, 3), device="cuda", generator=gen), dim=-1)
        u = (o + d * torch.linspace(0, 0.4, 128, device="cuda").view(1, 128, 1)).clamp(0.001, 0.999).reshape(N, 3).contiguous()
    else:
        u = torch.rand((N, 3), device="cuda", generator=gen)
    out = torch.empty((L * N * F,), device="cuda")
    st = ops._stream()
    launch = lambda: ops._launch("snf_hashgrid_fwd", ops._p(u), ops._p(enc.params), ops._p(enc.scalings), N, L, F, T, ops._p(out), 0, 0, st)  # noqa: E731
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    evs = []
    for _ in range(REPS):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); launch(); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    med = ts[len(ts) // 2]
    units = float(N) * L * 8 * F * 4
    print(f"{name}: N={N} L={L} F={F}  median {med * 1e3:.1f} us  gathers {N * L * 8 / 1e6:.1f} M  algorithmic {units / 1e6:.0f} MB -> "
          f"{units / med / 1e6:.0f} GB/s ({units / med / 1e6 / 8000:.3f} of 8 TB/s)", flush=True)

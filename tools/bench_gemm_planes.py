"""The split-operand GEMM (csrc/gemm_planes.hip) on the encoder's four token GEMMs: the library's own tile choice and every tile shape
forced, same box, HIP-event time of 20 back-to-back launches.  usage: python tools/bench_gemm_planes.py [auto-only]
GP_SHAPES=2x2,1x4,-1x5 restricts the forced shapes (-1xN: the 8-wave workgroup), GP_GEMMS=lin1,qkv the layers (tools/ablate_gp.sh),
GP_CT=1: GELU + operand-plane output (the encoder's lin1) instead of fp32 + bias."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import samnerf_amd
from samnerf_amd import ops

SHAPES = [None] if len(sys.argv) > 1 else [None, (2, 5), (2, 4), (1, 5), (1, 4), (2, 2), (1, 2), None]  # (auto first AND last: order effects)
if os.environ.get("GP_SHAPES"):
    SHAPES = [tuple(int(v) for v in t.split("x")) for t in os.environ["GP_SHAPES"].split(",")]
GEMMS = [("lin1 4096x1280->5120", 4096, 1280, 5120), ("lin2 4096x5120->1280", 4096, 5120, 1280),
         ("qkv  4900x1280->3840", 4900, 1280, 3840), ("proj 4900x1280->1280", 4900, 1280, 1280)]
GEMMS += [("lin1k2 4096x2560->5120", 4096, 2560, 5120), ("lin1k4 4096x5120->5120", 4096, 5120, 5120)]  # (k slope: GP_GEMMS=lin1,lin1k2,lin1k4)
if os.environ.get("GP_GEMMS"):
    GEMMS = [g for g in GEMMS if g[0].split()[0] in os.environ["GP_GEMMS"].split(",")]
for name, M, K, Nc in GEMMS:
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn((M, K), device="cuda", generator=g)
    w = torch.randn((Nc, K), device="cuda", generator=g) * K ** -0.5
    b = torch.randn((Nc,), device="cuda", generator=g)
    ap, wp = ops.split_planes_kb(a), ops.split_weight_planes(w)
    kw = dict(act=ops.ACT_GELU, out=ops.Planes.empty(M, Nc, "cuda")) if os.environ.get("GP_CT") else {}
    for _ in range(100):  # (clocks: the first shape timed on an idle GPU reads ~15 % slow)
        ops.linear_planes(ap, wp, b, **kw)
    best = {}
    for rep in range(2):
        for si, sh in enumerate(SHAPES):
            if sh is not None and Nc % (32 * sh[1]):
                continue
            for _ in range(3):
                ops.linear_planes(ap, wp, b, shape=sh, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.linear_planes(ap, wp, b, shape=sh, **kw)
            e1.record()
            torch.cuda.synchronize()
            best[si] = min(best.get(si, 1e9), e0.elapsed_time(e1) / 20 * 1e3)
    row = []
    if Nc % 320 == 0 or Nc % 256 == 0:  # both operands through LDS (k-blocked weights)
        wkb = ops.split_weight_planes_kb(w)
        t = 1e9
        for rep in range(2):
            for _ in range(3):
                ops.linear_planes(ap, wkb, b, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.linear_planes(ap, wkb, b, **kw)
            e1.record()
            torch.cuda.synchronize()
            t = min(t, e0.elapsed_time(e1) / 20 * 1e3)
        row.append(f"lds: {t:6.1f} us ({6.0 * M * K * Nc / t / 1e6 / 2500 * 100:4.1f} % bf16 peak)")
    for si, sh in enumerate(SHAPES):
        if si not in best:
            row.append(f"{sh[0]}x{sh[1]}: n/a")
            continue
        us = best[si]
        row.append(f"{'auto' if sh is None else '%dx%d' % sh}: {us:6.1f} us ({2.0 * M * K * Nc / us / 1e6:4.0f} TF/s-eq, {6.0 * M * K * Nc / us / 1e6 / 2500 * 100:4.1f} % bf16 peak)")
    print(name, " | ".join(row))

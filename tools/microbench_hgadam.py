"""Microbenchmark of the fused hash-grid backward + Adam (snf_hashgrid_bwd_presorted_adam) alone on the GPU, bench-shaped:
  f8b  one F=8 feature grid 128->512 (12 dense levels), N = 65536 top-K samples
  f8a  one F=8 feature grid 16->128 (8 reachable-row levels + 4 dense)
  f2   the F=2 field grid 16->2048 (16 levels), N = 524288, level-major gradient
  f2p  the F=2 proposal grid 16->128 (5 levels, T = 17), N = 262144
usage: CASES=f8b,f2 REPS=20 python tools/microbench_hgadam.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import samnerf_amd  # noqa: F401
from samnerf_amd import ops, tcnn_compat

CASES = {  # N, L, F, T, base, max, planar gradient, clustered positions
    "f8b": (65536, 12, 8, 19, 128, 512, True, False),
    "f8a": (65536, 12, 8, 19, 16, 128, True, False),
    "f2": (524288, 16, 2, 19, 16, 2048, True, True),
    "f2p": (262144, 5, 2, 17, 16, 128, False, True),
    # probes of the coarse-level tail: the same sample counts with every level hashed (no bucket holds more than 8N/256 records)
    "f8ab": (65536, 24, 8, 19, 16, 512, True, False),       # both grids of a head as ONE 24-level launch (what a paired launch would cost)
    "f8a_sparse": (65536, 8, 8, 19, 16, 90, True, False),   # the reachable-row levels of f8a alone
    "f8a_dense": (65536, 4, 8, 19, 99, 128, True, False),   # its dense levels alone
    "f2c": (524288, 5, 2, 19, 16, 58, True, True),        # the field grid's five coarse levels alone
    "f2f": (524288, 11, 2, 19, 80, 2048, True, True),     # its eleven fine levels alone
    "f2_fine": (524288, 16, 2, 19, 256, 2048, True, True),
    "f2p_fine": (262144, 5, 2, 17, 128, 512, False, True),
}
REPS = int(os.environ.get("REPS", "20"))
lib = ops._L()
REAL = {}
if os.environ.get("POS") == "real":  # positions of a bench-shaped train step (fine / proposal / top-K samples) instead of synthetic ones
    import bench
    tr = bench.build_trainer(bench.WORKLOADS["distill_4096x128"], 0, 1)
    for i in range(3):
        tr.train_iteration(i)
    torch.cuda.synchronize()
    bufs = tr._program.bufs
    REAL = {v.shape[0]: v.clone() for k, v in bufs.items() if k in ("u0", "u1", "uk@0")}
    print("real positions for N =", sorted(REAL), flush=True)
    for st in (tr._side or {}).values():
        torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    del tr, bufs
for name in os.environ.get("CASES", "f8b,f8a,f2,f2p").split(","):
    if name.startswith("lvl"):  # lvl<res>[p]: one F=2 level of that resolution alone (p: the proposal grid's N and T)
        res = int(name[3:].rstrip("p"))
        CASES[name] = (262144, 1, 2, 17, res, res, False, True) if name.endswith("p") else (524288, 1, 2, 19, res, res, True, True)
    N, L, F, T, mn, mx, planar, clustered = CASES[name]
    growth = float(np.exp((np.log(mx) - np.log(mn)) / (L - 1))) if L > 1 else 1.0
    enc = tcnn_compat.Encoding(3, {"otype": "HashGrid", "n_levels": L, "n_features_per_level": F, "log2_hashmap_size": T,
                                   "base_resolution": mn, "per_level_scale": growth}, device="cuda")
    n_sparse, _ = enc.active_rows()
    gen = torch.Generator(device="cuda").manual_seed(0)
    if N in REAL:
        u = REAL[N]
    elif clustered:  # samples along rays: consecutive samples are neighbours
        R = N // 128
        o = torch.rand((R, 1, 3), device="cuda", generator=gen) * 0.2 + 0.4
        d = torch.randn((R, 1, 3), device="cuda", generator=gen)
        d = d / d.norm(dim=-1, keepdim=True)
        t = torch.linspace(0, 0.4, 128, device="cuda").view(1, 128, 1)
        u = (o + d * t).clamp(0.001, 0.999).reshape(N, 3).contiguous()
    else:
        u = torch.rand((N, 3), device="cuda", generator=gen)
    sc = enc.scalings
    n = enc.params.numel()
    p, g = enc.params.data, torch.zeros(n, device="cuda")
    m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    ld = 0 if planar else L * F
    gy = torch.randn((L * N * F,), device="cuda", generator=gen) * 1e-3
    nbytes = int(lib.snf_hashgrid_bwd_workspace_bytes(N, L, T))
    ws = torch.empty(((nbytes + 3) // 4,), device="cuda", dtype=torch.int32)
    stage = torch.empty((L * N * F,), device="cuda")
    st = ops._stream()
    nrun = ops.hashgrid_run_levels(sc) if F == 2 else 0
    nrun = int(os.environ.get("NRUN", nrun))
    ops._launch("snf_hashgrid_sort", ops._p(u), ops._p(sc), N, L, T, ops._p(ws), nbytes, st)

    fx8 = F == 8 and os.environ.get("FX8") == "1"  # the fixed-point reduce for an F = 8 grid (snf_hashgrid_bwd_presorted_adam_fx)
    scratch = torch.zeros((64,), device="cuda", dtype=torch.int32)

    sp = os.environ.get("SP") == "1"  # reachable-row levels over compact rows (snf_hashgrid_bwd_presorted_adam_sp): the WHOLE table is stepped
    rows_adam = os.environ.get("ROWS_ADAM") == "1"  # ... the round-2 equivalent: + snf_adam_step_rows on the reachable rows
    if sp or rows_adam:
        log2B = int(lib.snf_hashgrid_bucket_bits(N, T))
        ns, rows, start, longest = enc.reach_lists(log2B, int(lib.snf_hashgrid_sparse_max_rows(F)))
        _, rows64 = enc.active_rows()
        offs = (rows64 * F).to(torch.int32).contiguous() if n_sparse else None
        rest = (rows64[rows.numel():] * F).to(torch.int32).contiguous() if ns else offs  # levels [ns, n_sparse): row-Adam as before
        print(f"  reachable-row levels {n_sparse}, through the compact reduce {ns}", flush=True)

    def launch(step):
        if sp and n_sparse and ns:
            ops._launch("snf_hashgrid_bwd_presorted_adam_sp", ops._p(gy), N, L, F, T, ld, 0, nrun, ops._p(g), ops._p(ws),
                        None if planar else ops._p(stage), n_sparse, ops._p(p), ops._p(m), ops._p(v), 5e-4, 0.9, 0.999, 1e-15, step,
                        1.0, ops._p(rows), ops._p(start), ns, longest, 1, ops._p(scratch), st)
            if rest is not None and rest.numel():
                ops.adam_step_rows_(p, g, m, v, rest, F, 5e-4, 0.9, 0.999, 1e-15, step, 1.0, True)
            return
        if rows_adam and n_sparse:
            ops._launch("snf_hashgrid_bwd_presorted_adam", ops._p(gy), N, L, F, T, ld, 0, nrun, ops._p(g), ops._p(ws),
                        None if planar else ops._p(stage), n_sparse, ops._p(p), ops._p(m), ops._p(v), 5e-4, 0.9, 0.999, 1e-15, step,
                        1.0, st)
            ops.adam_step_rows_(p, g, m, v, offs, F, 5e-4, 0.9, 0.999, 1e-15, step, 1.0, True)
            return
        if fx8:
            ops._launch("snf_hashgrid_bwd_presorted_adam_fx", ops._p(gy), N, L, F, T, ld, 0, nrun, ops._p(g), ops._p(ws),
                        None if planar else ops._p(stage), n_sparse, ops._p(p), ops._p(m), ops._p(v), 5e-4, 0.9, 0.999, 1e-15, step,
                        1.0, ops._p(scratch), st)
            return
        ops._launch("snf_hashgrid_bwd_presorted_adam", ops._p(gy), N, L, F, T, ld, 0, nrun, ops._p(g), ops._p(ws),
                    None if planar else ops._p(stage), n_sparse, ops._p(p), ops._p(m), ops._p(v), 5e-4, 0.9, 0.999, 1e-15, step,
                    1.0, st)

    for i in range(3):
        launch(i + 1)
    torch.cuda.synchronize()
    evs = []
    for i in range(REPS):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        launch(4 + i)
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    med = ts[len(ts) // 2]
    fused = ((L - n_sparse) << T) * F
    units = float(N) * 8 * F * 4 * (L + n_sparse) + 24.0 * fused
    print(f"{name}: N={N} L={L} F={F} sparse_levels={n_sparse}  median {med*1e3:.1f} us  min {ts[0]*1e3:.1f}  "
          f"algorithmic {units/1e6:.0f} MB -> {units/med/1e6:.0f} GB/s ({units/med/1e6/8000:.3f} of 8 TB/s)", flush=True)

#!/usr/bin/env python3
"""bench.py -- throughput of the SAM-NeRF render-and-distill TRAIN STEP on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one full train iteration of the `samnerf_distill` method on one batch of synthetic rays: proposal
sampling, nerfacto field, compositing, top-K feature branch (SAM 256-d + ClipSeg 192-d heads, conv head),
losses, backward, gradient mean across ranks (RCCL) and the fused Adam update of all ~221 M parameters.
Metric (BASELINE.json): ray-samples/s = ranks * R * S / t_step.  Weak scaling: every rank draws its own R rays.

One JSON line is printed by rank 0; besides the driver's contract it carries
  roofline      -- the dominant kernel of the step: algorithmic bytes (or flops) per launch / its HIP-event
                   duration measured live in the timed region, against the MI355X peak;
  cpu_baseline  -- the CPU oracle (oracle/samnerf_oracle.py, a port of the reference's torch path) timed on
                   the host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import copy
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver of these boxes only supports dmabuf IPC; without this RCCL / device-tensor sharing across processes fails with
# `hipIpcGetMemHandle: invalid argument`.  Must be in the environment before the HIP runtime starts (i.e. before `import torch`).
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

WORKLOADS = {
    # BASELINE.json configs[1..3]
    "no_distill_4096x128": dict(method="samnerf_no_distill", R=4096, P=64, S=128, K=3, patch=1),
    "distill_4096x128": dict(method="samnerf_distill", R=4096, P=64, S=128, K=16, patch=4),
    "distill_16384x128": dict(method="samnerf_distill", R=16384, P=64, S=128, K=16, patch=4),
}
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP32_MATRIX_PEAK_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32
BF16_MATRIX_PEAK_TFLOPS = 2500.0  # v_mfma_f32_32x32x16_bf16, dense (MI355X_MICROARCH.md)


def mfma_peak(key: str, gemm_mode: int = 1):
    """(peak in fp32-EQUIVALENT TFLOP/s, what it is) of the matrix instruction a GEMM-shaped entry point issues.  The bf16-split
    kernels spend several bf16 products per fp32-equivalent multiply-add, so their ceiling is the bf16 dense peak divided by that
    count -- not the fp32-matrix peak (`snf_linear_bwd_data_rows` "0.84 of 157 TFLOP/s" was 0.16 of its
    own 833)."""
    name, _, tag = key.partition("/")
    name = name[:-3] if name.endswith("_sh") else name  # (the colour net with its input row formed in the loader: same arithmetic)
    name = name[:-8] if name.endswith("_density") else name  # (the base net with trunc_exp in its epilogue)
    if gemm_mode >= 1 and name.startswith("snf_linear"):
        dims = [int(x) for x in re.sub(r"[a-z]+$", "", tag).split("x")] if tag else []
        if dims and max(dims) >= 64:  # (the narrow layers -- proposal net -- stay on the fp32 matrix cores)
            return BF16_MATRIX_PEAK_TFLOPS / 3.0, "v_mfma_f32_32x32x16_bf16, 3 products per fp32-equivalent MAC (hi*hi + hi*lo + lo*hi)"
    if gemm_mode >= 1 and name == "snf_mlp64_fwd":
        return BF16_MATRIX_PEAK_TFLOPS / 6.0, "v_mfma_f32_32x32x16_bf16, 6 products per fp32-equivalent MAC (three-piece split)"
    if gemm_mode >= 1 and name == "snf_mlp64_bwd_fused" and tag.count("x") == 3:
        # two hidden layers (round 4): data-gradient chain AND weight gradients on the 3-product split (the recomputed forward's
        # six-product work is not counted as useful flops)
        return BF16_MATRIX_PEAK_TFLOPS / 3.0, "data gradient and weight gradients: v_mfma_f32_32x32x16_bf16, 3 products per fp32-equivalent MAC"
    if gemm_mode >= 1 and name == "snf_mlp64_bwd_fused":
        # data-gradient chain on the fp32 matrix cores + weight gradients on the 3-product split, half of the counted flops each
        return 2.0 / (1.0 / FP32_MATRIX_PEAK_TFLOPS + 3.0 / BF16_MATRIX_PEAK_TFLOPS), \
            "data gradient: v_mfma_f32_32x32x2_f32; weight gradient: bf16 3-product split (harmonic mean of the two peaks)"
    return FP32_MATRIX_PEAK_TFLOPS, "v_mfma_f32_32x32x2_f32"


def algorithmic_model(key: str, w: dict):
    """(bound, units per launch, unit) for a kernel key 'entry/tag' -- requested bytes, no cache credit
    (8 corners x F floats x 4 B per level per sample; backward = read-modify-write)."""
    R, P, S, K = w["R"], w["P"], w["S"], w["K"]
    name, _, tag = key.partition("/")
    name = name[:-3] if name.endswith("_sh") else name
    name = name[:-8] if name.endswith("_density") else name
    if name == "snf_hashgrid_bwd_presorted_adam_pair":  # both F = 8 grids of a head in one launch ("F8L12+12"); the launch site
        m = re.fullmatch(r"F8L(\d+)\+(\d+)", tag)        # reports its own bytes (gathers + 24 B per fused parameter)
        return ("hbm", float(R * K * (int(m.group(1)) + int(m.group(2))) * 8 * 8 * 4 * 2), "GB/s") if m else (None, None, None)
    if name in ("snf_hashgrid_fwd", "snf_hashgrid_bwd", "snf_hashgrid_bwd_sorted", "snf_hashgrid_bwd_sorted_ex",
                "snf_hashgrid_bwd_presorted", "snf_hashgrid_bwd_presorted_adam", "snf_hashgrid_bwd_presorted_adam_sp",
                "snf_hashgrid_bwd_presorted_adam_fx", "snf_hashgrid_bwd_presorted_adam_xp"):
        # (the fused backward + Adam reports its own bytes per launch -- ops._hashgrid_bwd_launch: the corner
        # contributions as below plus 24 B per parameter of the fused levels -- and roof() prefers those)
        m = re.fullmatch(r"F(\d+)L(\d+)(tp)?", tag)
        if m is None:
            return None, None, None
        F, L = int(m.group(1)), int(m.group(2))
        rw = 1 if name.endswith("fwd") else 2
        # F=8: feature grids on the R*K top-K samples; F=2: proposal grid (L=5, R*P samples) or field grid (L=16, R*S).
        # Table-parallel feature grids (multi-GPU): a launch covers the L levels this rank owns at the samples of all ranks.
        # (its forward is one launch per destination rank: R*K samples each)
        n = R * K * (w.get("world", 1) if (m.group(3) and rw == 2) else 1) if F == 8 else (R * P if L == 5 else R * S)
        return "hbm", float(n * L * 8 * F * 4 * rw), "GB/s"
    if name.startswith("snf_linear"):
        pm = tag.endswith("pm")  # conv head: second convolution, on the patch means
        rendered = tag.endswith("r")  # a head's last layer, after the weighted mean over the K samples: one row per ray
        i, o = (int(x) for x in tag.rstrip("pmr").split("x"))
        if rendered:
            n = R
        elif i >= 1024:  # conv head as GEMMs: first convolution on every ray row, second on the patch means
            n = R // (w["patch"] ** 2) if pm else R
        else:
            n = R * K if max(i, o) >= 192 else R * S
        return "mfma", 2.0 * n * i * o, "TFLOP/s"
    if name.startswith("snf_mlp64"):
        dims = [int(x) for x in tag.split("x")]
        macs = sum(a * b for a, b in zip(dims[:-1], dims[1:]))
        if name == "snf_mlp64_bwd_fused":  # data-gradient chain + weight gradients
            macs *= 2
        return "mfma", 2.0 * R * S * macs, "TFLOP/s"
    return None, None, None


def adam_bytes(trainer) -> float:
    """p, g, m, v read + p, m, v, g(zero) written: 32 B per fp32 parameter slot of every arena."""
    return 32.0 * sum(a.numel for a in trainer.optimizers.arenas.values())


def build_trainer(w: dict, local_rank: int, world: int, seed: int = 0):
    import samnerf_amd  # noqa: F401
    from samnerf_amd import configs, tcnn_compat
    tc = copy.deepcopy(configs.method_configs[w["method"]])
    dm, mc = tc.pipeline.datamanager, tc.pipeline.model
    dm.train_num_rays_per_batch = w["R"]
    dm.seed = seed
    mc.num_proposal_samples_per_ray = (w["P"],)
    mc.num_nerf_samples_per_ray = w["S"]
    mc.num_sam_samples = w["K"]
    tcnn_compat.manual_seed(seed)
    trainer = tc.setup(local_rank=local_rank, world_size=world, device=f"cuda:{local_rank}")
    trainer.setup()
    trainer.pipeline_steps = os.environ.get("SNF_PIPELINE_STEPS", "1") == "1"
    return trainer


def _cpu_baseline_worker(w: dict, budget_s: float, threads: int, seed: int, sync_dir: str = "") -> dict:
    """One process of the CPU baseline: the oracle's fwd + bwd on its own R / 16 rays with `threads` torch threads.  With `sync_dir` the
    worker reports `ready.<seed>` after its warm-up and starts its clock when the parent has written `go` (all workers warm: the timed
    windows then overlap, every process is timed on a fully loaded host)."""
    from oracle import samnerf_oracle as O
    torch.set_num_threads(threads)
    distill = w["method"] == "samnerf_distill"
    cfg = O.PathConfig(num_proposal_samples=w["P"], num_nerf_samples=w["S"], num_sam_samples=w["K"],
                       patch_size=w["patch"], distill_sam=distill, use_clipseg=distill)
    R = w["R"] // 16  # "for the full-size configs allow R to be reduced x16 on CPU and scale linearly"
    params = {k: v.requires_grad_(True) for k, v in O.init_params(cfg, seed=0).items()}
    o, d = O.synthetic_rays(R, seed)
    batch = O.synthetic_batch(cfg, R, 1 + seed)
    gen = torch.Generator().manual_seed(2 + seed)
    t_rand, u_rand = torch.rand((R, 1), generator=gen), torch.rand((R, 1), generator=gen)

    def step():
        for p in params.values():
            p.grad = None
        out = O.forward(params, cfg, o, d, True, t_rand, u_rand, 1.0)
        sum(O.loss_dict(out, batch, cfg).values()).backward()

    n_warm = 3  # 3 warm-up + up to 10 timed steps (page faults of the 0.9 GB tables, thread pool)
    for _ in range(n_warm):
        step()
    if sync_dir:
        open(os.path.join(sync_dir, f"ready.{seed}"), "w").close()
        t_wait = time.perf_counter()
        while not os.path.exists(os.path.join(sync_dir, "go")) and time.perf_counter() - t_wait < 600:
            time.sleep(0.01)
    t0, n = time.perf_counter(), 0
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 10:
            break
    return {"rays": R, "steps": n, "seconds": el, "warmup": n_warm, "threads": torch.get_num_threads()}


def cpu_baseline(w: dict, budget_s: float):
    """Time the CPU oracle (fwd + bwd of the same step, full-size tables) on a bounded number of rays, on ALL host cores: torch's
    CPU kernels stop scaling (and start thrashing) near 32 threads, so the host's hardware threads are used as cpu_count // 32
    processes of 32 threads (at most 8), each on its own rays, running concurrently; `value` is the sum of their rates."""
    import subprocess
    import tempfile
    ncpu = os.cpu_count() or 1
    threads = min(ncpu, 32)
    procs = max(1, min(8, ncpu // 32))
    spec = json.dumps({k: w[k] for k in ("method", "R", "P", "S", "K", "patch")})
    sync_dir = tempfile.mkdtemp(prefix="snf_cpu_baseline_")
    cmd = lambda i: [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", spec, "--cpu-baseline-seconds", str(budget_s),  # noqa: E731
                     "--cpu-baseline-threads", str(threads), "--cpu-baseline-seed", str(i), "--cpu-baseline-sync", sync_dir]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["HIP_VISIBLE_DEVICES"] = ""  # (the workers never touch the GPU)
    ps = [subprocess.Popen(cmd(i), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env, cwd=ROOT) for i in range(procs)]
    # the workers' timed windows start together: when every live worker is warm (or after 10 minutes) the parent writes `go`
    t_wait = time.perf_counter()
    while time.perf_counter() - t_wait < 600:
        live = [i for i, p_ in enumerate(ps) if p_.poll() is None]
        if all(os.path.exists(os.path.join(sync_dir, f"ready.{i}")) for i in live):
            break
        time.sleep(0.05)
    open(os.path.join(sync_dir, "go"), "w").close()
    res = []
    for p_ in ps:
        out_, _ = p_.communicate(timeout=60 * 30)
        lines = [l for l in out_.splitlines() if l.startswith("{")]
        if p_.returncode == 0 and lines:
            res.append(json.loads(lines[-1]))
    failed = procs - len(res)
    if not res:  # (no worker came back: time one in-process, as before)
        res, procs = [_cpu_baseline_worker(w, budget_s, threads, 0)], 1
    import shutil
    shutil.rmtree(sync_dir, ignore_errors=True)
    try:  # core count and CPU model of the box beside the number
        model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except (OSError, StopIteration):
        model = "unknown"
    value = sum(r["rays"] * w["S"] * r["steps"] / r["seconds"] for r in res)
    used = sum(r["threads"] for r in res)
    r0 = res[0]
    return {"value": value, "unit": "ray-samples/s", "cores": used, "kind": "port", "cpu_model": model, "host_cores": ncpu,
            "processes": len(res), "processes_failed": failed, "threads_per_process": r0["threads"],
            "per_process_ray_samples_per_s": [round(r["rays"] * w["S"] * r["steps"] / r["seconds"], 1) for r in res],
            "sample": f"{len(res)} concurrent processes x {r0['threads']} threads, each {r0['warmup']} warm-up + {r0['steps']} timed fwd+bwd "
                      f"steps of {r0['rays']} rays (the workload's R / 16) x {w['S']} samples (P={w['P']}, K={w['K']}), full-size fp32 "
                      f"tables, no optimizer step, timed windows started together (all processes warm), "
                      f"{max(r['seconds'] for r in res):.1f} s of timed CPU work per process; {used} of {ncpu} "
                      f"hardware threads (one torch process stops scaling near 32 threads: the host is filled with processes instead)"}


def _free(trainer) -> None:
    """Drop a trainer's device memory (arenas, schedule buffers) before the next workload is built."""
    import gc
    trainer._program = None
    trainer.pipeline = None
    trainer.optimizers = None
    gc.collect()
    torch.cuda.empty_cache()


def set_exchange_mode(mode: str) -> None:
    """Multi-rank gradient exchange: 'table_parallel' (default: feature tables sharded by level over the ranks, ZeRO-1
    style reduce-scatter / all-gather for the rest) or 'allreduce' (north_star's wording: full replicas, plain all-reduce of
    every gradient, replicated Adam).  Takes effect for trainers built afterwards."""
    from samnerf_amd import ops
    tp = mode == "table_parallel"
    ops.TABLE_PARALLEL = tp
    os.environ["SNF_TABLE_PARALLEL"] = "1" if tp else "0"
    os.environ["SNF_SHARDED_OPTIMIZER"] = "1" if tp else "0"


def quick_measure(name: str, rank: int, local_rank: int, world: int, steps: int = 20, warmup: int = 6) -> dict:
    """Short run of another BASELINE workload (no serial replay, no breakdown): ms per step and the metric."""
    w = dict(WORKLOADS[name], world=world)
    trainer = build_trainer(w, local_rank, world)
    for i in range(warmup):
        trainer.train_iteration(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        trainer.train_iteration(warmup + i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ms = el / steps * 1e3
    feat = w["K"] * 12288 if w["method"] == "samnerf_distill" else 0
    b_step = 3 * (w["P"] * 320 + w["S"] * 1024 + feat) * w["R"]
    static = trainer._program is not None
    _free(trainer)
    return {"ms_per_step": round(ms, 4), "value": world * w["R"] * w["S"] * steps / el, "unit": "ray-samples/s", "steps": steps,
            "warmup": warmup, "rays_per_gpu": w["R"], "step_frac_of_hbm_peak": round(b_step / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "static_schedule": static}


def render_measure(local_rank: int, res: int = 512, reps: int = 5) -> dict:
    """BASELINE configs[4], render half (samnerf/sam_model.py:337-419): one `res` x `res` camera -> RGB / depth / accumulation, the
    64 x 64 x 256 SAM map (the [256, 256] feature ray grid in 4 x 4 patches through the conv head) and the 32 x 32 x 192 ClipSeg
    map; eval mode, full-size tables, P = 64 / S = 128 / K = 16; the recorded launch schedule (render_program.py)."""
    from samnerf_amd.rays import RayBundle
    tr = build_trainer(WORKLOADS["distill_4096x128"], local_rank, 1)
    model = tr.pipeline.model
    model.eval()
    dev = model.device
    g = torch.Generator(device=dev).manual_seed(0)
    o = torch.rand((res, res, 3), device=dev, generator=g) - 0.5
    d = torch.nn.functional.normalize(torch.randn((res, res, 3), device=dev, generator=g), dim=-1)
    cam = RayBundle(origins=o, directions=d, pixel_area=torch.full((res, res, 1), 1e-6, device=dev),
                    camera_indices=torch.zeros((res, res, 1), dtype=torch.long, device=dev))
    for _ in range(2):
        out = model.get_outputs_for_camera_ray_bundle(cam)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = model.get_outputs_for_camera_ray_bundle(cam)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    fh, fw = out["sam"].shape[:2]
    p = model.config.patch_size
    rays = res * res + fh * fw * p * p + 32 * 32
    S, P = model.config.num_nerf_samples_per_ray, model.config.num_proposal_samples_per_ray[0]
    shapes = {k: list(v.shape) for k, v in out.items() if torch.is_tensor(v)}
    static = model.__dict__.get("_render_prog") is not None
    from samnerf_amd import render_program as _rp
    reuse = bool(static and _rp.REUSE_PASS1)
    K = model.config.num_sam_samples
    # algorithmic HBM bytes, SURVEY 8(d)'s forward figures: P x 320 + S x 1024 B per sampled ray (proposal + field grid gathers),
    # K x 6144 B per feature ray and head (24 levels x 8 corners x 32-B rows).  The reference samples all three passes' rays
    # (`reference_work`); this schedule samples the camera's rays once and feeds the feature passes from pass 1 (`schedule`:
    # the conservative number the fraction is quoted on).
    feat_rays = rays - res * res
    per_ray = P * 320 + S * 1024
    b_ref = rays * per_ray + feat_rays * K * 6144
    b_sched = (res * res if reuse else rays) * per_ray + feat_rays * K * 6144
    _free(tr)
    return {"ms_per_image": round(ms, 3), "rays_per_s": rays / (ms * 1e-3), "ray_samples_per_s": rays * S / (ms * 1e-3),
            "rays_per_image": rays, "samples_per_ray": {"proposal": P, "fine": S}, "outputs": shapes, "images_timed": reps,
            "static_schedule": static, "feature_passes_reuse_pass1": reuse,
            "roofline": {"bound": "hbm", "achieved": round(b_sched / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(b_sched / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "algorithmic_bytes_per_image": b_sched,
                         "reference_work_bytes_per_image": b_ref,
                         "frac_reference_work": round(b_ref / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "traffic": None,
                         "basis": "whole image: requested gather bytes of the three passes (SURVEY 8d forward figures) / "
                                  "wall time per image; per-kernel shares in profiles/*render_kernel_stats.txt"},
            "note": "eval path: RGB pass over every pixel + SAM feature pass over the patch ray grid + "
            "ClipSeg pass over 32 x 32 rays, no_grad, full-size fp32 tables"}


def vit_measure(reps: int = 5) -> dict:
    """BASELINE configs[4], encoder half: SAM ViT-H image encoder forward on one 1024 x 1024 image (random weights of the
    architecture, samnerf/segment_anything/build_sam.py:14-21), every GEMM on the bf16 3-product split (the blocks' four on operands
    split by their producers, csrc/gemm_planes.hip)."""
    from samnerf_amd.image_encoder import build_sam_vit_h_encoder
    enc = build_sam_vit_h_encoder().eval()
    gen = torch.Generator(device="cuda").manual_seed(0)
    for prm in enc.parameters():
        prm.data.copy_(torch.randn(prm.shape, device="cuda", generator=gen) * 0.02)
    enc.reset_weight_cache()  # (the parameters were written through `.data`)
    x = torch.randn((1, 3, 1024, 1024), device="cuda", generator=gen)
    with torch.no_grad():
        for _ in range(2):
            y = enc(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            y = enc(x)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    # multiply-adds of the 32 blocks at 4096 tokens x 1280 channels: qkv 3d^2 + proj d^2 + MLP 8d^2 per token, attention
    # 2 x keys x d per token (14 x 14 windows in 28 blocks, 64 x 64 global in 4), patch embedding and neck
    d_, T = 1280, 4096
    macs = 32 * T * 12 * d_ * d_ + (28 * T * 196 * 2 * d_) + (4 * T * 4096 * 2 * d_) + T * 768 * d_ + T * (d_ * 256 + 2304 * 256)
    flops = 2.0 * macs
    peak = BF16_MATRIX_PEAK_TFLOPS / 3.0
    finite = bool(torch.isfinite(y).all())
    del enc
    torch.cuda.empty_cache()
    return {"ms_per_image": round(ms, 3), "tflops": round(flops / (ms * 1e-3) / 1e12, 1), "peak": round(peak, 1),
            "frac": round(flops / (ms * 1e-3) / 1e12 / peak, 4), "peak_basis": "bf16 dense 2500 TFLOP/s / 3 products per "
            "fp32-equivalent MAC", "flops_per_image": flops, "output": list(y.shape), "finite": finite, "images_timed": reps,
            "note": "ViT-H/16, 32 blocks, 1280 wide, 1024 x 1024 input, windowed + 4 global attention blocks, neck to 256 x 64 x 64"}


def exchange_summary(trainer, w: dict, world: int) -> dict:
    """Bytes one rank SENDS per step in each collective of the active exchange mode -- so that the first real multi-GPU line can be
    read without a profiler (ring all-reduce / reduce-scatter / all-gather move (W-1)/W of the buffer per rank, all-to-all the
    share of the other ranks)."""
    from samnerf_amd import ops as _ops
    opt = trainer.optimizers
    f = (world - 1) / max(world, 1)
    out = {"world": world, "table_parallel": bool(_ops.TABLE_PARALLEL), "sharded_optimizer": bool(opt.sharded)}
    tables_tp = 0
    for k, a in opt.arenas.items():
        tp = sum(t[1] - t[0] for t in opt._tp_tables(k))
        tables_tp += tp
        repl = a.numel - tp
        # replicated part of the arena: reduce-scatter + all-gather (sharded) or one all-reduce -- 2 (W-1)/W x bytes either way
        out[k] = {"replicated_params": int(repl), "table_parallel_params": int(tp), "grad_exchange_bytes": int(2 * f * repl * 4)}
    if tables_tp:
        nk = w["R"] * w["K"]
        heads = 2 if w["method"] == "samnerf_distill" else 0
        out["all_gather_positions_bytes"] = int(f * world * nk * 12)
        out["all_to_all_features_bytes"] = int(heads * 2 * f * nk * 192 * 4)  # encodings forward + their gradient backward
    out["total_bytes"] = int(sum(v["grad_exchange_bytes"] for v in out.values() if isinstance(v, dict))
                             + out.get("all_gather_positions_bytes", 0) + out.get("all_to_all_features_bytes", 0))
    return out


def mfma_util():
    """({entry point: matrix-core busy fraction}, file it came from): the NEWEST committed PMC pass profiles/r*_mfma_util.json
    (tools/mfma_util.sh -> tools/mfma_util.py: SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES per kernel)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_mfma_util.json")))
    for f in reversed(files):
        try:
            return json.load(open(f))["by_entry_point"], os.path.relpath(f, ROOT)
        except (OSError, KeyError, ValueError):
            continue
    return {}, None


def env_overrides() -> dict:
    """Every SNF_* variable set in this process's environment: the library's / schedule's documented switches (README
    "Switches").  A timing switch that skips work must not be reachable unannounced."""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("SNF_")}


class _Control:
    """Control plane of a multi-rank bench run: barriers, the max-over-ranks clock and the per-phase "did every rank get through"
    vote travel over a gloo group on CPU tensors, never over the collective library that is being measured -- so a mode whose
    RCCL exchange throws or hangs on some rank is seen as failed by ALL ranks, and the line is still printed."""

    def __init__(self, multi: bool):
        self.multi = multi
        self.group = None
        if multi:
            import datetime
            import torch.distributed as dist
            self.dist = dist
            to = datetime.timedelta(seconds=float(os.environ.get("SNF_BENCH_CONTROL_TIMEOUT", "900")))
            self.group = dist.new_group(backend="gloo", timeout=to)  # (collective: every rank creates it)

    def barrier(self) -> None:
        if self.multi:
            self.dist.barrier(group=self.group)

    def max(self, x: float) -> float:
        if not self.multi:
            return x
        t = torch.tensor([x], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def vote(self, ok: bool, err=None):
        """-> (every rank ok, ranks that failed, first error text with its rank).  Doubles as a barrier."""
        if not self.multi:
            return ok, ([] if ok else [0]), (None if ok else err)
        got = [None] * self.dist.get_world_size()
        self.dist.all_gather_object(got, None if ok else str(err), group=self.group)
        bad = [r for r, e in enumerate(got) if e is not None]
        first = f"rank {bad[0]}: {got[bad[0]]}" if bad else None
        return not bad, bad, first


class _PhaseHung(Exception):
    pass


def guarded(fn, timeout_s: float, device: int, threaded: bool):
    """Run `fn()` -> (True, result) or (False, 'ExcType: text').  threaded (multi-rank): on a worker thread joined with
    `timeout_s` -- a collective that never completes must not take the line with it; a phase that is still running at the
    deadline is reported as hung (its thread is abandoned: the process leaves through os._exit once the line is out)."""
    if not threaded:
        try:
            return True, fn()
        except Exception as e:  # noqa: BLE001
            return False, f"{type(e).__name__}: {e}"
    import threading
    box = {}

    def run():
        try:
            torch.cuda.set_device(device)  # (the current device is per thread)
            box["result"] = fn()
        except BaseException as e:  # noqa: BLE001
            box["error"] = f"{type(e).__name__}: {e}"

    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(timeout_s)
    if th.is_alive():
        return False, f"_PhaseHung: no completion within {timeout_s:.0f} s (watchdog; SNF_BENCH_MODE_TIMEOUT)"
    if "error" in box:
        return False, box["error"]
    return True, box.get("result")


def _inject(mode: str, rank: int) -> None:
    """SNF_BENCH_INJECT_FAILURE='<mode>:<raise|hang>[:rank]' -- test hook of the fail-soft flow (tests/test_model_gpu.py): the
    named exchange mode throws / never returns on the given rank (default: the last one) inside its timed steps."""
    spec = os.environ.get("SNF_BENCH_INJECT_FAILURE", "")
    if not spec:
        return
    parts = spec.split(":")
    who = int(parts[2]) if len(parts) > 2 else int(os.environ.get("WORLD_SIZE", "1")) - 1
    if parts[0] == mode and rank == who:
        if parts[1] == "raise":
            raise RuntimeError(f"injected failure in exchange mode {mode!r} on rank {rank}")
        while True:
            time.sleep(3600)


def _rccl_log_tail(path, nbytes: int = 1500):
    try:
        with open(path, "rb") as f:
            data = f.read()
        return data[-nbytes:].decode("utf-8", "replace") if data else ""
    except OSError:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="distill_4096x128", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-baseline-seconds", type=float, default=20.0)
    ap.add_argument("--roofline-kernel", default=None, help="kernel key 'entry/tag' to time live (default: auto)")
    ap.add_argument("--other-workloads", default=None,
                    help="comma list of further BASELINE workloads measured briefly and reported under 'other_workloads' "
                         "(default at --gpus 1 with the default workload: the other two; 'none' disables)")
    ap.add_argument("--exchange", default="both", choices=["both", "table_parallel", "allreduce"],
                    help="multi-rank gradient exchange to time (both: north_star's all-reduce first, then the table-parallel "
                         "one, each for --steps steps; the faster one that completed is `value`)")
    ap.add_argument("--steady-steps", type=int, default=100,
                    help="extra steps timed AFTER everything else and reported under 'steady_state' (0 disables)")
    ap.add_argument("--allow-ablation", action="store_true",
                    help="measurement only: accept SNF_ABLATE_SKIP (launches left out, results garbage); the line says so")
    ap.add_argument("--cpu-baseline-worker", default=None, help=argparse.SUPPRESS)   # (internal: one process of cpu_baseline)
    ap.add_argument("--cpu-baseline-threads", type=int, default=32, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-baseline-seed", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-baseline-sync", default="", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        print(json.dumps(_cpu_baseline_worker(json.loads(args.cpu_baseline_worker), args.cpu_baseline_seconds, args.cpu_baseline_threads,
                                              args.cpu_baseline_seed, args.cpu_baseline_sync)))
        return
    w = dict(WORKLOADS[args.workload])
    if os.environ.get("SNF_ABLATE_SKIP") and not args.allow_ablation:
        sys.exit("bench.py: SNF_ABLATE_SKIP is set -- the schedule would skip launches and the number would be invalid. "
                 "Unset it, or pass --allow-ablation for a timing probe (tools/ablate_step.sh); the JSON line then carries "
                 "\"invalid\": true.")

    rccl_log = None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("SNF_FORCE_COLLECTIVES") == "1":
        # RCCL's own warnings into a per-rank file whose tail goes into the line (a first multi-GPU run must explain itself)
        os.environ.setdefault("NCCL_DEBUG", "WARN")
        if "NCCL_DEBUG_FILE" not in os.environ:
            import tempfile
            rccl_log = os.path.join(tempfile.gettempdir(), f"snf_rccl_{os.getpid()}.log")
            os.environ["NCCL_DEBUG_FILE"] = rccl_log
        else:
            rccl_log = os.environ["NCCL_DEBUG_FILE"]

    import samnerf_amd  # noqa: F401
    from samnerf_amd import distributed as D
    from samnerf_amd import ops
    import torch.distributed as dist

    rank, local_rank, world = D.init_distributed()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # SNF_BENCH_DEVICE: ranks sharing one device (tests/test_model_gpu.py runs two gloo ranks on a 1-GPU box)
    local_rank = int(os.environ.get("SNF_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(local_rank)
    if os.environ.get("SNF_BENCH_MAIN_STREAM", "0") == "1":
        # probe: the step's main stream is torch's current stream, by default the legacy NULL stream, against which every
        # BLOCKING stream (hipExtStreamCreateWithCUMask makes only those) synchronises implicitly; 1 = run everything on a stream of our own
        torch.cuda.set_stream(torch.cuda.Stream())
    if os.environ.get("SNF_ADAM_LAUNCH"):  # tuning: "max_blocks,threads,unroll"
        from samnerf_amd import _lib
        assert _lib.load().snf_set_adam_launch(*(int(x) for x in os.environ["SNF_ADAM_LAUNCH"].split(","))) == 0
    w["world"] = world
    multi = dist.is_initialized()  # world > 1 (or SNF_FORCE_COLLECTIVES=1 under torchrun: RCCL paths on one GPU)
    ctl = _Control(multi)
    mode_timeout = float(os.environ.get("SNF_BENCH_MODE_TIMEOUT", "60"))   # watchdog per exchange mode's timed steps ...
    build_timeout = float(os.environ.get("SNF_BENCH_BUILD_TIMEOUT", "600"))  # ... and for building + warming a trainer
    failures = {}
    counts = {"updated": 0}  # steps of the last timed() call whose proposal network received a gradient (ray_samplers.py:586-591)

    def timed(trainer, first_step: int, nsteps: int, mode: str = ""):
        """EXACTLY nsteps train iterations between barrier + synchronize on both sides; max over the ranks."""
        ctl.barrier()
        torch.cuda.synchronize()
        ps = trainer.pipeline.model.proposal_sampler
        counts["updated"] = 0
        t0 = time.perf_counter()
        for i in range(nsteps):
            trainer.train_iteration(first_step + i)
            counts["updated"] += 1 if getattr(ps, "last_updated", True) else 0  # (a host flag: no device access)
        if mode:
            _inject(mode, rank)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    def timed_phase(trainer, first_step: int, nsteps: int, mode: str = ""):
        """(elapsed max over ranks, None) or (None, error): the closing barrier and the clock vote are on the control group."""
        ok, res = guarded(lambda: timed(trainer, first_step, nsteps, mode), mode_timeout + nsteps * 0.5, local_rank, multi)
        all_ok, _, first = ctl.vote(ok, res)  # (the closing barrier: every rank has synchronised its device or given up)
        if not all_ok:
            return None, first
        return ctl.max(res), None

    # ---- warm-up, then the timed region FIRST (this is `value`); every instrumented pass comes after it.
    # Multi-rank: north_star's plain all-reduce is timed first, then the table-parallel exchange, each on a trainer of its own and
    # each fenced (exception or watchdog): the line carries whatever completed plus the error text of what did not.
    exchange_modes = {}
    modes = ["single"]
    if multi:
        modes = ["allreduce", "table_parallel"] if args.exchange == "both" else [args.exchange]
    built = {}
    hung = False
    for mode in modes:
        if hung:
            failures[mode] = "skipped: an earlier mode hung, the communicator is not trusted any more"
            continue

        def build_and_warm(mode=mode):
            if multi:
                set_exchange_mode(mode)
            tr = build_trainer(w, local_rank, world)
            for i in range(args.warmup):
                tr.train_iteration(i)
            torch.cuda.synchronize()
            return tr

        ok, tr = guarded(build_and_warm, build_timeout, local_rank, multi)
        all_ok, bad, first = ctl.vote(ok, tr)
        if not all_ok:
            failures[mode] = "build / warm-up: " + str(first)
            # every rank threw (no watchdog): the same deterministic error everywhere, the ranks are still in step and the next
            # mode may run; anything else leaves ranks at different collectives and nothing more runs on this communicator
            hung = hung or len(bad) < world or "_PhaseHung" in str(first)
            continue
        el, err = timed_phase(tr, args.warmup, args.steps, mode if multi else "")
        if err is not None:
            failures[mode] = err
            hung = True  # a rank threw or hung inside collectives: ranks are out of step, nothing more runs on this communicator
            continue
        st = args.warmup + args.steps
        exchange_modes[mode] = {"ms_per_step": round(el / args.steps * 1e3, 4), "value": world * w["R"] * w["S"] * args.steps / el,
                                "proposal_update_steps": counts["updated"]}
        built[mode] = (tr, st, el)

    def emit(out: dict, code: int) -> None:
        if rank == 0:
            print(json.dumps(out), flush=True)
        if multi and (hung or code):
            os._exit(code)  # abandoned worker threads / a wedged communicator: no orderly teardown to wait for

    def base_line(elapsed, chosen) -> dict:
        R, S, K = w["R"], w["S"], w["K"]
        ms = elapsed / args.steps * 1e3 if elapsed else None
        feat = K * 12288 if w["method"] == "samnerf_distill" else 0
        # B_step = 3 B_f - 2 P g_prop [proposal under no_grad]: the proposal grid's backward bytes only count on the steps
        # where the sampler's `updated` gate let its gradient through (`proposal_update_steps` of the timed region)
        n_upd = exchange_modes.get(chosen, {}).get("proposal_update_steps", args.steps) if chosen else args.steps
        b_step = 3 * (w["P"] * 320 + S * 1024 + feat) * R - 2 * w["P"] * 320 * R * (1.0 - n_upd / max(args.steps, 1))
        prog = getattr(built[chosen][0], "_program", None) if chosen in built else None
        return {
            "metric": "ray-samples/sec (train step, samnerf_distill 256-d feat head)",
            "value": (world * R * S * args.steps / elapsed) if elapsed else None, "unit": "ray-samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "dtype_detail": "fp32 tables, activations, gradients, Adam state and accumulators; products of the layers >= 64 wide on "
                            "v_mfma_f32_32x32x16_bf16 with hi/lo(/mid) bf16 operand splits (3 products per fp32 MAC in the backward and "
                            "the heads, 6 in the field nets' forward), fp32 accumulate (SNF_GEMM_MODE=1)",
            "data": "synthetic",
            "schedule": prog.schedule_info() if prog is not None else None,
            "config": {"workload": f"{w['method']} R={R} rays/GPU x S={S} fine samples, P={w['P']} proposal samples, "
                                   f"K={K} feature samples, patch {w['patch']}, SAM 256-d"
                                   + (" + ClipSeg 192-d heads" if w["method"] == "samnerf_distill" else "")
                                   + ", full-size fp32 tables (T=19), fwd+bwd+RCCL grad mean+fused Adam",
                       "name": args.workload, "rays_per_gpu": R, "parallelism": f"ray-dp{world}"},
            "rays_per_s": (world * R * args.steps / elapsed) if elapsed else None,
            "feature_samples_per_s": (world * R * K * args.steps / elapsed) if elapsed else None,
            "step_algorithmic_GBps": (b_step / (ms * 1e-3) / 1e9) if elapsed else None,
            "step_frac_of_hbm_peak": (b_step / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if elapsed else None,
            "step_algorithmic_bytes": b_step, "proposal_update_steps": n_upd,
            "untimed_steps_before_timed_region": {"warmup": args.warmup},
            "rccl": {"backend": (dist.get_backend() if multi else None), "ranks": world, "collectives_on": bool(multi),
                     "exchange": chosen if multi else None, "exchange_modes_timed": exchange_modes,
                     "exchange_modes_failed": failures, "debug_log_tail": _rccl_log_tail(rccl_log) if rccl_log else None,
                     "env": {k: os.environ.get(k) for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG")}},
            "host": {"env_overrides": env_overrides()},
        }

    if not built:  # nothing completed: still one line, with the reasons
        out = base_line(None, None)
        out["roofline"], out["cpu_baseline"], out["invalid"] = None, None, True
        out["invalid_reason"] = "no exchange mode completed its timed steps: " + json.dumps(failures)
        emit(out, 1)
        sys.exit(1)
    chosen_mode = min(built, key=lambda m: built[m][2])
    for m in list(built):
        if m != chosen_mode:
            _free(built[m][0])
    if multi and not hung:
        set_exchange_mode(chosen_mode)
    trainer, step, elapsed = built[chosen_mode]
    out = base_line(elapsed, chosen_mode)
    if hung:  # a later mode failed inside its collectives: print what was measured and leave
        out["roofline"], out["cpu_baseline"] = None, None
        out["instrumentation_skipped"] = "an exchange mode failed inside its collectives; see rccl.exchange_modes_failed"
        emit(out, 0)
        return

    # ---- everything below is instrumentation AFTER the timed region; a failure in it is reported, never fatal to the line
    def instrument():
        nonlocal step
        res = {}
        # SNF_AUTOTUNE_STREAMS=1: untimed, the trainer times its stream layouts on this device and keeps the faster one
        res["stream_probe"] = trainer.autotune_streams() if os.environ.get("SNF_AUTOTUNE_STREAMS", "0") == "1" else {}
        # A short SERIAL replay (one stream, every C-ABI launch bracketed by HIP events on its stream) measures each kernel's own
        # duration and picks the dominant one; a second short pass of the real (multi-stream) step then times that kernel live.
        n_break = 3
        torch.cuda.synchronize()
        trainer.overlap = False
        presort_side, ops.PRESORT_SIDE_STREAM = ops.PRESORT_SIDE_STREAM, False  # keep the replay on one stream
        wgrad_side, ops.WGRAD_SIDE_STREAM = ops.WGRAD_SIDE_STREAM, False
        ops.enable_kernel_timing("all")
        for i in range(n_break):
            trainer.train_iteration(step)
            step += 1
        breakdown = ops.kernel_timing_summary()
        ops.enable_kernel_timing(None)
        trainer.overlap = True
        ops.PRESORT_SIDE_STREAM = presort_side
        ops.WGRAD_SIDE_STREAM = wgrad_side
        res["breakdown"], res["n_break"] = breakdown, n_break
        per_step = {k: v["total_ms"] / n_break for k, v in breakdown.items()}
        res["per_step"] = per_step

        def model_of(key):
            """(bound, algorithmic units per launch or None, unit).  Adam is modelled per STEP (its launches cover the
            arenas in pieces whose number differs between the serial and the concurrent schedule)."""
            if key == "snf_adam_step":
                return "hbm", adam_bytes(trainer), "GB/s"
            return algorithmic_model(key, w)

        res["model_of"] = model_of
        # "Dominant kernel" is meant at the GPU-kernel level (what rocprofv3 --stats ranks).  A C-ABI entry that is a pipeline of
        # several kernels counts with the share of its largest one: the bucketed hash-grid backward is stage/count/scan/scatter/
        # reduce, k_hg_reduce being ~60 % of it (profiles/*_kernel_stats.csv).
        largest_kernel_share = {"snf_hashgrid_bwd_sorted": 0.6, "snf_hashgrid_bwd_sorted_ex": 0.6,
                                "snf_hashgrid_bwd_presorted": 0.9, "snf_hashgrid_bwd_presorted_adam": 0.95,
                                "snf_hashgrid_bwd_presorted_adam_sp": 0.9,
                                "snf_hashgrid_bwd_presorted_adam_pair": 1.0,
                                "snf_hashgrid_bwd_presorted_adam_fx": 0.95}
        dom = args.roofline_kernel
        if dom is None and per_step:
            modelled = [k for k in per_step if model_of(k)[1]]
            dom = (max(modelled, key=lambda k: per_step[k] * largest_kernel_share.get(k.partition("/")[0], 1.0))
                   if modelled else None)
        res["dom"] = dom
        # the live pass: the concurrent schedule again (a few untimed steps first: the host-bound replay let the clocks come down),
        # the dominant kernel bracketed by HIP events on the stream it is launched on
        n_live = min(args.steps, 20)
        for _ in range(4):
            trainer.train_iteration(step)
            step += 1
        if dom is not None:
            ops.enable_kernel_timing([dom])
        t_live = timed(trainer, step, n_live)
        step += n_live
        res["live"] = ops.kernel_timing_summary() if dom is not None else {}
        res["live_ms_per_step"], res["n_live"] = t_live / n_live * 1e3, n_live
        ops.enable_kernel_timing(None)
        # the same step WITHOUT the exchange + optimizer (report both): gradients just accumulate
        n_fb = min(args.steps, 20)
        trainer.optimizers.enabled = False
        trainer.train_iteration(step)
        res["elapsed_fb"], res["n_fb"] = timed(trainer, step, n_fb), n_fb
        trainer.optimizers.enabled = True
        trainer.optimizers.zero_grad_all()
        if args.steady_steps > 0:  # the plateau on record (the proposal-weight anneal makes the first ~100 steps slower)
            for _ in range(2):
                trainer.train_iteration(step)
                step += 1
            res["steady"] = timed(trainer, step, args.steady_steps)
            step += args.steady_steps
        from samnerf_amd import _lib as _snf_lib
        res["gemm_mode"] = int(_snf_lib.load().snf_get_gemm_mode())
        res["static_schedule"], res["static_off"] = trainer._program is not None, trainer._program_off
        res["exchange_bytes"] = exchange_summary(trainer, w, world) if multi else None
        res["n_arena_slots"] = adam_bytes(trainer)
        return res

    ok, ins = guarded(instrument, build_timeout, local_rank, multi)
    all_ok, _, first = ctl.vote(ok, ins)
    if not all_ok:
        out["roofline"], out["cpu_baseline"] = None, None
        out["instrumentation_failed"] = first
        hung = True
        emit(out, 0 if multi else 1)
        if not multi:
            sys.exit(1)
        return
    if multi:  # (the fwd+bwd-only and steady clocks: max over the ranks like the headline)
        ins["elapsed_fb"] = ctl.max(ins["elapsed_fb"])
        if "steady" in ins:
            ins["steady"] = ctl.max(ins["steady"])

    # ---- the other BASELINE workloads, briefly (N = 1, default workload only): configs[1] and the per-rank load of configs[3]
    others_req = args.other_workloads
    if others_req is None:
        others_req = ("no_distill_4096x128,distill_16384x128" if (world == 1 and args.workload == "distill_4096x128") else "none")
    other_workloads = {}
    if others_req != "none":
        _free(trainer)
        for name in others_req.split(","):
            other_workloads[name] = quick_measure(name, rank, local_rank, world)
        if world == 1 and args.workload == "distill_4096x128" and args.other_workloads is None:
            # BASELINE configs[4] on one GPU: the patch-render eval pass and the SAM ViT-H encoder forward that produces its
            # distillation targets (8 GPUs: the image's rays / the images shard over the ranks, no exchange in either)
            other_workloads["render_512_patch64"] = render_measure(local_rank)
            other_workloads["vit_h_1024"] = vit_measure()

    if rank == 0:
        breakdown, per_step, live, dom = ins["breakdown"], ins["per_step"], ins["live"], ins["dom"]
        model_of, n_break, gemm_mode, n_arena_slots = ins["model_of"], ins["n_break"], ins["gemm_mode"], ins["n_arena_slots"]
        mu, mu_file = mfma_util()

        def roof(key, stat, nsteps, where):
            """achieved = algorithmic units of all timed launches / their summed HIP-event duration."""
            bound, units, unit = model_of(key) if key != "snf_adam_step" else ("hbm", n_arena_slots, "GB/s")
            nl, total_ms = stat["launches"], stat["total_ms"]
            # Adam and the fused backward + Adam: bytes of the actual launches (reported by the launch sites)
            total_units = stat["units"] if stat.get("units", 0) > 0 else units * nl
            achieved = total_units / (total_ms * 1e-3) / (1e9 if bound == "hbm" else 1e12)
            peak, basis = (HBM_PEAK_GBPS, "HBM3E") if bound == "hbm" else mfma_peak(key, gemm_mode)
            o = {"kernel": key, "bound": bound, "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": unit,
                 "frac": round(achieved / peak, 4), "traffic": None, "avg_launch_ms": round(total_ms / max(nl, 1), 4),
                 "launches_timed": nl, "algorithmic_units_per_launch": total_units / max(nl, 1), "measured": where}
            if bound != "hbm":
                o["peak_basis"] = basis
            elif achieved > peak:
                # SURVEY 8(d) counts REQUESTED gather bytes with no cache credit: once the proposal-weight anneal has pulled a ray's samples
                # together (the serial replay runs after the timed region), most of a grid forward's gathers are served by L2 / the
                # Infinity Cache, and the requested-byte rate can exceed what HBM could deliver
                o["note"] = ("requested (algorithmic) bytes exceed the HBM peak: the gathers are served from L2 / Infinity Cache; "
                             "this is a rate of requested bytes, not of HBM traffic")
            mkey = key.replace("_sh/", "/")
            if mkey in mu:  # matrix-core busy cycles / shader busy cycles of this entry point's kernels (rocprofv3 PMC pass)
                o["mfma_busy"] = mu[mkey]
            return o

        def pmc_traffic(key):
            """HBM bytes per launch from the committed PMC passes (profiles/pmc_traffic*.json, made by tools/gpu_record.sh +
            tools/pmc_traffic.py from `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this same command)."""
            import glob
            for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_traffic*.json"))):
                try:
                    t = json.load(open(fn))
                except (OSError, ValueError):
                    continue
                if t.get("workload") == args.workload and t.get("kernel") == key and world == 1:
                    return t["traffic_over_algorithmic"]
            return None

        roofline = None
        if dom is not None and dom in live:
            # contract: the dominant kernel timed live in the concurrent step (HIP events on its own stream).  The step runs three
            # streams, so this duration includes whatever shared the GPU with the kernel (a lower bound on its own roofline).
            roofline = roof(dom, live[dom], ins["n_live"], "HIP events, live pass of the concurrent step (3 streams) right after "
                                                           "the timed region")
            roofline["live_pass_ms_per_step"] = round(ins["live_ms_per_step"], 4)
            ratio = pmc_traffic(dom)  # measured HBM bytes / algorithmic bytes of the same launches in the PMC passes
            roofline["traffic"] = ratio * roofline["algorithmic_units_per_launch"] if ratio else None
            roofline["traffic_source"] = ("profiles/pmc_traffic.json: rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this command, "
                                          f"HBM bytes = {ratio:.4f} x algorithmic bytes, scaled to this run's launch mix"
                                          if ratio else None)
            roofline["serial"] = roof(dom, breakdown[dom], n_break, "HIP events, serial replay (kernel alone on the GPU)")
        others = []
        for k in sorted(per_step, key=lambda kk: -per_step[kk]):
            if k != dom and (k == "snf_adam_step" or algorithmic_model(k, w)[1]) and len(others) < 9:
                others.append(roof(k, breakdown[k], n_break, "HIP events, serial replay"))
                ratio = pmc_traffic(k)
                if ratio:  # (a committed counter pass exists for this kernel too)
                    others[-1]["traffic"] = ratio * others[-1]["algorithmic_units_per_launch"]
                    others[-1]["traffic_source"] = f"profiles/pmc_traffic*.json: HBM bytes = {ratio:.4f} x algorithmic bytes"
        R, S = w["R"], w["S"]
        n_fb, elapsed_fb = ins["n_fb"], ins["elapsed_fb"]
        out["fwd_bwd_only"] = {"value": world * R * S * n_fb / elapsed_fb, "unit": "ray-samples/s",
                               "ms_per_step": elapsed_fb / n_fb * 1e3, "steps": n_fb,
                               "note": "same step without the gradient exchange and the Adam pass"}
        if "steady" in ins:
            out["steady_state"] = {"ms_per_step": ins["steady"] / args.steady_steps * 1e3, "steps": args.steady_steps,
                                   "value": world * R * S * args.steady_steps / ins["steady"],
                                   "note": "timed after everything else (not `value`): the plateau the step reaches once the "
                                           "proposal-weight anneal has sharpened the samples"}
        out["roofline"] = roofline
        out["roofline_other_kernels"] = others
        out["host"].update({"static_schedule": ins["static_schedule"], "static_schedule_off_reason": ins["static_off"],
                            "mfma_busy_source": mu_file})
        out["rccl"]["bytes_per_rank_per_step"] = ins["exchange_bytes"]
        out["other_workloads"] = other_workloads
        out["stream_layout_probe_ms"] = {k: round(v, 3) for k, v in ins["stream_probe"].items()}
        out["serial_step_ms"] = round(sum(per_step.values()), 3)
        out["kernel_ms_per_step_serial"] = {k: round(v, 4) for k, v in sorted(per_step.items(), key=lambda kv: -kv[1])}
        if os.environ.get("SNF_ABLATE_SKIP"):
            out["invalid"] = True
            out["invalid_reason"] = "SNF_ABLATE_SKIP=" + os.environ["SNF_ABLATE_SKIP"] + ": launches left out (timing probe)"
        if world == 1 and args.cpu_baseline_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(w, args.cpu_baseline_seconds)
        else:
            out["cpu_baseline"] = None
    emit(out, 0)
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

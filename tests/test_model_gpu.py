"""GPU: the composed train step (SAMModel on the HIP kernels) against the reference-generated `ministep`
golden vectors and against the CPU oracle.  Bar (BASELINE.json north_star): rendered RGB / feature tensors
within 1e-4 of the reference CPU/PyTorch path on the same rays."""
import copy

import numpy as np
import pytest
import torch

from oracle import samnerf_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


def build_model(P, S, K, patch, log2_T, distill=True, clipseg=True):
    from samnerf_amd import configs, model as M
    mc = copy.deepcopy(configs.method_configs["samnerf_distill" if distill else "samnerf_no_distill"].pipeline.model)
    mc.num_proposal_samples_per_ray = (P,)
    mc.num_nerf_samples_per_ray = S
    mc.num_sam_samples = K
    mc.patch_size = patch
    mc.use_clipseg_feature = clipseg
    mc.log2_hashmap_size = min(19, log2_T)
    mc.hashgrid_sizes = (min(19, log2_T),) * 2
    mc.proposal_net_args_list = [dict(a, log2_hashmap_size=min(17, log2_T)) for a in mc.proposal_net_args_list]
    m = mc.setup(scene_box=M.SceneBox(), num_train_data=2, device="cuda")
    m.train()
    return m


def md(a, b):
    a, b = a.detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max())


def run_step(model, o, d, batch, t_rand, u_rand, anneal):
    from samnerf_amd.rays import RayBundle
    R = o.shape[0]
    rb = RayBundle(origins=o.cuda(), directions=d.cuda(), pixel_area=torch.full((R, 1), 1e-6, device="cuda"),
                   camera_indices=torch.zeros((R, 1), dtype=torch.long, device="cuda"))
    model.proposal_sampler.initial_sampler.jitter_override = t_rand
    model.proposal_sampler.pdf_sampler.jitter_override = u_rand
    model.proposal_sampler.set_anneal(anneal)
    out = model(rb)
    b = {k: v.cuda() for k, v in batch.items()}
    metrics = model.get_metrics_dict(out, b)
    losses = model.get_loss_dict(out, b, metrics)
    return out, losses


def test_ministep_golden(golden, grad_parity):
    from samnerf_amd.interop import load_named_params, named_grads
    g = golden("ministep")
    P, S, K, patch, T = int(g["P"]), int(g["S"]), int(g["K"]), int(g["patch"]), int(g["log2_T"])
    cfg = O.PathConfig(num_proposal_samples=P, num_nerf_samples=S, num_sam_samples=K, patch_size=patch).small(T)
    params = O.init_params(cfg, seed=int(g["seed_params"]), table_scale=float(g["table_scale"]))
    model = build_model(P, S, K, patch, T)
    load_named_params(model, params)
    model.build_arenas()
    o, d = torch.from_numpy(g["origins"]), torch.from_numpy(g["directions"])
    batch = O.synthetic_batch(cfg, int(g["num_rays"]), int(g["seed_batch"]))
    out, losses = run_step(model, o, d, batch, torch.from_numpy(g["t_rand"]), torch.from_numpy(g["u_rand"]),
                           float(g["anneal"]))
    assert md(out["ray_samples_list"][1].spacing_bins, g["sbins_fine"]) <= 1e-5
    assert md(out["weights_list"][0][..., 0], g["w_prop"]) <= 1e-5
    assert md(out["weights_list"][1][..., 0], g["w_fine"]) <= 1e-5
    assert md(out["rgb"], g["rgb"]) <= TOL
    assert md(out["accumulation"], g["accumulation"]) <= TOL
    assert md(out["sam"], g["sam"]) <= TOL
    assert md(out["clipseg"], g["clipseg"]) <= TOL
    rel = np.abs(out["depth"].cpu().numpy() - g["depth"]) / np.abs(g["depth"])
    assert rel.max() <= 1e-4
    rel = np.abs(out["prop_depth_0"].cpu().numpy() - g["prop_depth_0"]) / np.abs(g["prop_depth_0"])
    assert rel.max() <= 1e-4
    for k in ("rgb_loss", "interlevel_loss", "distortion_loss", "sam_loss", "clipseg_loss"):
        assert abs(float(losses[k]) - float(g[k])) <= 1e-5 * max(1.0, abs(float(g[k]))), k
    sum(losses.values()).backward()
    grads = named_grads(model)
    grad_parity(grads, {k: g["grad_" + k] for k in params})


@pytest.mark.parametrize("shape", [(256, 64, 128, 16, 4, 14), (208, 64, 32, 16, 4, 13), (192, 64, 48, 3, 1, 12)])
def test_step_vs_oracle(shape, grad_parity):
    """BASELINE-shaped sample counts (S=128, K=16) at table sizes the CPU oracle handles in seconds: outputs, losses and
    every parameter gradient."""
    from samnerf_amd.interop import load_named_params, named_grads
    R, P, S, K, patch, T = shape
    clipseg = patch > 1
    cfg = O.PathConfig(num_proposal_samples=P, num_nerf_samples=S, num_sam_samples=K, patch_size=patch,
                       use_clipseg=clipseg).small(T)
    params = O.init_params(cfg, seed=3, table_scale=0.05)
    o, d = O.synthetic_rays(R, 5)
    batch = O.synthetic_batch(cfg, R, 6)
    gen = torch.Generator().manual_seed(7)
    t_rand, u_rand = torch.rand((R, 1), generator=gen), torch.rand((R, 1), generator=gen)
    op = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = O.forward(op, cfg, o, d, True, t_rand, u_rand, 0.5)
    model = build_model(P, S, K, patch, T, clipseg=clipseg)
    load_named_params(model, params)
    out, losses = run_step(model, o, d, batch, t_rand, u_rand, 0.5)
    assert md(out["rgb"], ref["rgb"]) <= TOL
    assert md(out["sam"], ref["sam"]) <= TOL
    if clipseg:
        assert md(out["clipseg"], ref["clipseg"]) <= TOL
    ld = O.loss_dict(ref, batch, cfg)
    for k, v in ld.items():
        assert abs(float(losses[k]) - float(v)) <= 1e-4 * max(1.0, abs(float(v))), k
    sum(ld.values()).backward()
    sum(losses.values()).backward()
    grads = named_grads(model)
    grad_parity(grads, {k: v.grad.numpy() for k, v in op.items() if v.grad is not None and k in grads})


def test_eval_mode_and_no_distill():
    from samnerf_amd.interop import load_named_params
    from samnerf_amd.rays import RayBundle
    R, P, S, T = 128, 64, 48, 12
    cfg = O.PathConfig(num_proposal_samples=P, num_nerf_samples=S, distill_sam=False).small(T)
    params = O.init_params(cfg, seed=1, table_scale=0.05)
    o, d = O.synthetic_rays(R, 2)
    ref = O.forward(params, cfg, o, d, False)
    model = build_model(P, S, 3, 1, T, distill=False)
    load_named_params(model, params)
    model.eval()
    rb = RayBundle(origins=o.cuda(), directions=d.cuda(), pixel_area=torch.full((R, 1), 1e-6, device="cuda"),
                   camera_indices=torch.zeros((R, 1), dtype=torch.long, device="cuda"))
    with torch.no_grad():
        out = model(rb)
    assert md(out["rgb"], ref["rgb"]) <= TOL
    assert md(out["accumulation"], ref["accumulation"]) <= TOL
    assert "weights_list" not in out and "sam" not in out


def test_train_iterations_reduce_loss():
    """Trainer plumbing: a few fused-Adam steps on fixed rays must lower the loss and keep everything finite."""
    from samnerf_amd import configs
    tc = copy.deepcopy(configs.method_configs["samnerf_distill"])
    tc.pipeline.datamanager.train_num_rays_per_batch = 512
    mc = tc.pipeline.model
    mc.log2_hashmap_size, mc.hashgrid_sizes = 14, (14, 14)
    mc.proposal_net_args_list = [dict(a, log2_hashmap_size=12) for a in mc.proposal_net_args_list]
    trainer = tc.setup(device="cuda")
    trainer.setup()
    dm = trainer.pipeline.datamanager
    fixed = dm.next_train(0)
    dm.next_train = lambda step: (copy.copy(fixed[0]), fixed[1])
    first = last = None
    for step in range(30):
        loss, ld, _ = trainer.train_iteration(step)
        v = float(loss)
        assert np.isfinite(v)
        first = v if first is None else first
        last = v
    assert last < first
    for a in trainer.optimizers.arenas.values():
        assert float(a.grad.abs().max()) == 0.0  # re-zeroed by the fused Adam pass


@pytest.mark.parametrize("path", ["static", "static-recompute", "eager"])
def test_patch_render_eval_path_vs_oracle(path, monkeypatch):
    """SURVEY 8f rank 1 (samnerf/sam_model.py:337-419): full-image render + SAM / ClipSeg feature maps, eval mode -- through the
    recorded launch schedule (render_program.RenderProgram, the default: the feature passes take their rays' selected samples from
    pass 1), the same schedule sampling the feature rays again as the reference does, and the plugin classes' chunk loop."""
    from samnerf_amd import render_program
    from samnerf_amd.interop import load_named_params
    from samnerf_amd.rays import RayBundle
    static = path != "eager"
    monkeypatch.setenv("SNF_STATIC_RENDER", "1" if static else "0")
    monkeypatch.setattr(render_program, "REUSE_PASS1", path == "static")
    H, W, P, S, K, patch, T = 24, 40, 64, 32, 16, 4, 12
    cfg = O.PathConfig(num_proposal_samples=P, num_nerf_samples=S, num_sam_samples=K, patch_size=patch).small(T)
    params = O.init_params(cfg, seed=5, table_scale=0.05)
    o, d = O.synthetic_rays(H * W, 9)
    o, d = o.view(H, W, 3), d.view(H, W, 3)
    ref = O.render_camera(params, cfg, o, d, chunk=512)
    model = build_model(P, S, K, patch, T)
    model.config.eval_num_rays_per_chunk = 512
    load_named_params(model, params)
    model.eval()
    cam = RayBundle(origins=o.cuda(), directions=d.cuda(), pixel_area=torch.full((H, W, 1), 1e-6, device="cuda"),
                    camera_indices=torch.zeros((H, W, 1), dtype=torch.long, device="cuda"))
    out = model.get_outputs_for_camera_ray_bundle(cam)
    assert (model.__dict__.get("_render_prog") is not None) == static
    fh, fw = O.get_feature_size(H, W)
    assert out["rgb"].shape == (H, W, 3) and out["sam"].shape == (fh, fw, 256) and out["clipseg"].shape == (32, 32, 192)
    assert set(out) == {"rgb", "accumulation", "depth", "prop_depth_0", "sam", "clipseg"}
    assert md(out["rgb"], ref["rgb"]) <= TOL
    assert md(out["accumulation"], ref["accumulation"]) <= TOL
    assert md(out["sam"], ref["sam"]) <= TOL
    assert md(out["clipseg"], ref["clipseg"]) <= TOL
    rel = (out["depth"].cpu() - ref["depth"]).abs() / ref["depth"].abs()
    assert float(rel.max()) <= 1e-4
    fast = model.get_outputs_for_camera_ray_bundle(cam, fast=True)
    assert torch.equal(fast["rgb"], out["rgb"]) and torch.equal(fast["depth"], out["depth"]) and "accumulation" not in fast


def test_render_schedule_at_full_table_size_against_the_oracle():
    """the recorded render schedule AT T = 19 / 17, P = 64 / S = 128 / K = 16 against `O.render_camera`
    (samnerf/sam_model.py:337-419) -- a 16 x 64 camera (1024 pixels in two chunks; its feature ray grid is [16 * 4, 64 * 4] =
    16 384 rays, every pixel taken 16 times by the linspace indices; 32 x 32 ClipSeg rays), the size the oracle renders in
    seconds.  Both forms of the schedule: feature passes fed from pass 1 (default) and sampling their rays again."""
    from samnerf_amd import render_program
    from samnerf_amd.interop import load_named_params
    from samnerf_amd.rays import RayBundle
    H, W, P, S, K, patch = 16, 64, 64, 128, 16, 4
    cfg = O.PathConfig(num_proposal_samples=P, num_nerf_samples=S, num_sam_samples=K, patch_size=patch)
    params = O.init_params(cfg, seed=5, table_scale=0.05)
    o, d = O.synthetic_rays(H * W, 9)
    o, d = o.view(H, W, 3), d.view(H, W, 3)
    ref = O.render_camera(params, cfg, o, d, chunk=4096)
    model = build_model(P, S, K, patch, 19)
    model.config.eval_num_rays_per_chunk = 512
    load_named_params(model, params)
    model.eval()
    cam = RayBundle(origins=o.cuda(), directions=d.cuda(), pixel_area=torch.full((H, W, 1), 1e-6, device="cuda"),
                    camera_indices=torch.zeros((H, W, 1), dtype=torch.long, device="cuda"))
    outs = {}
    for reuse in (True, False):
        render_program.REUSE_PASS1 = reuse
        try:
            outs[reuse] = model.get_outputs_for_camera_ray_bundle(cam)
        finally:
            render_program.REUSE_PASS1 = True
    assert model.__dict__.get("_render_prog") is not None
    out = outs[True]
    assert out["sam"].shape == (16, 64, 256) and out["clipseg"].shape == (32, 32, 192)
    # top-K ties: a ray whose K-th and (K+1)-th weight agree to rounding may select the other sample; with the small
    # random field of this test the rendered feature moves by that sample's share.  Rows outside the bound are counted, not waved.
    for k in ("rgb", "accumulation"):
        assert md(out[k], ref[k]) <= TOL, k
    rel = (out["depth"].cpu() - ref["depth"]).abs() / ref["depth"].abs()
    assert float(rel.max()) <= 1e-4
    for k in ("sam", "clipseg"):
        err = (out[k].cpu().double() - ref[k].double()).abs().amax(-1)
        bad = int((err > TOL).sum())
        assert bad <= max(1, err.numel() // 50), (k, bad, float(err.max()))
        assert float(err.median()) <= 1e-5, k
        # the two forms of the schedule run the same kernels on the same numbers
        assert torch.equal(outs[True][k], outs[False][k]), k
    for k in ("rgb", "accumulation", "depth", "prop_depth_0"):
        assert torch.equal(outs[True][k], outs[False][k]), k


def test_render_schedule_at_the_size_of_config_5():
    """BASELINE config #5, render half, AT SIZE: a 512 x 512 camera -> 64 x 64 x 256 SAM map (the [256, 256] feature ray grid
    in 4 x 4 patches), 32 x 32 x 192 ClipSeg map, full-size tables (T = 19 / 17), P = 64 / S = 128 / K = 16 -- the recorded
    schedule against the plugin classes' chunk loop on the same model (which the small-image test above holds to the oracle):
    same kernels, so the maps agree to the order of the heads' last layer and the mean (1e-6), RGB / depth bit for bit."""
    import os
    from samnerf_amd.rays import RayBundle
    H = W = 512
    model = build_model(64, 128, 16, 4, 19)
    model.eval()
    g = torch.Generator(device="cuda").manual_seed(0)
    o = torch.rand((H, W, 3), device="cuda", generator=g) - 0.5
    d = torch.nn.functional.normalize(torch.randn((H, W, 3), device="cuda", generator=g), dim=-1)
    cam = RayBundle(origins=o, directions=d, pixel_area=torch.full((H, W, 1), 1e-6, device="cuda"),
                    camera_indices=torch.zeros((H, W, 1), dtype=torch.long, device="cuda"))
    from samnerf_amd import render_program
    res = {}
    for static in ("1", "0"):
        os.environ["SNF_STATIC_RENDER"] = static
        try:
            res[static] = model.get_outputs_for_camera_ray_bundle(cam)
        finally:
            os.environ.pop("SNF_STATIC_RENDER")
    a, b = res["1"], res["0"]
    # the default schedule feeds the feature passes from pass 1 (their rays are an index subset of the camera's rays); sampling
    # those rays again, as the reference does, gives the same maps BIT FOR BIT
    render_program.REUSE_PASS1 = False
    try:
        c = model.get_outputs_for_camera_ray_bundle(cam)
    finally:
        render_program.REUSE_PASS1 = True
    for k in a:
        assert torch.equal(a[k], c[k]), k
    assert a["sam"].shape == (64, 64, 256) and a["clipseg"].shape == (32, 32, 192) and a["rgb"].shape == (H, W, 3)
    for k in ("rgb", "accumulation", "depth", "prop_depth_0"):
        assert torch.equal(a[k], b[k]), k
    for k in ("sam", "clipseg"):
        assert torch.isfinite(a[k]).all()
        assert float((a[k] - b[k]).abs().max()) <= 2e-6 * max(1.0, float(b[k].abs().max())), k
    # size-independent properties of the renders: colours in [0, 1], accumulation in [0, 1], depth inside [near, far]
    assert float(a["rgb"].min()) >= 0.0 and float(a["rgb"].max()) <= 1.0
    assert float(a["accumulation"].min()) >= 0.0 and float(a["accumulation"].max()) <= 1.0 + 1e-5
    assert float(a["depth"].min()) >= 0.0 and float(a["depth"].max()) <= model.config.far_plane * (1 + 1e-5)


_RCCL_SCRIPT = r"""
import copy, json, os, sys
sys.path.insert(0, os.environ["SNF_ROOT"])
import torch
import samnerf_amd
from samnerf_amd import configs, distributed as D
rank, local_rank, world = D.init_distributed()
torch.manual_seed(0)  # the samplers draw their jitter from the default generator
tc = copy.deepcopy(configs.method_configs["samnerf_distill"])
tc.pipeline.datamanager.train_num_rays_per_batch = 256
mc = tc.pipeline.model
mc.log2_hashmap_size, mc.hashgrid_sizes = 12, (12, 12)
mc.proposal_net_args_list = [dict(a, log2_hashmap_size=11) for a in mc.proposal_net_args_list]
trainer = tc.setup(device="cuda")
trainer.setup()
dm = trainer.pipeline.datamanager
fixed = dm.next_train(0)
dm.next_train = lambda step: (copy.copy(fixed[0]), fixed[1])
losses = []
for step in range(6):
    loss, ld, _ = trainer.train_iteration(step)
    trainer.synchronize()
    losses.append(float(sum(v.detach() for v in ld.values())))
trainer.optimizers.consolidate_state()
gmax = max(float(a.grad.abs().max()) for a in trainer.optimizers.arenas.values())
# eval render of a small image: sharded over the ranks + gathered when a process group is up
from samnerf_amd.rays import RayBundle
g = torch.Generator().manual_seed(1)
o = (torch.rand((12, 20, 3), generator=g) - 0.5).cuda()
d = torch.nn.functional.normalize(torch.randn((12, 20, 3), generator=g), dim=-1).cuda()
cam = RayBundle(origins=o, directions=d, pixel_area=torch.full((12, 20, 1), 1e-6, device="cuda"),
                camera_indices=torch.zeros((12, 20, 1), dtype=torch.long, device="cuda"))
model = trainer.pipeline.model
model.eval()
model.config.eval_num_rays_per_chunk = 64
img = model.get_outputs_for_camera_ray_bundle(cam)
render = [float(img["rgb"].sum()), float(img["sam"].sum()), float(img["clipseg"].sum())]
print("RESULT " + json.dumps({"losses": losses, "gmax": gmax, "dist": torch.distributed.is_initialized(), "render": render}))
if torch.distributed.is_initialized():
    torch.distributed.destroy_process_group()
"""


def _run_rccl_script(force: bool):
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SNF_ROOT=root, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 300),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    if force:
        env.update(SNF_FORCE_COLLECTIVES="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    else:
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "SNF_FORCE_COLLECTIVES"):
            env.pop(k, None)
    out = subprocess.run([sys.executable, "-c", _RCCL_SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def test_sharded_exchange_runs_on_rccl_single_rank():
    """The data-parallel exchange (in-place reduce_scatter_tensor -> Adam on the shard -> in-place all_gather_into_tensor,
    all_reduce of the remainder, barrier) on the real RCCL backend: a one-rank process group must reproduce the plain run."""
    plain = _run_rccl_script(False)
    forced = _run_rccl_script(True)
    assert forced["dist"] and not plain["dist"]
    assert forced["gmax"] == 0.0 and plain["gmax"] == 0.0
    assert np.allclose(forced["losses"], plain["losses"], rtol=2e-4, atol=1e-6), (forced["losses"], plain["losses"])
    assert np.allclose(forced["render"], plain["render"], rtol=1e-3, atol=1e-4), (forced["render"], plain["render"])
    assert forced["losses"][-1] < forced["losses"][0]


def test_training_trajectory_and_psnr_match_oracle():
    """SURVEY 8(d) 'PSNR vs ref': the oracle (CPU autograd + torch.optim.Adam, eps 1e-15, the configs' learning rates) and
    the HIP trainer start from identical parameters and see identical rays / jitter / targets for N steps; per-step
    losses must agree to 1e-4 relative and the final eval PSNR to 0.01 dB."""
    from samnerf_amd import configs
    from samnerf_amd.interop import load_named_params
    from samnerf_amd.rays import RayBundle
    P, S, K, patch, T, R, NSTEP = 32, 32, 8, 4, 12, 256, 14
    cfg = O.PathConfig(num_proposal_samples=P, num_nerf_samples=S, num_sam_samples=K, patch_size=patch).small(T)
    params = O.init_params(cfg, seed=11, table_scale=0.05)
    tc = copy.deepcopy(configs.method_configs["samnerf_distill"])
    tc.pipeline.datamanager.train_num_rays_per_batch = R
    mc = tc.pipeline.model
    mc.num_proposal_samples_per_ray, mc.num_nerf_samples_per_ray, mc.num_sam_samples = (P,), S, K
    mc.log2_hashmap_size, mc.hashgrid_sizes = min(19, T), (min(19, T),) * 2
    mc.proposal_net_args_list = [dict(a, log2_hashmap_size=min(17, T)) for a in mc.proposal_net_args_list]
    trainer = tc.setup(device="cuda")
    trainer.setup()
    model = trainer.pipeline.model
    load_named_params(model, params)
    # fixed data: analytic colours (so PSNR means something), random feature targets
    o, d = O.synthetic_rays(R, 21)
    batch = O.synthetic_batch(cfg, R, 22)
    batch["image"] = 0.5 + 0.5 * torch.sin(3.0 * d + torch.tensor([0.0, 1.0, 2.0]))
    gen = torch.Generator().manual_seed(23)
    jit = [(torch.rand((R, 1), generator=gen), torch.rand((R, 1), generator=gen)) for _ in range(NSTEP)]
    rb = RayBundle(origins=o.cuda(), directions=d.cuda(), pixel_area=torch.full((R, 1), 1e-6, device="cuda"),
                   camera_indices=torch.zeros((R, 1), dtype=torch.long, device="cuda"))
    dev_batch = {k: v.cuda() for k, v in batch.items()}
    trainer.pipeline.datamanager.next_train = lambda step: (copy.copy(rb), dev_batch)
    # ---- oracle side
    ref = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    group_of = lambda n: ("proposal_networks" if n.startswith("prop") else "conv" if n.startswith("conv") else
                          "sam_field" if n.startswith(("sam", "clipseg")) else "fields")
    ocfg = trainer.config.optimizers
    opts = {g: torch.optim.Adam([v for n, v in ref.items() if group_of(n) == g], lr=ocfg[g]["optimizer"].lr,
                                eps=ocfg[g]["optimizer"].eps, betas=ocfg[g]["optimizer"].betas)
            for g in trainer.optimizers.arenas}
    ref_losses, hip_losses = [], []
    updated_steps = []
    for step in range(NSTEP):
        t_rand, u_rand = jit[step]
        # HIP side first: the sampler decides whether the proposal networks train on this step (every step below 10,
        # every other step after that: ray_samplers.py:566) and the oracle side mirrors the decision
        model.proposal_sampler.initial_sampler.jitter_override = t_rand.cuda()
        model.proposal_sampler.pdf_sampler.jitter_override = u_rand.cuda()
        lrs = {g: trainer.optimizers.lr(g) for g in opts}
        _, ld, _ = trainer.train_iteration(step)
        trainer.synchronize()
        hip_losses.append(float(sum(v.detach() for v in ld.values())))
        updated = model.proposal_sampler.last_updated
        updated_steps.append(updated)
        for g, opt in opts.items():
            for pg in opt.param_groups:
                pg["lr"] = lrs[g]
            opt.zero_grad(set_to_none=True)
        out = O.forward(ref, cfg, o, d, True, t_rand, u_rand, O.proposal_anneal(step), prop_requires_grad=updated)
        loss = sum(O.loss_dict(out, batch, cfg).values())
        loss.backward()
        if trainer.zero_grad_adam:
            # the reference pins torch < 2 (requirements.txt:32): zero_grad() zero-fills, so on a non-update step the proposal
            # parameters carry a ZERO gradient and Adam still steps them; torch >= 2 (this container) leaves grad = None and
            # would skip them -- mirror the pinned behaviour
            for n, v in ref.items():
                if v.grad is None:
                    v.grad = torch.zeros_like(v)
        for opt in opts.values():
            opt.step()
        ref_losses.append(float(loss))
    assert all(updated_steps[:10]) and not all(updated_steps[10:])
    rel = [abs(a - b) / abs(b) for a, b in zip(hip_losses, ref_losses)]
    assert max(rel) <= 1e-4, (hip_losses, ref_losses)
    assert ref_losses[-1] < ref_losses[0]
    # ---- eval PSNR on the training rays (eval mode: deterministic sampling)
    model.proposal_sampler.initial_sampler.jitter_override = None
    model.proposal_sampler.pdf_sampler.jitter_override = None
    model.eval()
    with torch.no_grad():
        rgb_hip = model(copy.copy(rb), get_feature=[])["rgb"].cpu()
        rgb_ref = O.forward({k: v.detach() for k, v in ref.items()}, cfg, o, d, False, get_feature=())["rgb"]
    psnr = lambda x: float(-10.0 * torch.log10(torch.mean((x - batch["image"]) ** 2)))
    assert abs(psnr(rgb_hip) - psnr(rgb_ref)) <= 0.01, (psnr(rgb_hip), psnr(rgb_ref))
    # per-pixel agreement after 14 Adam steps: with eps = 1e-15 an update is ~lr*sign(g) wherever a gradient is tiny, so
    # fp32 summation-order differences move individual table rows by O(lr); measured 6e-3 max on one pixel
    assert md(rgb_hip, rgb_ref) <= 2e-2


def test_adam_skips_only_unreachable_rows():
    """The optimizer visits the reachable rows of the coarse hash levels only: after some steps the other rows still hold
    their initial values with zero gradient and zero Adam moments, and the run agrees with an all-rows run."""
    import subprocess, sys, os, json
    from samnerf_amd import configs
    torch.manual_seed(0)
    tc = copy.deepcopy(configs.method_configs["samnerf_distill"])
    tc.pipeline.datamanager.train_num_rays_per_batch = 256
    mc = tc.pipeline.model
    mc.log2_hashmap_size, mc.hashgrid_sizes = 17, (17, 17)
    trainer = tc.setup(device="cuda")
    trainer.setup()
    opt = trainer.optimizers
    assert opt.skip_unreachable_rows
    init = {k: a.param.clone() for k, a in opt.arenas.items()}
    for step in range(4):
        trainer.train_iteration(step)
    trainer.synchronize()
    torch.cuda.synchronize()
    n_rows_segments = 0
    for k, a in opt.arenas.items():
        visited = torch.zeros(a.numel, dtype=torch.bool, device="cuda")
        for seg in opt._plan(k):
            if seg[0] == "dense":
                visited[seg[1]:seg[2]] = True
            else:
                n_rows_segments += 1
                offs = seg[3].long()
                for f in range(seg[4]):
                    visited[offs + f] = True
        skipped = ~visited
        assert float(a.grad.abs().max()) == 0.0                      # everything re-zeroed, nothing left behind
        assert torch.equal(a.param[skipped], init[k][skipped])       # never written
        assert float(a.exp_avg[skipped].abs().max() if skipped.any() else 0.0) == 0.0
        if k == "sam_field":
            assert int(skipped.sum()) > 0.1 * a.numel                # a real saving even at T = 17 (45 % at T = 19)
            changed = (a.param != init[k]) & visited
            assert int(changed.sum()) > 0
    assert n_rows_segments >= 3  # sam a, clipseg a, field (and the proposal grid)


def test_bench_prints_the_contract_line():
    """bench.py must print exactly one JSON line with the driver's keys plus `roofline` and `cpu_baseline`."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1",
                          "--cpu-baseline-seconds", "2"], env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[-500:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["unit"] == "ray-samples/s" and d["value"] > 0
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["vs_baseline"] is None and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["unit"] in ("GB/s", "TFLOP/s")
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c


def _run_bench(extra, env_extra=None, launcher_ranks=0, timeout=1200):
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "SNF_FORCE_COLLECTIVES")}
    env.update(env_extra or {})
    cmd = [sys.executable]
    if launcher_ranks:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(launcher_ranks), "--master-addr",
                "127.0.0.1", "--master-port", str(29300 + os.getpid() % 90)]
    cmd += [os.path.join(root, "bench.py")] + extra
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-500:]
    return json.loads(lines[0])


@pytest.mark.parametrize("workload,R,K", [("no_distill_4096x128", 4096, 3), ("distill_16384x128", 16384, 16)])
def test_bench_other_baseline_configs_as_the_main_workload(workload, R, K):
    """BASELINE configs[1] (samnerf_no_distill, 4096 x 128) and the per-rank load of configs[3] (samnerf_distill, 16384 x 128)
    through bench.py at their full sample counts: one JSON line, value = R * 128 / t_step, a roofline block, and the train
    step ran as the static launch schedule."""
    d = _run_bench(["--workload", workload, "--steps", "3", "--warmup", "2", "--cpu-baseline-seconds", "0"])
    assert d["config"]["name"] == workload and d["config"]["rays_per_gpu"] == R and d["value"] > 0
    assert abs(d["value"] - R * 128 / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]
    assert abs(d["feature_samples_per_s"] - R * K / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["feature_samples_per_s"]
    assert d["roofline"] is not None and d["roofline"]["frac"] > 0 and d["host"]["static_schedule"] is True
    assert d["other_workloads"] == {}  # only the default workload carries the other two


def test_bench_default_line_carries_the_other_workloads():
    """The driver only ever runs `bench.py --gpus 1`: that line reports configs[1], the per-rank load of configs[3] and both
    halves of configs[4] (the 64 x 64 patch render of a 512 x 512 camera, the ViT-H encoder forward) too."""
    d = _run_bench(["--steps", "3", "--warmup", "1", "--cpu-baseline-seconds", "0"])
    ow = d["other_workloads"]
    assert set(ow) == {"no_distill_4096x128", "distill_16384x128", "render_512_patch64", "vit_h_1024"}
    for name in ("no_distill_4096x128", "distill_16384x128"):
        v = ow[name]
        assert v["value"] > 0 and v["ms_per_step"] > 0 and v["static_schedule"] is True, name
    assert ow["distill_16384x128"]["rays_per_gpu"] == 16384
    r = ow["render_512_patch64"]
    assert r["outputs"]["sam"] == [64, 64, 256] and r["outputs"]["clipseg"] == [32, 32, 192] and r["outputs"]["rgb"] == [512, 512, 3]
    assert r["static_schedule"] is True and 0 < r["ms_per_image"] < 200
    v = ow["vit_h_1024"]
    assert v["finite"] is True and v["output"] == [1, 256, 64, 64] and 0 < v["frac"] < 1 and v["peak"] > 800
    # the matrix kernels are priced against the peak of the instruction they issue
    for o in [d["roofline"]] + d["roofline_other_kernels"]:
        if o["bound"] == "mfma":
            assert o["peak"] in (157.3, 264.6, 416.7, 833.3) and "peak_basis" in o, o
    assert d["rccl"]["ranks"] == 1 and d["rccl"]["collectives_on"] is False


def test_bench_times_both_exchange_modes_under_the_launcher():
    """N > 1 flow with one forced-collective rank on RCCL: the table-parallel exchange AND north_star's plain all-reduce are
    both timed, the faster one is the line's value, and the record shows backend and rank count."""
    d = _run_bench(["--gpus", "1", "--steps", "3", "--warmup", "1", "--cpu-baseline-seconds", "0"],
                   {"SNF_FORCE_COLLECTIVES": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}, launcher_ranks=1)
    r = d["rccl"]
    assert r["backend"] == "nccl" and r["ranks"] == 1 and r["collectives_on"] is True
    assert set(r["exchange_modes_timed"]) == {"table_parallel", "allreduce"} and r["exchange"] in r["exchange_modes_timed"]
    assert all(v["value"] > 0 for v in r["exchange_modes_timed"].values())
    best = max(v["value"] for v in r["exchange_modes_timed"].values())
    assert r["exchange_modes_timed"][r["exchange"]]["value"] == best


def test_no_distill_train_step_at_128_samples_vs_oracle():
    """BASELINE configs[1] sample counts (P = 64, S = 128, K = 3 unused: no feature heads) at oracle-sized tables: outputs and
    every loss term of one samnerf_no_distill train step against the CPU oracle."""
    from samnerf_amd.interop import load_named_params
    R, P, S, T = 256, 64, 128, 13
    cfg = O.PathConfig(num_proposal_samples=P, num_nerf_samples=S, distill_sam=False).small(T)
    params = O.init_params(cfg, seed=4, table_scale=0.05)
    o, d = O.synthetic_rays(R, 8)
    batch = O.synthetic_batch(cfg, R, 9)
    gen = torch.Generator().manual_seed(10)
    t_rand, u_rand = torch.rand((R, 1), generator=gen), torch.rand((R, 1), generator=gen)
    ref = O.forward(params, cfg, o, d, True, t_rand, u_rand, 0.7)
    model = build_model(P, S, 3, 1, T, distill=False)
    load_named_params(model, params)
    out, losses = run_step(model, o, d, batch, t_rand, u_rand, 0.7)
    assert md(out["rgb"], ref["rgb"]) <= TOL and md(out["accumulation"], ref["accumulation"]) <= TOL
    assert md(out["depth"], ref["depth"]) <= 1e-3  # median depth: a bin mid-point far from the camera
    ld = O.loss_dict(ref, batch, cfg)
    assert set(ld) == set(losses)
    for k, v in ld.items():
        assert abs(float(losses[k]) - float(v)) <= 1e-4 * max(1.0, abs(float(v))), k


def test_bench_under_the_multi_gpu_launcher_single_rank():
    """bench.py the way the driver starts it for N > 1 (python -m torch.distributed.run ... bench.py --gpus N), with one rank
    and SNF_FORCE_COLLECTIVES=1: process group on RCCL, table-parallel feature grids (all-gather / all-to-all in forward and
    backward), sharded exchange, max-over-ranks timing -- and still exactly one JSON line."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(SNF_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = str(29900 + os.getpid() % 90)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", port, os.path.join(root, "bench.py"),
                          "--gpus", "1", "--steps", "3", "--warmup", "1", "--cpu-baseline-seconds", "0", "--exchange", "table_parallel"],
                         env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["roofline"] is not None
    assert any(k.endswith("tp") for k in d["kernel_ms_per_step_serial"]), "the table-parallel path did not run"


_TWO_RANK_SCRIPT = r"""
import copy, json, os, sys
sys.path.insert(0, os.environ["SNF_ROOT"])
import torch
import samnerf_amd
from samnerf_amd import configs, distributed as D
from samnerf_amd.rays import RayBundle
MODE, OUT, R, NSTEP = os.environ["SNF_MODE"], os.environ["SNF_OUT"], 256, int(os.environ.get("SNF_NSTEP", "3"))
NRANKS = int(os.environ.get("SNF_NRANKS", "2"))
rank, local_rank, world = D.init_distributed() if MODE == "ranks" else (0, 0, 1)
torch.manual_seed(0)  # the conv head's nn.Conv2d initialisation draws from the default generator
tc = copy.deepcopy(configs.method_configs["samnerf_distill"])
tc.pipeline.datamanager.train_num_rays_per_batch = R
mc = tc.pipeline.model
mc.log2_hashmap_size, mc.hashgrid_sizes = 12, (14, 14)   # T = 14: the coarsest feature level is a reachable-row segment
mc.proposal_net_args_list = [dict(a, log2_hashmap_size=11) for a in mc.proposal_net_args_list]
trainer = tc.setup(device="cuda")
trainer.setup()
model, opt = trainer.pipeline.model, trainer.optimizers


def data(r):
    g = torch.Generator(device="cuda").manual_seed(1000 + r)
    rnd = lambda *s: torch.rand(s, device="cuda", generator=g)
    d = torch.nn.functional.normalize(torch.randn((R, 3), device="cuda", generator=g), dim=-1)
    rb = RayBundle(origins=rnd(R, 3) - 0.5, directions=d, pixel_area=torch.full((R, 1), 1e-6, device="cuda"),
                   camera_indices=torch.zeros((R, 1), dtype=torch.long, device="cuda"))
    batch = {"image": rnd(R, 3), "indices": torch.zeros((R, 3), dtype=torch.long, device="cuda"),
             "sam": torch.randn((R // 16, 256), device="cuda", generator=g),
             "clipseg": torch.randn((R, 192), device="cuda", generator=g)}
    return rb, batch, rnd(NSTEP, R, 1), rnd(NSTEP, R, 1)


def use(r, step, cache={}):
    if r not in cache:
        cache[r] = data(r)
    rb, batch, tj, uj = cache[r]
    trainer.pipeline.datamanager.next_train = lambda s: (copy.copy(rb), batch)
    model.proposal_sampler.initial_sampler.jitter_override = tj[step]
    model.proposal_sampler.pdf_sampler.jitter_override = uj[step]
    torch.manual_seed(4321 + 10 * step + r)  # the random training background comes from the default generator


losses = []
if MODE == "ranks":
    for step in range(NSTEP):
        use(rank, step)
        _, ld, _ = trainer.train_iteration(step)
        trainer.synchronize()
        losses.append(float(sum(v.detach() for v in ld.values())))
    opt.consolidate_state()
else:
    # one process, both ranks' rays per step: gradients accumulate in the arenas, Adam on their mean
    from samnerf_amd.pipeline import BEFORE_TRAIN_ITERATION, AFTER_TRAIN_ITERATION
    for step in range(NSTEP):
        for cb in trainer.callbacks:
            cb.run_callback_at_location(step, BEFORE_TRAIN_ITERATION)
        for r in range(NRANKS):
            use(r, step)
            _, ld, _ = trainer.pipeline.get_train_loss_dict(step=step)
            sum(ld.values()).backward()
            losses.append(float(sum(v.detach() for v in ld.values())))
        for g in opt.arenas:
            opt.optimizer_step(g, grad_scale=1.0 / NRANKS)
        opt.scheduler_step_all(step)
        for cb in trainer.callbacks:
            cb.run_callback_at_location(step, AFTER_TRAIN_ITERATION)
    torch.cuda.synchronize()
if rank == 0:
    torch.save({"losses": losses, **{f"{k}.{n}": getattr(a, n).cpu() for k, a in opt.arenas.items()
                                     for n in ("param", "exp_avg", "exp_avg_sq", "grad")}}, OUT)
if MODE == "ranks":
    torch.save({"losses": losses, "static": trainer._program is not None, "why_not": trainer._program_off}, OUT + f".rank{rank}")
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
"""


def _run_two_rank(tmp_path, nstep: int, static: bool = True, world: int = 2):
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "SNF_FORCE_COLLECTIVES")}
    base.update(SNF_ROOT=root, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29300 + (os.getpid() + nstep + 7 * world) % 300),
                SNF_NSTEP=str(nstep), SNF_NRANKS=str(world))
    ref_out, rk_out = str(tmp_path / f"ref{nstep}.pt"), str(tmp_path / f"ranks{nstep}.pt")
    ref = subprocess.run([sys.executable, "-c", _TWO_RANK_SCRIPT], env=dict(base, SNF_MODE="ref", SNF_OUT=ref_out),
                         capture_output=True, text=True, timeout=600)
    assert ref.returncode == 0, ref.stderr[-3000:]
    procs = [subprocess.Popen([sys.executable, "-c", _TWO_RANK_SCRIPT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(base, SNF_MODE="ranks", SNF_OUT=rk_out, SNF_DIST_BACKEND="gloo", RANK=str(r),
                                       LOCAL_RANK=str(r), WORLD_SIZE=str(world), SNF_STATIC_STEP="1" if static else "0"))
             for r in range(world)]
    outs = [p.communicate(timeout=1200) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    a, b = torch.load(ref_out), torch.load(rk_out)
    for r in range(world):
        rr = torch.load(rk_out + f".rank{r}")
        assert rr["static"] == static, rr["why_not"]  # the path under test is the one that ran
        for step in range(nstep):  # the reference interleaves (step, rank)
            want = a["losses"][world * step + r]
            assert abs(rr["losses"][step] - want) <= 2e-4 * abs(want), (r, step, rr["losses"], a["losses"])
    return a, b


@pytest.mark.parametrize("static", [True, False], ids=["static_schedule", "eager"])
def test_two_ranks_on_one_gpu_match_accumulated_single_process(tmp_path, static):
    """The whole multi-GPU train step with the real kernels and two ranks, as the static launch schedule with its recorded
    collectives (step_program.py, the default) and as the eager autograd path: ray data-parallelism, table-parallel feature grids
    (all-gather / all-to-all in forward and backward), fused backward + Adam on the owned levels, sharded exchange of the
    replicated groups, consolidation.  RCCL refuses two ranks on one device, so the ranks share cuda:0 and the collectives go
    through gloo with host staging (distributed._staged); the arithmetic and the schedule are the product's.  Reference: one
    process that runs both ranks' rays per step, accumulates the gradients and applies Adam to their mean.

    One step pins the GRADIENT: after it exp_avg = 0.1 * mean gradient, and the two sides may differ by fp32 summation order
    only (measured 1e-7 of the largest entry).  Three steps check the schedule (step counts, re-zeroing, owned levels,
    consolidation); there Adam (eps 1e-15, scale-free) turns the rounding noise of near-zero gradients into lr-sized
    differences on a vanishing fraction of the entries, so parameters are compared where the first moment is significant."""
    a, b = _run_two_rank(tmp_path, 1, static)
    for k in [k for k in a if k.endswith(".exp_avg")]:
        scale = float(a[k].abs().max())
        assert scale > 0 and float((a[k] - b[k]).abs().max()) <= 1e-5 * scale, (k, scale, float((a[k] - b[k]).abs().max()))
    a, b = _run_two_rank(tmp_path, 3, static)
    for k in a:
        if k == "losses":
            continue
        d = (a[k] - b[k]).abs()
        if k.endswith(".grad"):
            assert float(b[k].abs().max()) == 0.0, k          # every gradient slot re-zeroed on the multi-rank side
        elif k.endswith(".param"):
            ea = a[k.replace("param", "exp_avg")].abs()
            sig = ea > 1e-2 * float(ea.max())
            info = (k, float(d[sig].max()), float(d.max()), int((d > 1e-5).sum()), d.numel())
            # (the static schedule applies the heads' last layer AFTER the weighted mean over a ray's samples, the accumulating
            # reference process -- eager path -- before it: their bf16-split GEMMs round differently (7e-6 of the largest
            # gradient entry after one step, checked above at 1e-5), and three Adam steps at eps = 1e-15 spread that further)
            loose = static and k.startswith("sam_field")
            b_sig, b_max, b_cnt = (5e-4, 5e-3, 3e-2) if loose else (6e-5, 1.5e-3, 5e-3 if static else 2e-3)
            assert float(d[sig].max()) <= b_sig and float(d.max()) <= b_max and int((d > 1e-5).sum()) <= b_cnt * d.numel(), info
        else:
            assert float(d.max()) <= (1e-2 if static else 2e-3) * max(float(a[k].abs().max()), 1e-30), (k, float(d.max()))


def test_two_ranks_with_the_step_guard_agree_on_the_verdict(tmp_path, monkeypatch):
    """SNF_STEP_GUARD=all: on many ranks the guard's veto word is MAX-all-reduced behind snf_guard_update (three 4-byte collectives per
    step, recorded into the schedule on the streams of their losses).  With finite losses the guarded two-rank step must give the
    unguarded result: exp_avg = 0.1 x mean gradient after one step, every gradient slot re-zeroed."""
    monkeypatch.setenv("SNF_STEP_GUARD", "all")
    a, b = _run_two_rank(tmp_path, 1, True)
    for k in [k for k in a if k.endswith(".exp_avg")]:
        scale = float(a[k].abs().max())
        assert scale > 0 and float((a[k] - b[k]).abs().max()) <= 1e-5 * scale, (k, scale, float((a[k] - b[k]).abs().max()))
    for k in [k for k in b if k.endswith(".grad")]:
        assert float(b[k].abs().max()) == 0.0, k


def test_eight_ranks_on_one_gpu_match_accumulated_single_process(tmp_path):
    """The world size of BASELINE configs[3] / [4]: EIGHT ranks (sharing the box's one GPU, gloo with host staging) through the
    static schedule -- 24 feature slabs / 8 ranks = 3 (grid, level) slabs per rank and head, so a rank's run straddles the two
    grids of a head; all-gather of eight ranks' top-K positions, eight-way all-to-all of encodings and gradients, reduce-scatter /
    all-gather over eight shards with their unaligned remainders -- against one process that accumulates all eight ranks' rays.
    One step pins the gradient mean: exp_avg = 0.1 x mean gradient to fp32 summation order."""
    a, b = _run_two_rank(tmp_path, 1, True, world=8)
    for k in [k for k in a if k.endswith(".exp_avg")]:
        scale = float(a[k].abs().max())
        assert scale > 0 and float((a[k] - b[k]).abs().max()) <= 2e-5 * scale, (k, scale, float((a[k] - b[k]).abs().max()))
    for k in [k for k in b if k.endswith(".grad")]:
        assert float(b[k].abs().max()) == 0.0, k


def test_bench_with_two_ranks_sharing_the_gpu():
    """bench.py --gpus 2 under the driver's launcher, both ranks on the one device of the box (gloo collectives with host staging):
    the N > 1 flow end to end -- table-parallel grids, sharded exchange, barriers, max-over-ranks timing -- prints one JSON line
    from rank 0 with the whole-job value (2 x rays per step)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "SNF_FORCE_COLLECTIVES")}
    env.update(SNF_DIST_BACKEND="gloo", SNF_BENCH_DEVICE="0")
    port = str(29500 + os.getpid() % 90)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", port, os.path.join(root, "bench.py"),
                          "--gpus", "2", "--steps", "3", "--warmup", "1", "--cpu-baseline-seconds", "0", "--exchange", "table_parallel"],
                         env=env, capture_output=True, text=True, timeout=1200, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["parallelism"] == "ray-dp2"
    assert abs(d["value"] - 2 * 4096 * 128 / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]
    assert any(k.endswith("tp") for k in d["kernel_ms_per_step_serial"]), "the table-parallel path did not run"


@pytest.mark.parametrize("how", ["raise", "hang"])
def test_bench_still_prints_its_line_when_an_exchange_mode_fails(how):
    """Fail-soft multi-rank flow: two ranks on the box's one GPU (gloo with host staging), north_star's all-reduce
    mode timed first, then the table-parallel mode with an injected failure on rank 1 -- an exception, or a rank that never returns
    (watchdog).  Rank 0 still prints exactly one JSON line: value from the mode that completed, the failed mode with its error text."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "SNF_FORCE_COLLECTIVES")}
    env.update(SNF_DIST_BACKEND="gloo", SNF_BENCH_DEVICE="0", SNF_BENCH_INJECT_FAILURE=f"table_parallel:{how}:1",
               SNF_BENCH_MODE_TIMEOUT="20")
    port = str(29600 + os.getpid() % 90 + (7 if how == "hang" else 0))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", port, os.path.join(root, "bench.py"),
                          "--gpus", "2", "--steps", "3", "--warmup", "1", "--cpu-baseline-seconds", "0", "--steady-steps", "0"],
                         env=env, capture_output=True, text=True, timeout=1200, cwd=root)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (out.returncode, out.stdout[-800:], out.stderr[-1500:])
    d = json.loads(lines[0])
    r = d["rccl"]
    assert d["n_gpus"] == 2 and d["value"] > 0 and r["exchange"] == "allreduce"
    assert set(r["exchange_modes_timed"]) == {"allreduce"} and "table_parallel" in r["exchange_modes_failed"]
    msg = r["exchange_modes_failed"]["table_parallel"]
    assert ("injected failure" in msg) if how == "raise" else ("_PhaseHung" in msg or "watchdog" in msg), msg
    assert abs(d["value"] - 2 * 4096 * 128 / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]


def test_cu_masked_and_prioritised_streams_run_kernels():
    """snf_stream_create_cu_mask / snf_stream_create_priority / snf_stream_destroy: the handles are HIP streams a kernel runs on (wrapped
    by torch.cuda.ExternalStream) and ordinary events order them against other streams."""
    import ctypes
    from samnerf_amd import _lib, ops
    lib = _lib.load()
    x = torch.zeros((1 << 20,), device="cuda")
    m = torch.zeros_like(x)
    v = torch.zeros_like(x)
    g = torch.ones_like(x)
    for make, arg in ((lib.snf_stream_create_cu_mask, 64), (lib.snf_stream_create_priority, 1), (lib.snf_stream_create_priority, -1)):
        h = ctypes.c_void_p()
        assert make(arg, ctypes.byref(h)) == 0 and h.value
        st = torch.cuda.ExternalStream(h.value)
        st.wait_stream(torch.cuda.current_stream())
        before = x.clone()
        rc = lib.snf_adam_step(x.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), x.numel(), 1e-2, 0.9, 0.999, 1e-15, 1, 1.0, 0,
                               st.cuda_stream)
        assert rc == 0
        torch.cuda.current_stream().wait_stream(st)
        torch.cuda.synchronize()
        assert float((x - before).abs().min()) > 0  # every element stepped
        assert lib.snf_stream_destroy(h.value) == 0
    assert lib.snf_stream_create_cu_mask(64, None) != 0  # (null out pointer: an error code, no crash)


def test_autotune_streams_leaves_the_training_state_untouched():
    """Trainer.autotune_streams times the stream layouts with real train steps and must hand back the exact state it found:
    parameters, gradients, Adam moments, step counters, sampler counters and the random streams."""
    import copy
    from samnerf_amd import configs, ops
    torch.manual_seed(0)
    tc = copy.deepcopy(configs.method_configs["samnerf_distill"])
    tc.pipeline.datamanager.train_num_rays_per_batch = 256
    mc = tc.pipeline.model
    mc.log2_hashmap_size, mc.hashgrid_sizes = 12, (12, 12)
    mc.proposal_net_args_list = [dict(a, log2_hashmap_size=11) for a in mc.proposal_net_args_list]
    trainer = tc.setup(device="cuda")
    trainer.setup()
    for step in range(2):
        trainer.train_iteration(step)
    trainer.synchronize()
    torch.cuda.synchronize()
    opt = trainer.optimizers
    before = {k: [t.clone() for t in (a.param, a.grad, a.exp_avg, a.exp_avg_sq)] for k, a in opt.arenas.items()}
    counts = (dict(opt.step_count), dict(opt.sched_step))
    probe_rng = torch.rand(4, device="cuda").cpu()          # what the default generator yields next ...
    torch.cuda.set_rng_state(torch.cuda.get_rng_state())     # (no-op; keeps the call pattern explicit)
    rng_state = torch.cuda.get_rng_state()
    res = trainer.autotune_streams(steps=2, warm=1)
    assert trainer.presort_host in ("sam", "own")
    assert len(res) == 2 and all(v > 0 for v in res.values())
    assert torch.equal(torch.cuda.get_rng_state(), rng_state)
    assert (dict(opt.step_count), dict(opt.sched_step)) == counts
    for k, a in opt.arenas.items():
        for got, ref in zip((a.param, a.grad, a.exp_avg, a.exp_avg_sq), before[k]):
            assert torch.equal(got, ref), k
    del probe_rng

"""GPU: the static launch schedule of the train step (samnerf_amd/step_program.py) against the eager autograd path.

Both run the same kernels with the same arguments; the only differences allowed are fp32 summation order inside the
atomically accumulated weight gradients.  The eager path is itself checked against the CPU oracle and the golden
`ministep` vectors (tests/test_model_gpu.py), so equality here carries that parity over to the schedule the trainer and
bench.py actually run."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _trainer(method: str, static: bool, R: int, T: int, P: int = 32, S: int = 32, K: int = 8, seed: int = 3):
    from samnerf_amd import configs, tcnn_compat
    tc = copy.deepcopy(configs.method_configs[method])
    tc.pipeline.datamanager.train_num_rays_per_batch = R
    tc.pipeline.datamanager.seed = seed
    mc = tc.pipeline.model
    mc.num_proposal_samples_per_ray, mc.num_nerf_samples_per_ray, mc.num_sam_samples = (P,), S, K
    mc.log2_hashmap_size, mc.hashgrid_sizes = min(19, T), (min(19, T),) * 2
    mc.proposal_net_args_list = [dict(a, log2_hashmap_size=min(17, T)) for a in mc.proposal_net_args_list]
    tcnn_compat.manual_seed(seed)
    torch.manual_seed(seed)  # nn.Conv2d initialises the conv head from the default generator
    trainer = tc.setup(device="cuda")
    trainer.setup()
    trainer.static_step = static
    return trainer


def _run(trainer, steps: int):
    torch.manual_seed(17)  # the samplers' per-ray jitter comes from the default generator
    losses = []
    for step in range(steps):
        loss, ld, md = trainer.train_iteration(step)
        trainer.synchronize()
        losses.append({k: float(v) for k, v in ld.items()} | {"psnr": float(md["psnr"]), "distortion": float(md["distortion"])})
    torch.cuda.synchronize()
    return losses


def _rel_to_max(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("method", ["samnerf_distill", "samnerf_no_distill"])
def test_one_step_gives_the_eager_gradients(method):
    """After ONE step from identical parameters, Adam's first moment is 0.1 x the gradient of every parameter: it must
    match the eager path's to fp32 summation order, for all four groups, and so must every loss term."""
    ref = _trainer(method, False, 512, 13)
    l_ref = _run(ref, 1)
    assert ref._program is None
    new = _trainer(method, True, 512, 13)
    l_new = _run(new, 1)
    assert new._program is not None, new._program_off
    for k, v in l_ref[0].items():
        assert abs(l_new[0][k] - v) <= 1e-5 * max(1.0, abs(v)), (k, l_new[0][k], v)
    for g, a in ref.optimizers.arenas.items():
        b = new.optimizers.arenas[g]
        # (the schedule forms the 64-wide nets' weight gradients inside the chain kernel on the bf16 3-term split, the eager
        # path on the exact-fp32 matrix cores: 1e-5 of the largest entry; everything else is the same kernel on both sides)
        # (sam_field: the schedule renders the hidden activations before the heads' linear last layer -- step_program.
        # MEAN_BEFORE_LAST_LAYER -- the eager path after it: same sums in a different fp32 order, 7e-6 of the largest entry)
        tol = 3e-5 if g == "fields" else (2e-5 if g == "sam_field" else 2e-6)
        assert _rel_to_max(b.exp_avg, a.exp_avg) <= tol, g
        assert _rel_to_max(b.exp_avg_sq, a.exp_avg_sq) <= 2 * tol, g
        assert float(b.grad.abs().max()) == 0.0, g  # re-zeroed by the fused Adam passes
        assert new.optimizers.step_count[g] == ref.optimizers.step_count[g] == 1


def test_loss_multipliers_scale_the_gradients_like_autograd(monkeypatch):
    """interlevel_loss_mult / distortion_loss_mult other than the defaults (nerfacto.py:316-344 multiplies the loss terms, so
    autograd scales their gradients): the schedule's proposal-network and field moments after one step equal the eager path's.
    (the schedule applied interlevel_loss_mult to the reported loss only.)"""
    from samnerf_amd import configs
    base = configs.method_configs["samnerf_no_distill"].pipeline.model
    monkeypatch.setattr(base, "interlevel_loss_mult", 3.0)
    monkeypatch.setattr(base, "distortion_loss_mult", 0.01)
    ref = _trainer("samnerf_no_distill", False, 512, 13)
    l_ref = _run(ref, 1)
    new = _trainer("samnerf_no_distill", True, 512, 13)
    l_new = _run(new, 1)
    assert new._program is not None and ref._program is None
    assert new.pipeline.model.config.interlevel_loss_mult == 3.0
    for k, v in l_ref[0].items():
        assert abs(l_new[0][k] - v) <= 1e-5 * max(1.0, abs(v)), (k, l_new[0][k], v)
    for g, a in ref.optimizers.arenas.items():
        b = new.optimizers.arenas[g]
        assert float(a.exp_avg.abs().max()) > 0
        assert _rel_to_max(b.exp_avg, a.exp_avg) <= (3e-5 if g == "fields" else 2e-6), g


@pytest.mark.parametrize("overlap", [True, False])
def test_trajectory_follows_the_eager_path(overlap):
    """14 steps (the proposal network trains on every step below 10 and on every other step after that, so both variants
    of the schedule run, on both buffer parities): per-step loss terms agree to 1e-4 relative over the first ten steps and
    to 5e-4 after, step counters are equal.

    Why the bound widens: the two paths' gradients differ by <= 3e-5 of the largest entry after one step (the one-step test
    above) and Adam at step counts this low moves every parameter by ~lr whatever the gradient's size, so the difference grows
    by ~1.6x per step -- measured with tools/trajectory_spread.py (profiles/r04_trajectory_spread.txt): eager vs schedule
    8e-7 at step 5, 1.1e-5 at step 8, 7e-5 .. 1.2e-4 at step 13, always in the distortion term; the eager path against ITSELF
    (float atomics in the weight gradients) reaches 5e-5 at step 13.  A stale or clobbered buffer changes a step's samples,
    i.e. the loss terms by percents from the step it happens on."""
    ref = _trainer("samnerf_distill", False, 256, 12)
    new = _trainer("samnerf_distill", True, 256, 12)
    ref.overlap = new.overlap = overlap
    ref.pipeline_steps = new.pipeline_steps = overlap
    l_ref, l_new = _run(ref, 14), _run(new, 14)
    upd = [s for s in range(14)]
    assert new._program is not None and len(new._program.plans) >= 3  # parities x {updated, not updated}
    for step, (a, b) in enumerate(zip(l_ref, l_new)):
        for k, v in a.items():
            tol = 1e-4 if step < 10 else 5e-4
            assert abs(b[k] - v) <= tol * max(1e-3, abs(v)), (step, k, b[k], v)
    assert dict(new.optimizers.step_count) == dict(ref.optimizers.step_count)
    assert dict(new.optimizers.sched_step) == dict(ref.optimizers.sched_step)
    ps_r, ps_n = ref.pipeline.model.proposal_sampler, new.pipeline.model.proposal_sampler
    assert (ps_r._steps_since_update, ps_r._step) == (ps_n._steps_since_update, ps_n._step)
    del upd


def test_schedule_trajectory_is_reproducible():
    """The SCHEDULE against itself over the 14 steps of the trajectory test: the product path's run-to-run spread.  Its table
    gradients are exact fixed-point or fixed-order sums and the chain / full-width weight gradients fixed trees; what is left are the
    float atomics of the tiled weight-gradient kernel on the heads' small layers (csrc/linear_b3.hip: k_gemm_wgrad_b3) and of the
    proposal net (csrc/mlp_tiny.hip), which Adam at eps = 1e-15 amplifies by ~1.6x per step: 2e-6 at step 13 measured
    (profiles/r04_trajectory_spread.txt, column sched-sched) against 5e-5 for the eager path against itself -- the 5e-4 the test above
    allows after step 10 is the eager side's spread, not the schedule's.  Bound here: 2e-5 on every loss term of every step."""
    a = _trainer("samnerf_distill", True, 256, 12)
    b = _trainer("samnerf_distill", True, 256, 12)
    a.overlap = b.overlap = True
    a.pipeline_steps = b.pipeline_steps = True
    la, lb = _run(a, 14), _run(b, 14)
    assert a._program is not None and b._program is not None
    worst = 0.0
    for step, (x, y) in enumerate(zip(la, lb)):
        for k, v in x.items():
            d = abs(y[k] - v) / max(1e-3, abs(v))
            worst = max(worst, d)
            assert d <= 2e-5, (step, k, y[k], v)
    print(f"[reproducibility] worst relative difference of any loss term over 14 steps: {worst:.2e}")


def test_prologue_on_the_side_stream_keeps_the_trajectory(monkeypatch):
    """samnerf_no_distill: the head of step t+1 (sampling, proposal network, resampling, sorts) runs on the side stream under the
    field backward of step t, on parity buffers.  Steps are enqueued back to back (no host synchronisation in between, so the
    overlap is real).  After 2 steps the field group's first moments are BIT-identical to the serial schedule's (its kernels
    are order-independent; the proposal network's tiny-MLP weight gradients use float atomics).  After 16 steps (update and
    non-update steps, both parities) the two schedules differ by no more than the serial schedule differs from itself run to
    run (the atomics' rounding noise, amplified by Adam: 1e-3 of the largest moment, so the bound is 10x the measured noise or
    2 % of the largest moment) -- a clobbered or stale buffer changes the samples of a step, i.e. the gradient by O(1)."""
    from samnerf_amd import step_program

    def run(on: bool, steps: int):
        monkeypatch.setattr(step_program, "XSTEP_PROLOGUE", on)
        tr = _trainer("samnerf_no_distill", True, 1024, 13, P=64, S=64)
        tr.pipeline_steps = True
        torch.manual_seed(17)
        for step in range(steps):
            tr.train_iteration(step)
        tr.synchronize()
        torch.cuda.synchronize()
        prog = tr._program
        _, _, _, pre = prog._side_streams(True)
        assert (pre.stream_id != prog.main.stream_id) == on
        return {g: a.exp_avg.clone() for g, a in tr.optimizers.arenas.items()}, dict(tr.optimizers.step_count)

    (a, ca), (c, cc) = run(False, 2), run(True, 2)
    assert ca == cc
    assert torch.equal(a["fields"], c["fields"])
    # The proposal group: its tiny-MLP weight gradients are float-atomic sums (rounding differs run to run by ~1e-7 relative) of
    # gradients that are ~1e-10 in this miniature (largest first moment 5e-11) while Adam's eps is 1e-15: an element whose two
    # steps' gradients nearly cancel takes an lr-sized step in one direction or the other depending on that rounding, and the
    # second step's moments then differ by up to ~1.3e-3 of the largest one -- between two runs of the SAME schedule as much as
    # between the two schedules (tools/debug_xstep.py: 1 run in 3, always the same two outcomes).  The bound therefore is the
    # serial schedule's own run-to-run spread, with a floor well below what a stale or clobbered buffer does (O(0.1 - 1)).
    a2, _ = run(False, 2)
    spread = max(_rel_to_max(a2["proposal_networks"], a["proposal_networks"]), 5e-3)
    assert _rel_to_max(c["proposal_networks"], a["proposal_networks"]) <= 2.0 * spread
    (a, ca), (b, _), (c, cc) = run(False, 16), run(False, 16), run(True, 16)
    assert ca == cc
    for g in a:
        noise = float((a[g] - b[g]).abs().max())
        diff = float((a[g] - c[g]).abs().max())
        assert diff <= max(10.0 * noise, 2e-2 * float(a[g].abs().max())), (g, diff, noise, float(a[g].abs().max()))


def test_proposal_group_is_stepped_with_zero_gradient_on_non_update_steps():
    """Reference semantics (torch < 2 pinned, requirements.txt:32): zero_grad() zero-fills, so on a step where the proposal
    network gets no gradient Adam still decays its moments and moves the parameters; SNF_TORCH2_NONE_GRADS=1 skips."""
    for static in (True, False):
        tr = _trainer("samnerf_no_distill", static, 256, 12)
        ps = tr.pipeline.model.proposal_sampler
        _run(tr, 10)
        a = tr.optimizers.arenas["proposal_networks"]
        seen = False
        for step in range(10, 14):
            m0, p0, c0 = a.exp_avg.clone(), a.param.clone(), tr.optimizers.step_count["proposal_networks"]
            tr.train_iteration(step)
            torch.cuda.synchronize()
            if not ps.last_updated:
                seen = True
                assert tr.optimizers.step_count["proposal_networks"] == c0 + 1
                nz = m0 != 0
                assert torch.allclose(a.exp_avg[nz], 0.9 * m0[nz], rtol=1e-6, atol=0)  # m <- beta1 m + 0 gradient
                assert not torch.equal(a.param, p0)
        assert seen


def test_bench_workload_full_size_one_step():
    """BASELINE configs[2] at full size (R=4096 x S=128, K=16, T=19): the schedule's first-moment arenas equal the eager
    path's after one step (all 221 M parameters)."""
    import bench
    w = dict(bench.WORKLOADS["distill_4096x128"])
    trs = []
    for static in (False, True):
        torch.manual_seed(1)
        tr = bench.build_trainer(w, 0, 1, seed=1)
        tr.static_step = static
        torch.manual_seed(9)
        tr.train_iteration(0)
        tr.synchronize()
        torch.cuda.synchronize()
        trs.append(tr)
    assert trs[1]._program is not None, trs[1]._program_off
    for g, a in trs[0].optimizers.arenas.items():
        b = trs[1].optimizers.arenas[g]
        assert _rel_to_max(b.exp_avg, a.exp_avg) <= (3e-5 if g == "fields" else (2e-5 if g == "sam_field" else 2e-6)), g


def test_checkpoint_round_trip_continues_the_same_trajectory(tmp_path):
    """Trainer.save_checkpoint / load_checkpoint (trainer.py:379-406 contract: step, pipeline state_dict, optimizer state):
    a run resumed from the file takes exactly the steps the uninterrupted run takes (same fixed batch and jitter)."""
    import copy

    def fixed(tr, batch=None):
        dm = tr.pipeline.datamanager
        batch = batch or dm.next_train(0)
        dm.next_train = lambda step: (copy.copy(batch[0]), batch[1])
        g = torch.Generator(device="cuda").manual_seed(5)
        ps = tr.pipeline.model.proposal_sampler
        ps.initial_sampler.jitter_override = torch.rand((256, 1), device="cuda", generator=g)
        ps.pdf_sampler.jitter_override = torch.rand((256, 1), device="cuda", generator=g)
        return batch

    a = _trainer("samnerf_distill", True, 256, 12)
    batch = fixed(a)
    for step in range(3):
        a.train_iteration(step)
    path = str(tmp_path / "step-000000002.ckpt")
    a.save_checkpoint(path, 2)
    for step in range(3, 5):
        a.train_iteration(step)
    a.synchronize()
    torch.cuda.synchronize()
    b = _trainer("samnerf_distill", True, 256, 12, seed=99)  # different initial parameters: everything must come from the file
    fixed(b, batch)
    assert b.load_checkpoint(path) == 3
    ps_a, ps_b = a.pipeline.model.proposal_sampler, b.pipeline.model.proposal_sampler
    ps_b._step, ps_b._steps_since_update = 2, 0  # (the reference does not checkpoint the sampler's counters either)
    for step in range(3, 5):
        b.train_iteration(step)
    b.synchronize()
    torch.cuda.synchronize()
    assert dict(a.optimizers.step_count) == dict(b.optimizers.step_count)
    for g, arena in a.optimizers.arenas.items():
        other = b.optimizers.arenas[g]
        # (two independent runs of two steps: the float-atomic weight gradients and the float table reduce add in a different
        #  order from run to run, and Adam at eps = 1e-15 turns that into lr-sized differences on near-zero-gradient entries; a
        #  checkpoint that lost its moments or step counts would be off by the whole update)
        d = (other.param - arena.param).abs()
        assert float(d.max()) <= 2e-3 and float((d > 1e-5).double().mean()) <= 1e-3, (g, float(d.max()))
        assert _rel_to_max(other.exp_avg, arena.exp_avg) <= 1e-3, g
    del ps_a


def test_static_schedule_against_the_reference_ministep(golden, grad_parity):
    """The product path itself -- the static launch schedule, level-major head encodings, fixed-point table backward, fused
    chain weight gradients -- on the reference-generated `ministep` fixture (tests/golden/make_golden.py: the reference's own
    torch components composed into one train step): rendered outputs within 1e-4, every loss term, and EVERY parameter
    gradient (criterion: conftest.grad_parity)."""
    import copy
    import numpy as np
    from oracle import samnerf_oracle as O
    from samnerf_amd import configs, tcnn_compat
    from samnerf_amd.interop import load_named_params, named_grads
    from samnerf_amd.rays import RayBundle
    from samnerf_amd.step_program import StepProgram
    g = golden("ministep")
    P, S, K, patch, T, R = (int(g[k]) for k in ("P", "S", "K", "patch", "log2_T", "num_rays"))
    cfg = O.PathConfig(num_proposal_samples=P, num_nerf_samples=S, num_sam_samples=K, patch_size=patch).small(T)
    params = O.init_params(cfg, seed=int(g["seed_params"]), table_scale=float(g["table_scale"]))
    tc = copy.deepcopy(configs.method_configs["samnerf_distill"])
    tc.pipeline.datamanager.train_num_rays_per_batch = R
    mc = tc.pipeline.model
    mc.num_proposal_samples_per_ray, mc.num_nerf_samples_per_ray, mc.num_sam_samples, mc.patch_size = (P,), S, K, patch
    mc.log2_hashmap_size, mc.hashgrid_sizes = min(19, T), (min(19, T),) * 2
    mc.proposal_net_args_list = [dict(a, log2_hashmap_size=min(17, T)) for a in mc.proposal_net_args_list]
    tcnn_compat.manual_seed(0)
    trainer = tc.setup(device="cuda")
    trainer.setup()
    model = trainer.pipeline.model
    load_named_params(model, params)
    o, d = torch.from_numpy(g["origins"]), torch.from_numpy(g["directions"])
    batch = {k: v.cuda() for k, v in O.synthetic_batch(cfg, R, int(g["seed_batch"])).items()}
    rb = RayBundle(origins=o.cuda(), directions=d.cuda(), pixel_area=torch.full((R, 1), 1e-6, device="cuda"),
                   camera_indices=torch.zeros((R, 1), dtype=torch.long, device="cuda"))
    trainer.pipeline.datamanager.next_train = lambda step: (copy.copy(rb), batch)
    ps = model.proposal_sampler
    ps.initial_sampler.jitter_override = torch.from_numpy(g["t_rand"]).cuda()
    ps.pdf_sampler.jitter_override = torch.from_numpy(g["u_rand"]).cuda()
    ps.set_anneal(float(g["anneal"]))
    assert StepProgram.unsupported_reason(trainer) is None
    prog = StepProgram(trainer)
    trainer.optimizers.enabled = False  # gradients stay in the arenas
    loss, ld, md_ = prog.run(0)
    for st in (trainer._side or {}).values():
        torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    out = prog.outputs()

    def md(a, b):
        return float((a.detach().cpu().double().reshape(-1) - torch.as_tensor(np.asarray(b)).double().reshape(-1)).abs().max())

    assert md(prog.bufs["sb1"], g["sbins_fine"]) <= 1e-5
    assert md(prog.bufs["w0"], g["w_prop"]) <= 1e-5 and md(prog.bufs["w1"], g["w_fine"]) <= 1e-5
    assert md(out["rgb"], g["rgb"]) <= 1e-4 and md(out["accumulation"], g["accumulation"]) <= 1e-4
    assert md(out["sam"], g["sam"]) <= 1e-4 and md(out["clipseg"], g["clipseg"]) <= 1e-4
    for key in ("depth", "prop_depth_0"):
        rel = np.abs(out[key].cpu().numpy().reshape(-1) - g[key].reshape(-1)) / np.abs(g[key].reshape(-1))
        assert rel.max() <= 1e-4, key
    for k in ("rgb_loss", "interlevel_loss", "distortion_loss", "sam_loss", "clipseg_loss"):
        assert abs(float(ld[k]) - float(g[k])) <= 1e-5 * max(1.0, abs(float(g[k]))), k
    assert abs(float(md_["psnr"]) + 10.0 * np.log10(float(g["rgb_loss"]))) <= 1e-3
    grads = named_grads(model)
    grad_parity(grads, {k: g["grad_" + k] for k in params})


def test_rendered_head_path_against_the_oracle(grad_parity):
    """R x K = 8192 feature samples: the size from which the schedule renders the heads' hidden activations inside the GEMMs
    (snf_linear_fwd_mean, snf_linear_bwd_*_rows, both grids of a head in one table-backward launch) -- outputs, losses and every
    parameter gradient against the CPU oracle on the same rays (the oracle applies the last layer per sample and renders after
    it, sam_field.py:121-137 / sam_model.py:126-137)."""
    import copy
    from oracle import samnerf_oracle as O
    from samnerf_amd import configs, tcnn_compat
    from samnerf_amd.interop import load_named_params, named_grads
    from samnerf_amd.rays import RayBundle
    from samnerf_amd.step_program import StepProgram
    R, P, S, K, patch, T = 512, 64, 32, 16, 4, 13
    cfg = O.PathConfig(num_proposal_samples=P, num_nerf_samples=S, num_sam_samples=K, patch_size=patch).small(T)
    params = O.init_params(cfg, seed=11, table_scale=0.05)
    o, d = O.synthetic_rays(R, 12)
    batch = O.synthetic_batch(cfg, R, 13)
    gen = torch.Generator().manual_seed(14)
    t_rand, u_rand = torch.rand((R, 1), generator=gen), torch.rand((R, 1), generator=gen)
    op = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = O.forward(op, cfg, o, d, True, t_rand, u_rand, 1.0)
    rl = O.loss_dict(ref, batch, cfg)
    sum(rl.values()).backward()
    tc = copy.deepcopy(configs.method_configs["samnerf_distill"])
    tc.pipeline.datamanager.train_num_rays_per_batch = R
    mc = tc.pipeline.model
    mc.num_proposal_samples_per_ray, mc.num_nerf_samples_per_ray, mc.num_sam_samples, mc.patch_size = (P,), S, K, patch
    mc.log2_hashmap_size, mc.hashgrid_sizes = T, (T, T)
    mc.proposal_net_args_list = [dict(a, log2_hashmap_size=T) for a in mc.proposal_net_args_list]
    tcnn_compat.manual_seed(0)
    trainer = tc.setup(device="cuda")
    trainer.setup()
    model = trainer.pipeline.model
    load_named_params(model, params)
    rb = RayBundle(origins=o.cuda(), directions=d.cuda(), pixel_area=torch.full((R, 1), 1e-6, device="cuda"),
                   camera_indices=torch.zeros((R, 1), dtype=torch.long, device="cuda"))
    dev_batch = {k: v.cuda() for k, v in batch.items()}
    trainer.pipeline.datamanager.next_train = lambda step: (copy.copy(rb), dev_batch)
    ps = model.proposal_sampler
    ps.initial_sampler.jitter_override, ps.pdf_sampler.jitter_override = t_rand.cuda(), u_rand.cuda()
    ps.set_anneal(1.0)
    assert StepProgram.unsupported_reason(trainer) is None
    prog = StepProgram(trainer)
    trainer.optimizers.enabled = False
    loss, ld, _ = prog.run(0)
    for st in (trainer._side or {}).values():
        torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    names = {e[3].partition("/")[0] for plan in prog.plans.values() for e in plan.entries if e[0] == 0}
    assert {"snf_linear_fwd_mean", "snf_linear_bwd_data_rows", "snf_linear_bwd_weight_rows",
            "snf_hashgrid_bwd_presorted_adam_pair"} <= names, names
    out = prog.outputs()
    for k in ("rgb", "sam", "clipseg"):
        assert float((out[k].cpu() - ref[k].detach()).abs().max()) <= 1e-4, k
    for k, v in rl.items():
        assert abs(float(ld[k]) - float(v)) <= 1e-4 * max(1.0, abs(float(v))), k
    grads = named_grads(model)
    grad_parity(grads, {k: v.grad.numpy() for k, v in op.items() if v.grad is not None and k in grads})


def _snapshot(trainer):
    return {g: (a.param.clone(), a.exp_avg.clone(), a.exp_avg_sq.clone()) for g, a in trainer.optimizers.arenas.items()}


def _head_slices(trainer):
    """{head: (lo, hi)} element ranges of the two heads inside the `sam_field` arena."""
    arena = trainer.optimizers.arenas["sam_field"]
    names = list(arena.offsets)
    out = {}
    for h, (lo_i, hi_i) in trainer._head_param_ranges().items():
        out[h] = (arena.offsets[names[lo_i]][0], arena.offsets[names[hi_i]][0] if hi_i < len(names) else arena.numel)
    return out


def test_non_finite_head_loss_vetoes_the_optimizer_step_of_that_head():
    """trainer.py:419-437 / optimizers.py:138-149: GradScaler.step does not step an optimizer whose gradients hold an inf / NaN.
    Here Adam is fused into the table backward, so the veto is a device record read by the optimizer-side kernels (snf_step_guard).
    Every SAM target row NaN at step 2 (nanmean of no rows = NaN): parameters AND moments of the SAM head and of the conv head keep
    their bits, everything else steps; the next step steps the SAM head again and the vetoed step does not count."""
    tr = _trainer("samnerf_distill", True, 512, 13)
    dm = tr.pipeline.datamanager
    orig, poison = dm.next_train, {"step": -1}

    def next_train(step):
        rb, batch = orig(step)
        if step == poison["step"]:
            batch = dict(batch)
            batch["sam"] = torch.full_like(batch["sam"], float("nan"))
        return rb, batch

    dm.next_train = next_train
    _run(tr, 2)
    assert tr._program is not None, tr._program_off
    before = _snapshot(tr)
    poison["step"] = 2
    torch.manual_seed(5)
    tr.train_iteration(2)
    tr.synchronize()
    torch.cuda.synchronize()
    after = _snapshot(tr)
    rep = tr._program.guard_report()
    assert rep["sam"] == {"veto": 1, "skipped": 0} and rep["clipseg"]["veto"] == 0 and rep["nerf"]["veto"] == 0, rep
    sl = _head_slices(tr)
    lo, hi = sl["sam"]
    for i in range(3):  # param, exp_avg, exp_avg_sq of the vetoed head and of the conv head: bit-identical
        assert torch.equal(after["sam_field"][i][lo:hi], before["sam_field"][i][lo:hi]), i
        assert torch.equal(after["conv"][i], before["conv"][i]), i
    lo, hi = sl["clipseg"]
    assert not torch.equal(after["sam_field"][0][lo:hi], before["sam_field"][0][lo:hi])
    assert not torch.equal(after["fields"][0], before["fields"][0])
    assert not torch.equal(after["proposal_networks"][0], before["proposal_networks"][0])
    for g, a in tr.optimizers.arenas.items():
        assert bool(torch.isfinite(a.param).all()) and bool(torch.isfinite(a.exp_avg).all()), g
        assert float(a.grad.abs().max()) == 0.0, g  # the vetoed step's (NaN) gradients were cleared all the same
    # the next step is a normal one: the head moves again, the record has counted the vetoed step
    tr.train_iteration(3)
    tr.synchronize()
    torch.cuda.synchronize()
    lo, hi = sl["sam"]
    assert not torch.equal(tr.optimizers.arenas["sam_field"].param[lo:hi], after["sam_field"][0][lo:hi])
    assert not torch.equal(tr.optimizers.arenas["conv"].param, after["conv"][0])
    rep = tr._program.guard_report()
    assert rep["sam"] == {"veto": 0, "skipped": 1}, rep
    assert bool(torch.isfinite(tr.optimizers.arenas["sam_field"].param).all())
    tr._program.fold_guards()
    assert tr.optimizers.step_count["conv"] == 3 and tr.optimizers.step_count["fields"] == 4
    assert tr._program.guard_report()["sam"] == {"veto": 0, "skipped": 0}


def test_one_inf_weight_vetoes_the_nerf_groups():
    """An inf in the colour net's first layer does NOT show in the loss (the neuron's pre-activation is NaN and fmaxf(NaN, 0) = 0), but
    the gradients behind it are NaN (0 * inf) -- GradScaler would find them; here snf_guard_scan finds the weight.  `fields` and
    `proposal_networks` keep parameters and moments (the table of 16.8 M parameters included: its Adam runs inside the backward
    kernel); once the weight is finite again the groups step and the vetoed step has not counted."""
    tr = _trainer("samnerf_no_distill", True, 512, 13)
    _run(tr, 2)
    w = tr.pipeline.model.field.mlp_head.weights()[0]
    keep = w.detach().clone()
    w.data.view(-1)[7] = float("inf")
    before = _snapshot(tr)
    torch.manual_seed(5)
    tr.train_iteration(2)
    tr.synchronize()
    torch.cuda.synchronize()
    after = _snapshot(tr)
    assert tr._program.guard_report()["nerf"] == {"veto": 1, "skipped": 0}
    for g in ("fields", "proposal_networks"):
        for i in range(3):
            assert torch.equal(after[g][i], before[g][i]), (g, i)
        assert float(tr.optimizers.arenas[g].grad.abs().nan_to_num(1.0).max()) == 0.0, g
    w.data.copy_(keep)
    tr.train_iteration(3)
    tr.synchronize()
    torch.cuda.synchronize()
    assert tr._program.guard_report()["nerf"] == {"veto": 0, "skipped": 1}
    assert not torch.equal(tr.optimizers.arenas["fields"].param, after["fields"][0])
    assert bool(torch.isfinite(tr.optimizers.arenas["fields"].param).all())


def test_guarded_adam_skips_and_recounts():
    """snf_adam_step under a bound guard: veto -> p / m / v untouched and g cleared; skipped = 1 at step 5 -> the update of an
    unguarded step 4 (bias corrections from step - skipped); an all-zero record -> bit-identical to an unguarded launch."""
    from samnerf_amd import _lib
    lib = _lib.load()
    n = 4096 + 3
    gen = torch.Generator(device="cuda").manual_seed(1)
    p0, g0 = torch.randn(n, device="cuda", generator=gen), torch.randn(n, device="cuda", generator=gen)
    m0, v0 = 0.1 * torch.randn(n, device="cuda", generator=gen), torch.rand(n, device="cuda", generator=gen)

    def step(t, guard):
        p, g, m, v = p0.clone(), g0.clone(), m0.clone(), v0.clone()
        st = torch.cuda.current_stream().cuda_stream
        lib.snf_step_guard(guard.data_ptr() if guard is not None else None)
        rc = lib.snf_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-2, 0.9, 0.999, 1e-15, t, 1.0, 1, st)
        lib.snf_step_guard(None)
        assert rc == 0
        torch.cuda.synchronize()
        return p, g, m, v

    plain5, plain4 = step(5, None), step(4, None)
    clear = step(5, torch.zeros(2, dtype=torch.int32, device="cuda"))
    for a, b in zip(clear, plain5):
        assert torch.equal(a, b)
    veto = step(5, torch.tensor([1, 0], dtype=torch.int32, device="cuda"))
    assert torch.equal(veto[0], p0) and torch.equal(veto[2], m0) and torch.equal(veto[3], v0) and float(veto[1].abs().max()) == 0.0
    late = step(5, torch.tensor([0, 1], dtype=torch.int32, device="cuda"))
    assert torch.equal(late[2], plain4[2]) and torch.equal(late[3], plain4[3])
    # (the device forms the bias corrections in fp32 where the host uses double: a few ulp of the parameter)
    assert float((late[0] - plain4[0]).abs().max()) <= 1e-6 * float(p0.abs().max())
    assert float((late[0] - plain5[0]).abs().max()) > 1e-4 * float((plain5[0] - p0).abs().max())
    # snf_guard_update: commit, then judge
    rec = torch.tensor([1, 2], dtype=torch.int32, device="cuda")
    vals = torch.tensor([0.5, float("inf")], device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert lib.snf_guard_update(vals.data_ptr(), 2, rec.data_ptr(), st) == 0
    assert rec.cpu().tolist() == [1, 3]
    assert lib.snf_guard_update(vals.data_ptr(), 1, rec.data_ptr(), st) == 0
    assert rec.cpu().tolist() == [0, 4]
    big = torch.zeros(1 << 20, device="cuda")
    assert lib.snf_guard_scan(big.data_ptr(), big.numel(), rec.data_ptr(), st) == 0
    assert rec.cpu().tolist() == [0, 4]
    big[777777] = float("nan")
    assert lib.snf_guard_scan(big.data_ptr(), big.numel(), rec.data_ptr(), st) == 0
    assert rec.cpu().tolist() == [1, 4]

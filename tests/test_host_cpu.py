"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol of include/samnerf_hip.h, the host
logic (configs, parameter groups, arenas, schedules) mirrors the reference, the product path refuses CPU tensors,
and the data-parallel gradient mean works over gloo with world_size 2."""
import copy
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import samnerf_amd  # noqa: F401
    from samnerf_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "samnerf_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(snf_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 25
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in samnerf_hip.h but not exported"
    # and every symbol bound through ctypes is declared in the header
    for name in _lib.SIGNATURES:
        assert name in declared
    assert lib.snf_version() >= 100
    # records (8 N per level, 16 bytes each: room for x-pair records) | tile histograms + offsets | bucket starts | 64-word scratch of
    # the fixed-point reduce | staged gradients
    assert lib.snf_hashgrid_bwd_workspace_bytes(65536, 12, 19) == 4 * (12 * 8 * 65536 * 4 + 2 * 12 * 64 * 256 + 3084 + 64
                                                                        + 12 * 65536 * 8)


def test_bad_arguments_are_rejected_without_a_gpu():
    import samnerf_amd  # noqa: F401
    from samnerf_amd import _lib
    lib = _lib.load()
    rc = lib.snf_topk_sharpen(None, 4, 300, 16, 10.0, None, None, None)
    assert rc == -1 and b"null pointer" in lib.snf_last_error()


def test_ops_refuse_cpu_tensors():
    import samnerf_amd.ops as ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.sample_spacing(torch.zeros(4), torch.ones(4), 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.linear(torch.zeros(4, 8), torch.zeros(3, 8))


def test_method_configs_match_reference_values():
    from samnerf_amd import configs
    assert set(configs.method_configs) == {"samnerf_no_distill", "samnerf_distill"}
    d = configs.method_configs["samnerf_distill"]
    m = d.pipeline.model
    assert (m.num_proposal_samples_per_ray, m.num_nerf_samples_per_ray, m.num_sam_samples) == ((64,), 32, 16)
    assert m.patch_size == 4 and m.hidden_layers == 1 and m.use_clipseg_feature and m.distill_sam
    assert m.hashgrid_layers == (12, 12) and m.hashgrid_resolutions == ((16, 128), (128, 512))
    assert m.hashgrid_sizes == (19, 19) and m.sharpening_temperature == 10.0
    assert d.pipeline.datamanager.train_num_rays_per_batch == 16384 and d.max_num_iterations == 10000
    assert set(d.optimizers) == {"proposal_networks", "fields", "conv", "sam_field"}
    assert d.optimizers["sam_field"]["optimizer"].lr == 5e-4 and d.optimizers["fields"]["optimizer"].eps == 1e-15
    n = configs.method_configs["samnerf_no_distill"]
    assert n.pipeline.model.num_sam_samples == 3 and not n.pipeline.model.distill_sam
    assert set(n.optimizers) == {"proposal_networks", "fields"} and n.max_num_iterations == 30000


def _tiny_model(distill=True):
    from samnerf_amd import configs, model
    mc = copy.deepcopy(configs.method_configs["samnerf_distill" if distill else "samnerf_no_distill"].pipeline.model)
    mc.log2_hashmap_size, mc.hashgrid_sizes = 8, (8, 8)
    mc.proposal_net_args_list = [dict(a, log2_hashmap_size=8) for a in mc.proposal_net_args_list]
    return mc.setup(scene_box=model.SceneBox(), num_train_data=2, device="cpu")


def test_param_groups_state_dict_and_arenas():
    from oracle import samnerf_oracle as O
    m = _tiny_model()
    groups = m.get_param_groups()
    assert list(groups) == ["proposal_networks", "fields", "sam_field", "conv"]
    cfg = O.PathConfig().small(8)
    ref = O.init_params(cfg)
    count = lambda keys: sum(ref[k].numel() for k in ref if k.startswith(keys))  # noqa: E731
    assert sum(p.numel() for p in groups["proposal_networks"]) == count(("prop_",))
    assert sum(p.numel() for p in groups["fields"]) == count(("field_", "base_", "head_"))
    assert sum(p.numel() for p in groups["sam_field"]) == count(("sam_", "clipseg_"))
    assert sum(p.numel() for p in groups["conv"]) == count(("conv",))
    sd = m.state_dict()
    assert all(isinstance(v, torch.Tensor) for v in sd.values())
    before = {k: v.clone() for k, v in sd.items()}
    arenas = m.build_arenas()
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k]), k  # re-homing keeps values
    for g, a in arenas.items():
        for p in groups[g]:
            off = p.data_ptr() - a.param.data_ptr()
            assert 0 <= off < a.nbytes() and off % 256 == 0  # 256-B slots inside the arena
            assert p.main_grad.data_ptr() - a.grad.data_ptr() == off
    # full-size parameter count of the distill method (~220.8 M)
    full = O.PathConfig()
    n = (full.prop_grid.rows * 2 + full.field_grid.rows * 2 + 4 * full.feat_grids[0].rows * 8
         + 176 + 3072 + 6272 + 114688 + 98304 + 1180160)
    assert abs(n - 220.8e6) < 0.2e6


def test_hash_scalings_match_reference_vectors():
    from oracle import samnerf_oracle as O
    from samnerf_amd.tcnn_compat import hash_scalings
    for (L, mn, mx) in [(5, 16, 128), (16, 16, 2048), (12, 16, 128), (12, 128, 512)]:
        g = np.exp((np.log(mx) - np.log(mn)) / (L - 1))
        assert torch.equal(hash_scalings(L, mn, float(g)), O.hash_scalings(L, mn, mx))


def test_schedules():
    from oracle import samnerf_oracle as O
    from samnerf_amd.engine import ExponentialDecaySchedulerConfig
    s = ExponentialDecaySchedulerConfig(lr_final=0.0005, max_steps=10000)
    assert s.lr_at(0, 1e-2) == pytest.approx(1e-2)
    assert s.lr_at(10000, 1e-2) == pytest.approx(5e-4)
    assert s.lr_at(5000, 1e-2) == pytest.approx(np.sqrt(1e-2 * 5e-4))
    assert s.lr_at(20000, 1e-2) == pytest.approx(5e-4)
    m = _tiny_model(False)
    cbs = m.get_training_callbacks()
    cbs[0].run_callback_at_location(200, "before_train_iteration")
    assert m.proposal_sampler._anneal == pytest.approx(O.proposal_anneal(200))
    for step in range(12):
        cbs[1].run_callback_at_location(step, "after_train_iteration")
    assert m.proposal_sampler._step == 11 and m.proposal_sampler._steps_since_update == 12


def test_fill_uniform_reference_is_deterministic_and_uniform():
    from samnerf_amd.arena import fill_uniform_reference
    a = fill_uniform_reference(100000, 7, -1e-3, 1e-3)
    b = fill_uniform_reference(100000, 7, -1e-3, 1e-3)
    assert np.array_equal(a, b) and a.dtype == np.float32
    assert a.min() >= -1e-3 and a.max() <= 1e-3 and abs(a.mean()) < 1e-5
    assert not np.array_equal(a, fill_uniform_reference(100000, 8, -1e-3, 1e-3))


def _dp_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import samnerf_amd  # noqa: F401
    from samnerf_amd import distributed as D
    from samnerf_amd.arena import ParamGroupArena
    D.init_distributed(backend="gloo")
    arenas = [ParamGroupArena("a", [("0", (5, 7)), ("1", (33,))], "cpu"), ParamGroupArena("b", [("0", (130,))], "cpu")]
    for i, a in enumerate(arenas):
        a.param.fill_(float(rank + 1))  # replicas disagree before the broadcast
        a.grad.copy_(torch.arange(a.numel, dtype=torch.float32) * (rank + 1) + i)
    D.broadcast_parameters([a.param for a in arenas])
    D.allreduce_gradients([a.grad for a in arenas])
    ok = all(float(a.param.min()) == 1.0 == float(a.param.max()) for a in arenas)
    for i, a in enumerate(arenas):
        expect = torch.arange(a.numel, dtype=torch.float32) * sum(range(1, world + 1)) + i * world
        ok = ok and torch.equal(a.grad, expect)
    # the mean the fused Adam applies: grad_scale = 1/world
    ok = ok and D.world_size() == world
    # the step guard's verdict word (step_program._guard_update on many ranks): MAX over the ranks, in place, on the veto word only
    rec = torch.tensor([1 if rank == world - 1 else 0, 3], dtype=torch.int32)
    D._all_reduce_max(rec[0:1])
    ok = ok and rec.tolist() == [1, 3]
    q.put((rank, bool(ok)))
    torch.distributed.destroy_process_group()


def test_data_parallel_gradient_mean_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def _adam_ref(p, g, m, v, lr, b1, b2, eps, t, scale):
    """torch.optim.Adam's update on flat tensors (the fused kernel's contract), gradient pre-scaled by `scale`."""
    g = g * scale
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    denom = (v.sqrt() / np.sqrt(1 - b2 ** t)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / (1 - b1 ** t))


def _sharded_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import samnerf_amd  # noqa: F401
    from samnerf_amd import distributed as D
    D.init_distributed(backend="gloo")
    ok = True
    for n in (1000, 64 * world, 64 * world * 5 + 17, 30):  # bulk + remainder, exact multiple, remainder only
        gen = torch.Generator().manual_seed(n)
        p0 = torch.randn(n, generator=gen)
        grads = [torch.randn(n, generator=gen) for _ in range(world)]  # what each rank's backward produced
        # reference: replicated Adam on the mean gradient, three steps
        pr, mr, vr = p0.clone(), torch.zeros(n), torch.zeros(n)
        p, g, m, v = p0.clone(), torch.zeros(n), torch.zeros(n), torch.zeros(n)
        for t in (1, 2, 3):
            _adam_ref(pr, sum(grads) * (0.5 + t), mr, vr, 1e-2, 0.9, 0.999, 1e-15, t, 1.0 / world)
            g += grads[rank] * (0.5 + t)  # backward accumulates into the (zeroed) gradient slice

            def step_fn(lo, hi, t=t):
                _adam_ref(p[lo:hi], g[lo:hi], m[lo:hi], v[lo:hi], 1e-2, 0.9, 0.999, 1e-15, t, 1.0 / world)
                g[lo:hi].zero_()

            D.sharded_step(p, g, step_fn)
            ok = ok and bool(torch.count_nonzero(g) == 0)          # the whole gradient slice is re-zeroed
            ok = ok and torch.allclose(p, pr, rtol=1e-5, atol=1e-6)  # every rank holds the full updated parameters
        # moments live only in the own shard (+ remainder) until gathered
        D.gather_sharded_state(m)
        D.gather_sharded_state(v)
        ok = ok and torch.allclose(m, mr, rtol=1e-5, atol=1e-6) and torch.allclose(v, vr, rtol=1e-5, atol=1e-6)
    # reachable-row exchange: only the listed rows of a [rows, F] gradient view are summed, the rest is left alone
    g2 = torch.arange(40 * 8, dtype=torch.float32).view(40, 8) * (rank + 1)
    idx = torch.tensor([1, 5, 6, 39])
    before = g2.clone()
    D.exchange_rows(g2, idx)
    expect = before.clone()
    expect[idx] = torch.arange(40 * 8, dtype=torch.float32).view(40, 8)[idx] * sum(range(1, world + 1))
    ok = ok and torch.equal(g2, expect)
    # eval sharding helpers: contiguous shares in whole granules; ragged all-gather restores row order
    for n, gran in ((1000, 1), (160, 16), (16, 16)):
        lo, hi = D.split_range(n, gran)
        ok = ok and lo % gran == 0 and (hi % gran == 0 or hi == n) and 0 <= lo <= hi <= n
        rows = torch.arange(lo, hi, dtype=torch.float32)[:, None] * torch.ones((1, 3))
        full = D.all_gather_rows(rows)
        ok = ok and torch.equal(full, torch.arange(n, dtype=torch.float32)[:, None] * torch.ones((1, 3)))
    q.put((rank, bool(ok)))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_optimizer_exchange_gloo(world):
    """distributed.sharded_step (reduce-scatter -> Adam on 1/world -> all-gather) equals replicated Adam on the mean."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31000 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(r, True) for r in range(world)]


def _table_parallel_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import samnerf_amd  # noqa: F401
    from samnerf_amd import distributed as D
    from oracle import samnerf_oracle as O
    D.init_distributed(backend="gloo")
    T, F, N = 10, 8, 96
    grids = [(6, F, T), (6, F, T)]
    layout = D.TableParallelLayout(grids, world)
    gen = torch.Generator().manual_seed(5)
    tables = [torch.randn((L << T, F), generator=gen) for L, _, _ in grids]
    scal = [O.hash_scalings(6, 16, 128), O.hash_scalings(6, 128, 512)]
    us = [torch.rand((N, 3), generator=gen) for _ in range(world)]          # every rank's samples ...
    gs = [torch.randn((N, layout.total), generator=gen) for _ in range(world)]  # ... and upstream gradients
    ok = True

    # forward: features of the own samples, assembled from the owners, equal the replicated evaluation
    U = D.tp_gather_positions(us[rank])
    ok = ok and torch.equal(U, torch.cat(us))

    def eval_run(gi, l0, nl, out, ld, col):
        out[:, col:col + nl * F] = O.hashgrid_fwd(U, tables[gi][l0 << T:(l0 + nl) << T], scal[gi][l0:l0 + nl], T)

    out = D.tp_forward(U, N, layout, eval_run)
    full = torch.cat([O.hashgrid_fwd(us[rank], tables[g], scal[g], T) for g in range(2)], dim=1)
    ok = ok and torch.equal(out, full)

    # backward: the owned levels receive the sum over every rank's samples, exactly what the all-reduce would have formed
    G = D.tp_backward(gs[rank], layout)
    ref = [torch.zeros_like(t) for t in tables]
    for w in range(world):
        leaves = [t.clone().requires_grad_(True) for t in tables]
        y = torch.cat([O.hashgrid_fwd(us[w], leaves[g], scal[g], T) for g in range(2)], dim=1)
        y.backward(gs[w])
        for g in range(2):
            ref[g] += leaves[g].grad
    for gi, l0, nl, col in layout.runs(rank):
        slab = tables[gi][l0 << T:(l0 + nl) << T].clone().requires_grad_(True)
        O.hashgrid_fwd(U, slab, scal[gi][l0:l0 + nl], T).backward(G[:, col:col + nl * F])
        ok = ok and torch.allclose(slab.grad, ref[gi][l0 << T:(l0 + nl) << T], rtol=1e-5, atol=1e-6)
        lo, hi = layout.owned_elements(rank, gi)
        ok = ok and (lo, hi) == ((l0 << T) * F, ((l0 + nl) << T) * F)

    # consolidation: every owner's levels reach every rank
    for gi in range(2):
        flat = tables[gi].clone().view(-1)
        lo, hi = layout.owned_elements(rank, gi)
        flat[lo:hi] += rank + 1
        D.tp_refresh_table(flat, layout, gi)
        expect = tables[gi].clone().view(-1)
        for w in range(world):
            lo, hi = layout.owned_elements(w, gi)
            expect[lo:hi] += w + 1
        ok = ok and torch.equal(flat, expect)
    q.put((rank, bool(ok)))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4])
def test_table_parallel_hash_grids_gloo(world):
    """distributed.tp_forward / tp_backward (levels sharded over the ranks, activations exchanged) against the replicated
    evaluation and the all-reduced gradient, with the oracle's hash grid as the per-level arithmetic."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33000 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_table_parallel_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(r, True) for r in range(world)]


def test_table_parallel_layout_covers_every_slab_once():
    from samnerf_amd.distributed import TableParallelLayout as TPL
    for grids in ([(12, 8, 19), (12, 8, 19)], [(6, 8, 10), (6, 8, 10)], [(5, 8, 4), (7, 8, 6), (12, 8, 5)]):
        n = sum(L for L, _, _ in grids)
        for world in range(1, 13):
            if n % world:
                assert not TPL.supported(grids, world)
                continue
            lay = TPL(grids, world)
            seen = []
            for r in range(world):
                col = 0
                for gi, l0, nl, c in lay.runs(r):
                    assert c == col and nl > 0
                    col += nl * lay.F
                    seen += [(gi, l) for l in range(l0, l0 + nl)]
                    assert lay.owned_levels(r, gi) == (l0, l0 + nl)
                assert col == lay.width
            assert seen == [(gi, l) for gi, (L, _, _) in enumerate(grids) for l in range(L)]
    assert not TPL.supported([(12, 8, 19), (12, 2, 19)], 2)


def _engine_worker(rank, world, port, q, table_parallel):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), SNF_TABLE_PARALLEL="1" if table_parallel else "0")
    sys.path.insert(0, ROOT)
    import samnerf_amd  # noqa: F401
    from samnerf_amd import configs, distributed as D, engine, model, ops
    D.init_distributed(backend="gloo")

    # the two Adam launches as plain torch (the C-ABI kernels need a GPU); everything around them is the product code
    def adam_step_(p, g, m, v, lr, b1, b2, eps, t, scale=1.0, zero=True):
        _adam_ref(p, g, m, v, lr, b1, b2, eps, t, scale)
        if zero:
            g.zero_()

    def adam_step_rows_(p, g, m, v, rows, F, lr, b1, b2, eps, t, scale=1.0, zero=True):
        idx = (rows.long()[:, None] + torch.arange(F)[None, :]).reshape(-1)
        pp, gg, mm, vv = p[idx], g[idx], m[idx], v[idx]
        _adam_ref(pp, gg, mm, vv, lr, b1, b2, eps, t, scale)
        p[idx], m[idx], v[idx] = pp, mm, vv
        if zero:
            g[idx] = 0

    ops.adam_step_, ops.adam_step_rows_ = adam_step_, adam_step_rows_
    mc = copy.deepcopy(configs.method_configs["samnerf_distill"].pipeline.model)
    mc.log2_hashmap_size, mc.hashgrid_sizes = 8, (14, 14)  # T=14: the coarsest feature level is a reachable-row segment
    mc.proposal_net_args_list = [dict(a, log2_hashmap_size=8) for a in mc.proposal_net_args_list]
    samnerf_amd.tcnn_compat.manual_seed(3)
    m = mc.setup(scene_box=model.SceneBox(), num_train_data=2, device="cpu")
    arenas = m.build_arenas()
    opt = engine.Optimizers(copy.deepcopy(configs.method_configs["samnerf_distill"].optimizers), arenas)
    a = arenas["sam_field"]
    names = list(a.offsets)
    n_sam = len(list(m.sam_field.clip_encs.parameters())) + len(list(m.sam_field.sam_net.parameters()))
    slices = [(0, n_sam), (n_sam, len(names))]  # the trainer steps the two heads separately (pipeline.py)
    opt.shard_slices["sam_field"] = [(a.offsets[names[lo]][0], a.offsets[names[hi]][0] if hi < len(names) else a.numel)
                                     for lo, hi in slices]
    owned = opt._tp_tables("sam_field")
    plan = opt._plan("sam_field")
    ok = (len(owned) == 4) == table_parallel and any(s[0] == "rows" for s in plan)
    # gradients can only be non-zero on rows some input can address
    live = torch.ones(a.numel)
    for seg in plan:
        if seg[0] == "rows":
            live[seg[1]:seg[2]] = 0
            live[(seg[3].long()[:, None] + torch.arange(seg[4])[None, :]).reshape(-1)] = 1
    pr, mr, vr = a.param.clone(), torch.zeros(a.numel), torch.zeros(a.numel)
    oc = opt.config["sam_field"]["optimizer"]
    for t in (1, 2, 3):
        gen = torch.Generator().manual_seed(100 + t)
        local = [torch.randn(a.numel, generator=gen) * live for _ in range(world)]  # each rank's backward, replicated tables
        total = sum(local)
        _adam_ref(pr, total.clone(), mr, vr, opt.lr("sam_field"), oc.betas[0], oc.betas[1], oc.eps, t, 1.0 / world)
        a.grad.copy_(local[rank])
        for toff, tend, lo, hi, _, _ in owned:  # table-parallel backward: all ranks' samples, owned levels only
            a.grad[toff:tend] = 0
            a.grad[lo:hi] = total[lo:hi]
        first = True
        for lo_i, hi_i in slices:
            opt.exchange_and_step("sam_field", lo_i, hi_i, count_step=first)
            first = False
        ok = ok and bool(torch.count_nonzero(a.grad) == 0)
    opt.consolidate_state()
    for got, ref in ((a.param, pr), (a.exp_avg, mr), (a.exp_avg_sq, vr)):
        ok = ok and torch.allclose(got, ref, rtol=1e-5, atol=1e-7)
    # checkpoint payload: per-parameter moments keyed by name (whole on every rank after the consolidation) into a fresh optimizer
    sd = copy.deepcopy(opt.state_dict())
    ok = ok and list(sd["sam_field"]["state"]) == names and sd["sam_field"]["step"] == 3
    ok = ok and all(v["exp_avg"].shape == tuple(a.offsets[n][1]) for n, v in sd["sam_field"]["state"].items())
    arenas2 = mc.setup(scene_box=model.SceneBox(), num_train_data=2, device="cpu").build_arenas()
    opt2 = engine.Optimizers(copy.deepcopy(configs.method_configs["samnerf_distill"].optimizers), arenas2)
    opt2.load_optimizers(sd)
    a2 = arenas2["sam_field"]
    for name, (off, shape) in a.offsets.items():
        n = int(np.prod(shape))
        ok = ok and torch.equal(a2.exp_avg[off:off + n], a.exp_avg[off:off + n]) and torch.equal(a2.exp_avg_sq[off:off + n], a.exp_avg_sq[off:off + n])
    ok = ok and opt2.step_count["sam_field"] == 3
    q.put((rank, bool(ok)))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world,table_parallel", [(2, True), (3, True), (2, False)])
def test_engine_exchange_and_step_gloo(world, table_parallel):
    """engine.Optimizers.exchange_and_step on the 'sam_field' arena of a small model, per head slice as the trainer calls it
    (plan segments x table-parallel ownership x sharded dense exchange x reachable-row segments), against replicated Adam
    on the mean gradient; then consolidate_state makes parameters and moments whole on every rank."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35000 + (os.getpid() % 2000) + world + 10 * table_parallel
    procs = [ctx.Process(target=_engine_worker, args=(r, world, port, q, table_parallel)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(r, True) for r in range(world)]


def test_shard_bounds_cover_the_slice():
    from samnerf_amd.distributed import shard_bounds
    for n in (0, 63, 64, 1000, 201_326_592 + 640):
        for world in (1, 2, 8):
            chunk, bulk, lo, hi = shard_bounds(n, world, world - 1)
            assert chunk % 64 == 0 and bulk == chunk * world <= n and n - bulk < world * 64 + 64
            assert hi == bulk and lo == bulk - chunk


def test_dataparser_pose_normalisation_vs_reference(golden):
    """data.auto_orient_and_center_poses ('up' / 'none', centred) against camera_utils.py:432-487 outputs."""
    import samnerf_amd  # noqa: F401
    from samnerf_amd.data import Cameras, auto_orient_and_center_poses
    g = golden("batch_builder")
    poses4 = torch.from_numpy(g["poses4"])
    for m in ("up", "none"):
        poses, tf = auto_orient_and_center_poses(poses4.clone(), m, True)
        assert float((poses - torch.from_numpy(g[f"{m}_poses"])).abs().max()) <= 1e-6
        assert float((tf - torch.from_numpy(g[f"{m}_tf"])).abs().max()) <= 1e-6
    cams = Cameras(poses, 50.0, 52.0, 20.0, 12.0, 40, 24)
    assert len(cams) == poses4.shape[0] and cams.intrinsics().shape == (len(cams), 4)
    assert cams.get_image_coords().shape == (24, 40, 2) and float(cams.get_image_coords()[0, 0, 0]) == 0.5


def test_reachable_rows_cover_every_addressable_row():
    """Encoding.active_rows: the sparse levels' row lists contain every row any input in [0,1]^3 (plus border slack) can
    address -- the invariant that makes skipping the other rows in Adam exact."""
    import samnerf_amd  # noqa: F401
    from samnerf_amd.tcnn_compat import Encoding
    from oracle import samnerf_oracle as O
    for (L, F, T, lo, hi) in ((12, 8, 17, 16, 128), (16, 2, 19, 16, 2048), (5, 2, 17, 16, 128)):
        growth = float(np.exp((np.log(hi) - np.log(lo)) / (L - 1)))
        enc = Encoding(3, {"otype": "HashGrid", "n_levels": L, "n_features_per_level": F, "log2_hashmap_size": T,
                           "base_resolution": lo, "per_level_scale": growth}, device="cpu")
        n_sparse, rows = enc.active_rows()
        assert 1 <= n_sparse < L and torch.equal(rows, torch.unique(rows))  # sorted, unique
        active = set(rows.tolist())
        g = torch.Generator().manual_seed(L)
        u = torch.cat([torch.rand((20000, 3), generator=g), torch.zeros((1, 3)), torch.ones((1, 3)),
                       torch.tensor([[1.0 + 1e-7, -1e-8, 0.5]]), torch.randint(0, 2, (64, 3)).float()])
        s = enc.scalings.view(-1, 1)
        scaled = u[:, None, :] * s
        c, f = torch.ceil(scaled).to(torch.int32), torch.floor(scaled).to(torch.int32)
        off = (torch.arange(L) * (1 << T)).to(torch.int64)
        touched = set()
        for a in (c[..., 0], f[..., 0]):
            for b in (c[..., 1], f[..., 1]):
                for cc in (c[..., 2], f[..., 2]):
                    idx = O.hash_index(a, b, cc, T, off)[:, :n_sparse]
                    touched |= set(idx.reshape(-1).tolist())
        assert touched <= active
        # the list is a real saving: fewer than 40 % of the rows of every sparse level
        per_level = torch.bincount((rows >> T), minlength=n_sparse)
        assert int(per_level.max()) < 0.4 * (1 << T)
    # full-size SAM grid (T = 19, 16 -> 128): 8 sparse levels, ~1.0 M of 4.2 M rows
    enc = Encoding(3, {"otype": "HashGrid", "n_levels": 12, "n_features_per_level": 8, "log2_hashmap_size": 19,
                       "base_resolution": 16, "per_level_scale": float(np.exp(np.log(128 / 16) / 11))}, device="cpu")
    n_sparse, rows = enc.active_rows()
    assert n_sparse == 8 and rows.numel() < 0.3 * (n_sparse << 19)


def test_every_exported_symbol_is_documented():
    """INTEGRATION.md names the reference interface behind every entry point declared in include/samnerf_hip.h."""
    hdr = open(os.path.join(ROOT, "include", "samnerf_hip.h")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    declared = sorted(set(re.findall(r"\b(snf_[a-z0-9_]+)\s*\(", hdr)))
    missing = [n for n in declared if n not in doc]
    assert not missing, missing


def test_eval_regroup_and_set_feature_vs_reference(golden):
    """Plugin-side eval regroup (model.feature_ray_bundle / clipseg_ray_bundle on this package's RayBundle) and the
    SamPredictor.set_feature zero-pad against fixtures produced by the reference's own RayBundle / TensorDataclass and its
    own set_feature body (tests/golden/make_golden.py: fx_eval_regroup)."""
    import samnerf_amd  # noqa: F401
    from samnerf_amd.model import clipseg_ray_bundle, feature_ray_bundle
    from samnerf_amd.rays import RayBundle
    from samnerf_amd.sam_utils import get_feature_size, set_feature
    g = golden("eval_regroup")
    for ci in range(3):
        H, W, fh, fw, p = (int(v) for v in g[f"c{ci}_hw"])
        assert get_feature_size(H, W) == (fh, fw)
        ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
        cam = RayBundle(origins=torch.stack([ys, xs, torch.zeros_like(ys)], -1).float(),
                        directions=torch.from_numpy(g[f"c{ci}_directions"]),
                        pixel_area=torch.from_numpy(g[f"c{ci}_pixel_area"]),
                        camera_indices=torch.zeros((H, W, 1), dtype=torch.long))
        fb = feature_ray_bundle(cam, fh, fw, p)
        chunk = 1000  # the chunk loop of get_outputs_for_camera_ray_bundle walks the bundle row-major
        parts = [fb.get_row_major_sliced_ray_bundle(i, min(i + chunk, len(fb))) for i in range(0, len(fb), chunk)]
        assert torch.equal(torch.cat([q.origins for q in parts]), torch.from_numpy(g[f"c{ci}_feat_origins"]))
        assert torch.equal(torch.cat([q.directions for q in parts]), torch.from_numpy(g[f"c{ci}_feat_directions"]))
        assert torch.equal(torch.cat([q.pixel_area for q in parts]), torch.from_numpy(g[f"c{ci}_feat_pixel_area"]))
        cb = clipseg_ray_bundle(cam)
        assert torch.equal(cb.flatten().origins, torch.from_numpy(g[f"c{ci}_clip_origins"]))
    for name in ("land", "square"):
        out, input_size = set_feature(torch.from_numpy(g[f"sf_{name}_in"]), tuple(int(v) for v in g[f"sf_{name}_hw"]))
        assert torch.equal(out, torch.from_numpy(g[f"sf_{name}_out"]))
        assert tuple(input_size) == tuple(int(v) for v in g[f"sf_{name}_input_size"])
    # portrait: the reference's own code raises (its zero block has the wrong axis); here the map is padded to a square
    assert int(g["sf_portrait_reference_raises"]) == 1
    out, input_size = set_feature(torch.ones((6, 64, 40)), (512, 320))
    assert out.shape == (1, 6, 64, 64) and float(out[..., 40:].abs().max()) == 0.0 and input_size == (1024, 640)


def test_plugin_entry_points_resolve():
    """pyproject.toml registers the two methods in nerfstudio's `nerfstudio.method_configs` group
    (nerfstudio/plugins/registry.py:35-51); every entry point must load to a MethodSpecification-shaped object whose config
    is a TrainerConfig with this package's SAMModelConfig.  With nerfstudio importable the real registry type is used."""
    import importlib
    import tomli
    import samnerf_amd  # noqa: F401
    from samnerf_amd import model as M
    meta = tomli.load(open(os.path.join(ROOT, "pyproject.toml"), "rb"))
    eps = meta["project"]["entry-points"]["nerfstudio.method_configs"]
    assert set(eps) == {"samnerf_distill_mi355x", "samnerf_no_distill_mi355x"}
    assert meta["tool"]["setuptools"]["package-dir"]["samnerf_amd"] == "segment-anything-in-nerf_amd"
    for name, target in eps.items():
        mod, attr = target.split(":")
        spec = getattr(importlib.import_module(mod), attr)
        assert spec.config.method_name == name and isinstance(spec.description, str) and spec.description
        mc = spec.config.pipeline.model
        assert isinstance(mc, M.SAMModelConfig) and mc._target is M.SAMModel
        assert mc.distill_sam == ("no_distill" not in name)
        assert set(spec.config.optimizers) >= {"proposal_networks", "fields"}
    plugin = importlib.import_module("samnerf_amd.plugin")
    if plugin.HAVE_NERFSTUDIO:
        from nerfstudio.plugins.types import MethodSpecification
        assert isinstance(plugin.samnerf_distill, MethodSpecification)


def test_eval_image_metrics_psnr_and_ssim():
    """`get_image_metrics_and_images` (nerfacto.py:346-383): psnr = -10 log10(mse); ssim is 1 for identical images, symmetric, below 1
    for a perturbed image and falls with the perturbation; the 11 x 11 / sigma 1.5 window against a direct per-pixel evaluation of the
    definition on a small image."""
    import samnerf_amd  # noqa: F401
    from samnerf_amd.model import NerfactoModel
    g = torch.Generator().manual_seed(0)
    img = torch.rand((24, 20, 3), generator=g)
    noisy = (img + 0.05 * torch.randn(img.shape, generator=g)).clamp(0, 1)
    worse = (img + 0.2 * torch.randn(img.shape, generator=g)).clamp(0, 1)
    m, images = NerfactoModel.get_image_metrics_and_images(None, {"rgb": noisy}, {"image": img})
    assert abs(m["psnr"] + 10 * np.log10(float(((img - noisy) ** 2).mean()))) < 1e-4 and images["img"].shape == (24, 40, 3)
    to = lambda x: torch.moveaxis(x, -1, 0)[None]  # noqa: E731
    s_same, s_n, s_w = (float(NerfactoModel.ssim(to(img), to(x))) for x in (img, noisy, worse))
    assert abs(s_same - 1.0) < 1e-6 and s_w < s_n < 1.0 and abs(m["ssim"] - s_n) < 1e-6
    assert abs(float(NerfactoModel.ssim(to(noisy), to(img))) - s_n) < 1e-6
    # direct evaluation at one interior pixel of channel 0
    k, sig = 11, 1.5
    w1 = np.exp(-((np.arange(k) - 5) ** 2) / (2 * sig * sig)); w1 /= w1.sum()
    w = np.outer(w1, w1)
    a, b = img[..., 0].numpy().astype(np.float64), noisy[..., 0].numpy().astype(np.float64)
    y, x = 12, 10
    pa, pb = a[y - 5:y + 6, x - 5:x + 6], b[y - 5:y + 6, x - 5:x + 6]
    mu_a, mu_b = (w * pa).sum(), (w * pb).sum()
    sa, sb, sab = (w * pa * pa).sum() - mu_a ** 2, (w * pb * pb).sum() - mu_b ** 2, (w * pa * pb).sum() - mu_a * mu_b
    dr = max(float(img.max() - img.min()), float(noisy.max() - noisy.min()))
    c1, c2 = (0.01 * dr) ** 2, (0.03 * dr) ** 2
    ref = ((2 * mu_a * mu_b + c1) * (2 * sab + c2)) / ((mu_a ** 2 + mu_b ** 2 + c1) * (sa + sb + c2))
    # the same pixel from the implementation's map: recompute the map on channel 0 alone with the full-image data range
    import torch.nn.functional as Fn
    g1 = torch.tensor(w1, dtype=torch.float32)
    win = (g1[:, None] * g1[None, :])[None, None]
    def blur(t): return Fn.conv2d(Fn.pad(t[None, None], (5, 5, 5, 5), mode="reflect"), win)[0, 0, 5:-5, 5:-5]
    ta, tb = img[..., 0], noisy[..., 0]
    mu_p, mu_t = blur(ta), blur(tb)
    s_p, s_t, s_pt = blur(ta * ta) - mu_p ** 2, blur(tb * tb) - mu_t ** 2, blur(ta * tb) - mu_p * mu_t
    mp = ((2 * mu_p * mu_t + c1) * (2 * s_pt + c2)) / ((mu_p ** 2 + mu_t ** 2 + c1) * (s_p + s_t + c2))
    assert abs(float(mp[y - 5, x - 5]) - ref) < 1e-4

"""CPU: the oracle restatement (oracle/samnerf_oracle.py) against the committed golden vectors.

The vectors were produced by importing the reference's own torch components
(tests/golden/make_golden.py); these tests keep the oracle pinned to them everywhere.
"""
import numpy as np
import pytest
import torch

from oracle import samnerf_oracle as O


def T(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, tol=0.0):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    assert a.shape == b.shape
    nan = torch.isnan(a) & torch.isnan(b)
    d = torch.where(nan, torch.zeros_like(a, dtype=torch.float32), (a - b).abs().float())
    assert float(d.max()) <= tol, float(d.max())


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_spacing(golden, mode):
    g = golden(f"spacing_{mode}")
    t = T(g["t_rand"]) if mode == "train" else None
    sb, eb = O.sample_spacing(T(g["nears"]), T(g["fars"]), 64, t)
    close(sb.expand(64, -1), T(g["sbins"]))
    close(eb, T(g["ebins"]))
    close(O.sample_positions(T(g["origins"]), T(g["directions"]), eb), T(g["positions"]))


def test_contraction(golden):
    g = golden("contraction")
    x = T(g["x"])
    close(O.contract(x, float("inf")), T(g["linf"]))
    close(O.contract(x, None), T(g["l2"]))
    u, sel = O.normalize_positions(x, float("inf"), True)
    close(u, T(g["u_linf_sel"]))
    assert torch.equal(sel, T(g["selector"]))


@pytest.mark.parametrize("name", ["prop", "field", "feat_a", "feat_b"])
@pytest.mark.parametrize("log2_T", [10, 12])
def test_hashgrid(golden, name, log2_T):
    g = golden(f"hashgrid_{name}_T{log2_T}")
    table = T(g["table"]).clone().requires_grad_(True)
    out = O.hashgrid_fwd(T(g["u"]), table, T(g["scalings"]), int(g["log2_T"]))
    close(out, T(g["out"]))
    (out * T(g["grad_out"])).sum().backward()
    close(table.grad, T(g["grad_table"]), 1e-7)


def test_scalings_known_vectors():
    assert O.hash_scalings(5, 16, 128).tolist() == [16, 26, 45, 76, 128]
    assert O.hash_scalings(12, 128, 512).tolist()[-1] == 511  # fp32 rounding: not 512
    assert O.hash_scalings(16, 16, 2048).tolist()[-1] == 2047


@pytest.mark.parametrize("name", ["prop_nobias", "prop_bias", "base_nobias", "base_bias", "head_nobias",
                                  "head_bias", "sam_nobias", "clipseg_nobias"])
def test_mlp(golden, name):
    g = golden("mlp_" + name)
    n = int(g["n_layers"])
    ws = [T(g[f"w{i}"]).clone().requires_grad_(True) for i in range(n)]
    bs = [T(g[f"b{i}"]).clone().requires_grad_(True) for i in range(n)] if "b0" in g else None
    x = T(g["x"]).clone().requires_grad_(True)
    act = str(g["out_act"])
    y = O.mlp_fwd(x, ws, bs, None if act == "none" else act)
    close(y, T(g["y"]))
    (y * T(g["grad_y"])).sum().backward()
    close(x.grad, T(g["grad_x"]), 1e-7)
    for i in range(n):
        close(ws[i].grad, T(g[f"gw{i}"]), 1e-6)


def test_sh16(golden):
    g = golden("sh16")
    close(O.sh16(T(g["directions"])), T(g["sh"]))


def test_weights_and_trunc_exp(golden):
    g = golden("weights")
    d = T(g["density"]).clone().requires_grad_(True)
    w = O.weights_from_density(d, T(g["deltas"]))
    close(w, T(g["weights"]))
    rows = T(g["finite_rows"]).long()
    (w[rows] * T(g["grad_w"])[rows]).sum().backward()
    close(torch.nan_to_num(d.grad), T(g["grad_density"]), 1e-6)
    x = T(g["te_x"]).clone().requires_grad_(True)
    y = O.trunc_exp(x)
    close(y, T(g["te_y"]))
    y.sum().backward()
    close(x.grad, T(g["te_grad"]))


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_pdf(golden, mode):
    g = golden(f"pdf_{mode}")
    u = T(g["u_rand"]) if mode == "train" else None
    sb = O.pdf_resample(T(g["weights"]), T(g["sbins_in"]), int(g["num_samples"]), u)
    close(sb, T(g["sbins"]))
    close(O.s_to_euclid(sb, T(g["nears"]), T(g["fars"])), T(g["ebins"]))


def test_render(golden):
    g = golden("render")
    w, rgb, eb = T(g["weights"]), T(g["rgb_samples"]), T(g["ebins"])
    close(O.render_rgb(rgb.clone(), w, True), T(g["rgb_train"]))
    close(O.render_rgb(rgb.clone(), w, False), T(g["rgb_eval"]))
    close(O.render_accumulation(w), T(g["accumulation"]))
    close(O.render_depth_median(w, eb), T(g["depth"]))


def test_topk_mean_loss(golden):
    g = golden("topk")
    w, feats = T(g["weights"]), T(g["feats"])
    sw, ids = O.topk_sharpen(w, int(g["k"]), float(g["temperature"]))
    rows = [r for r in range(w.shape[0]) if r not in set(g["nan_rows"].tolist())]
    assert torch.equal(torch.sort(ids, -1)[0][rows], torch.sort(T(g["ids"]), -1)[0][rows])
    assert torch.isnan(sw[0]).all() and torch.isnan(sw[1]).all()
    mean = O.feature_mean(torch.gather(feats, 1, ids[..., None].expand(-1, -1, feats.shape[-1])), sw)
    close(mean, T(g["mean"]), 1e-6)
    close(O.feature_loss(mean, T(g["target"])), T(g["loss"]), 1e-6)


def test_losses(golden):
    g = golden("losses")
    wp = T(g["w_prop"]).clone().requires_grad_(True)
    wf = T(g["w_fine"]).clone().requires_grad_(True)
    li = O.interlevel_loss(T(g["sbins_fine"]), wf, T(g["sbins_prop"]), wp)
    ld = O.distortion_loss(T(g["sbins_fine"]), wf)
    close(li, T(g["interlevel"]), 1e-7)
    close(ld, T(g["distortion"]), 1e-7)
    (li + ld).backward()
    close(wp.grad, T(g["grad_w_prop"]), 1e-7)
    close(wf.grad, T(g["grad_w_fine"]), 1e-7)


def test_ministep(golden):
    g = golden("ministep")
    cfg = O.PathConfig(num_proposal_samples=int(g["P"]), num_nerf_samples=int(g["S"]),
                       num_sam_samples=int(g["K"]), patch_size=int(g["patch"])).small(int(g["log2_T"]))
    params = O.init_params(cfg, seed=int(g["seed_params"]), table_scale=float(g["table_scale"]))
    params = {k: v.requires_grad_(True) for k, v in params.items()}
    o, d = O.synthetic_rays(int(g["num_rays"]), int(g["seed_rays"]))
    close(o, T(g["origins"]))
    batch = O.synthetic_batch(cfg, int(g["num_rays"]), int(g["seed_batch"]))
    out = O.forward(params, cfg, o, d, True, T(g["t_rand"]), T(g["u_rand"]), float(g["anneal"]))
    close(out["sbins_fine"], T(g["sbins_fine"]))
    close(out["weights_fine"], T(g["w_fine"]), 1e-7)
    close(out["rgb"], T(g["rgb"]), 1e-6)
    close(out["depth"], T(g["depth"]))
    close(out["prop_depth_0"], T(g["prop_depth_0"]))
    close(out["sam"], T(g["sam"]), 1e-6)
    close(out["clipseg"], T(g["clipseg"]), 1e-6)
    ld = O.loss_dict(out, batch, cfg)
    for k in ("rgb_loss", "interlevel_loss", "distortion_loss", "sam_loss", "clipseg_loss"):
        close(ld[k], T(g[k]), 1e-6)
    sum(ld.values()).backward()
    for k, p in params.items():
        close(p.grad, T(g["grad_" + k]), 2e-6)


def test_batch_builder(golden):
    """pixel samplers, pinhole ray generation, nearest feature gather vs the reference's outputs."""
    g = golden("batch_builder")
    T = lambda k: torch.from_numpy(np.ascontiguousarray(g[k]))
    N, H, W, p = int(g["N"]), int(g["H"]), int(g["W"]), int(g["p"])
    ind = O.patch_pixel_indices(T("u_patch"), N, H, W, p)
    assert torch.equal(ind, T("patch_indices"))
    assert torch.equal(O.pixel_indices(T("u_pix"), N, H, W), T("pix_indices"))
    o, d, pa, ci = O.generate_rays(ind, T("c2w"), T("fx"), T("fy"), T("cx"), T("cy"))
    assert torch.equal(o, T("origins")) and torch.equal(d, T("directions")) and torch.equal(pa, T("pixel_area"))
    assert torch.equal(ci, T("camera_indices"))
    (bo, bd, bpa, bci), batch = O.build_batch(T("u_patch"), T("images"), T("c2w"), T("fx"), T("fy"), T("cx"), T("cy"), p,
                                              T("sam"), T("clip"))
    assert torch.equal(batch["image"], T("batch_image")) and torch.equal(batch["sam"], T("batch_sam"))
    assert torch.equal(batch["clipseg"], T("batch_clipseg")) and torch.equal(bd, T("directions"))
    assert float((d.norm(dim=-1) - 1).abs().max()) < 1e-6


def test_vit_oracle_vs_reference_fixture(golden):
    """oracle/vit_oracle.py against the outputs of the reference's ImageEncoderViT (small configuration:
    one padded windowed block, one global block, random rel-pos tables)."""
    from oracle import vit_oracle as V
    g = golden("vit_small")
    cfg = V.ViTConfig(img_size=224, patch_size=16, embed_dim=32, depth=2, num_heads=2, mlp_ratio=2.0, out_chans=16,
                      window_size=5, global_attn_indexes=(1,))
    sd = {k[2:]: torch.from_numpy(np.ascontiguousarray(g[k])) for k in g.keys() if k.startswith("w:")}
    assert set(sd) == set(V.init_weights(cfg).keys())
    y, trace = V.forward(sd, torch.from_numpy(g["x"]), cfg, return_tokens=True)
    close(trace[0], T(g["t0"]), 1e-6)
    close(trace[1], T(g["t1"]), 2e-6)
    close(trace[2], T(g["t2"]), 2e-6)
    close(y, T(g["y"]), 2e-6)


def test_eval_regroup(golden):
    """The eval-path ray regroup (sam_model.py:371-398) against what the reference's own RayBundle / TensorDataclass
    machinery produced: round 1 checked O.render_camera only against the HIP path."""
    g = golden("eval_regroup")
    for ci in range(3):
        H, W, fh, fw, p = (int(v) for v in g[f"c{ci}_hw"])
        ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
        origins = torch.stack([ys, xs, torch.zeros_like(ys)], -1).float()
        assert O.get_feature_size(H, W) == (fh, fw)
        close(O.feature_ray_grid(origins, fh, fw, p), T(g[f"c{ci}_feat_origins"]))
        close(O.feature_ray_grid(T(g[f"c{ci}_directions"]), fh, fw, p), T(g[f"c{ci}_feat_directions"]))
        close(O.feature_ray_grid(T(g[f"c{ci}_pixel_area"]), fh, fw, p), T(g[f"c{ci}_feat_pixel_area"]))
        close(O.clipseg_ray_grid(origins), T(g[f"c{ci}_clip_origins"]))


def test_gradient_conditioning():
    """Why the composed-step gradient tests (conftest.grad_parity) use a relative L1 criterion: the oracle's own train step in
    fp32 and in fp64 (same parameters, rays, jitter).  The first-layer ReLU masks of the fields flip for the few (sample, unit)
    pairs whose pre-activation is inside the fp32 rounding of zero, which moves single rows / table entries by up to ~1 % of the
    tensor's largest entry while the rest agrees to 1e-5; the last layers agree to 2e-5 everywhere."""
    R, P, S, K, patch, T = 192, 64, 48, 3, 1, 12
    cfg = O.PathConfig(num_proposal_samples=P, num_nerf_samples=S, num_sam_samples=K, patch_size=patch, use_clipseg=False).small(T)
    params = O.init_params(cfg, seed=3, table_scale=0.05)
    o, d = O.synthetic_rays(R, 5)
    batch = O.synthetic_batch(cfg, R, 6)
    gen = torch.Generator().manual_seed(7)
    t_rand, u_rand = torch.rand((R, 1), generator=gen), torch.rand((R, 1), generator=gen)
    grads = {}
    for dt in (torch.float32, torch.float64):
        op = {k: v.clone().to(dt).requires_grad_(True) for k, v in params.items()}
        ref = O.forward(op, cfg, o.to(dt), d.to(dt), True, t_rand.to(dt), u_rand.to(dt), 0.5)
        b = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in batch.items()}
        sum(O.loss_dict(ref, b, cfg).values()).backward()
        grads[dt] = {k: v.grad.double() for k, v in op.items() if v.grad is not None}
    l1, mx = {}, {}
    for k, g64 in grads[torch.float64].items():
        e = (grads[torch.float32][k] - g64).abs()
        l1[k], mx[k] = float(e.sum() / g64.abs().sum()), float(e.max() / g64.abs().max())
    assert max(l1.values()) <= 5e-3, l1
    assert max(mx[k] for k in ("head_w2", "sam_w1", "base_w1")) <= 1e-4, mx
    # the flip signature: outliers of >1e-3 of the largest entry in tensors whose relative L1 error stays at 1e-3
    assert max(mx["field_table"], mx["base_w0"], mx["sam_w0"]) >= 1e-3 >= max(l1["field_table"], l1["base_w0"], l1["sam_w0"]), (l1, mx)


def test_sam_preprocess(golden):
    """oracle/vit_oracle.sam_preprocess against the reference's own Sam.preprocess (modeling/sam.py:164-174) on uint8 / float,
    landscape / portrait / square inputs; the offline writer's crop (get_image_embeddings.py:29-35) on the shapes it documents."""
    from oracle import vit_oracle as V
    g = golden("sam_preprocess")
    for k in "abc":
        got = V.sam_preprocess(T(g[k]), g["mean"], g["std"], 64)
        assert got.shape == g[k + "_out"].shape and torch.equal(got, T(g[k + "_out"])), k
    import samnerf_amd  # noqa: F401  (repo-root shim; conftest put the root on sys.path)
    from samnerf_amd.sam_utils import crop_embedding, get_feature_size, set_feature
    f = torch.arange(2 * 64 * 64, dtype=torch.float32).view(1, 2, 64, 64)
    assert crop_embedding(f, (683, 1024)).shape == (1, 2, 43, 64) and crop_embedding(f, (1024, 683)).shape == (1, 2, 64, 43)
    assert crop_embedding(f, (512, 512)).shape == (1, 2, 64, 64)
    assert crop_embedding(f, (683, 1024)).shape[-2:] == get_feature_size(683, 1024)  # the shape the eval path renders
    back, _ = set_feature(crop_embedding(f, (683, 1024)).squeeze(0), (683, 1024))  # writer's crop -> set_feature's pad: round trip
    assert torch.equal(back[..., :43, :], f[..., :43, :]) and float(back[..., 43:, :].abs().max()) == 0.0

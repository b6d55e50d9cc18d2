#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/*.npz by IMPORTING THE REFERENCE (build container only).

    python tests/golden/make_golden.py            # needs /root/reference; writes tests/golden/*.npz

Every fixture holds inputs + the outputs of the reference's own torch components (nerfstudio /
samnerf python, CPU fp32) for one hot-path function of SURVEY.md section 8(a).  While generating,
each reference output is also compared with the restatement in oracle/samnerf_oracle.py (the
script aborts on a mismatch), which is how the oracle is pinned.  The reference itself never
travels: fixtures contain only numeric arrays and scalar hyper-parameters.

The two packages under tests/golden/_stubs (torchtyping, nerfacc) only satisfy import statements
of names the reference never executes on this path (fx_batch_builder adds empty cv2 / torchvision modules likewise).
`python tests/golden/make_golden.py fx_batch_builder` regenerates a single fixture.
"""
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("SAMNERF_REFERENCE", "/root/reference")
sys.path[:0] = [os.path.join(HERE, "_stubs"), REF, ROOT]
warnings.filterwarnings("ignore")

import numpy as np  # noqa: E402
import torch  # noqa: E402

from nerfstudio.cameras.rays import RayBundle  # noqa: E402
from nerfstudio.field_components.activations import trunc_exp as ref_trunc_exp  # noqa: E402
from nerfstudio.field_components.encodings import HashEncoding  # noqa: E402
from nerfstudio.field_components.mlp import MLP  # noqa: E402
from nerfstudio.field_components.spatial_distortions import SceneContraction  # noqa: E402
from nerfstudio.model_components.losses import distortion_loss as ref_distortion  # noqa: E402
from nerfstudio.model_components.losses import interlevel_loss as ref_interlevel  # noqa: E402
from nerfstudio.model_components.ray_samplers import (  # noqa: E402
    PDFSampler, ProposalNetworkSampler, UniformLinDispPiecewiseSampler)
from nerfstudio.model_components.renderers import (  # noqa: E402
    AccumulationRenderer, DepthRenderer, RGBRenderer)
from nerfstudio.model_components.scene_colliders import NearFarCollider  # noqa: E402
from nerfstudio.utils.math import components_from_spherical_harmonics  # noqa: E402

from oracle import samnerf_oracle as O  # noqa: E402

torch.set_num_threads(8)


def npz(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {name}.npz  ({os.path.getsize(path) / 1024:.1f} KiB)")


def check(name, a, b, tol=0.0):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    both_nan = torch.isnan(a) & torch.isnan(b)
    d = torch.where(both_nan, torch.zeros_like(a, dtype=torch.float32), (a - b).abs().float())
    m = float(d.max()) if d.numel() else 0.0
    assert m <= tol, f"ORACLE != REFERENCE for {name}: max abs diff {m} > {tol}"


def make_bundle(R, seed, training):
    o, d = O.synthetic_rays(R, seed)
    rb = RayBundle(origins=o, directions=d, pixel_area=torch.full((R, 1), 1e-6),
                   camera_indices=torch.zeros((R, 1), dtype=torch.long))
    col = NearFarCollider(near_plane=0.05, far_plane=1000.0)
    col.train(training)
    return col(rb), o, d


# ---------------------------------------------------------------------------------------------
def fx_spacing():
    R, P = 64, 64
    for mode in ("train", "eval"):
        training = mode == "train"
        rb, o, d = make_bundle(R, 3, training)
        s = UniformLinDispPiecewiseSampler(num_samples=P, single_jitter=True)
        s.train(training)
        torch.manual_seed(11)
        rs = s(rb, num_samples=P)
        torch.manual_seed(11)
        t_rand = torch.rand((R, 1)) if training else None
        sb = torch.cat([rs.spacing_starts[..., 0], rs.spacing_ends[..., -1:, 0]], -1)
        eb = torch.cat([rs.frustums.starts[..., 0], rs.frustums.ends[..., -1:, 0]], -1)
        pos = rs.frustums.get_positions()
        osb, oeb = O.sample_spacing(rb.nears, rb.fars, P, t_rand)
        check("spacing sbins " + mode, sb, osb.expand_as(sb))
        check("spacing ebins " + mode, eb, oeb)
        check("positions " + mode, pos, O.sample_positions(o, d, oeb))
        npz(f"spacing_{mode}", origins=o, directions=d, nears=rb.nears, fars=rb.fars,
            t_rand=t_rand if training else np.zeros((0,)), sbins=sb, ebins=eb, positions=pos,
            deltas=rs.deltas[..., 0])


def fx_contraction():
    gen = torch.Generator().manual_seed(5)
    x = torch.randn((512, 3), generator=gen) * 2.0
    x[0] = torch.tensor([1.0, 0.0, 0.0])  # |x| == 1 exactly
    x[1] = torch.tensor([0.5, -1.0, 0.25])
    x[2] = torch.tensor([0.0, 0.0, 0.0])
    x[3] = torch.tensor([1000.0, -3.0, 2.0])
    x[4] = torch.tensor([0.6, 0.8, 0.0])  # L2 norm 1
    linf = SceneContraction(order=float("inf"))(x)
    l2 = SceneContraction()(x)
    check("contract linf", linf, O.contract(x, float("inf")))
    check("contract l2", l2, O.contract(x, None))
    u, sel = O.normalize_positions(x, float("inf"), True)
    npz("contraction", x=x, linf=linf, l2=l2, u_linf_sel=u, selector=sel)


def fx_hashgrid():
    shapes = {"prop": (5, 2, 16, 128), "field": (16, 2, 16, 2048), "feat_a": (12, 8, 16, 128),
              "feat_b": (12, 8, 128, 512)}
    for name, (L, F, mn, mx) in shapes.items():
        for T in (10, 12):
            enc = HashEncoding(num_levels=L, min_res=mn, max_res=mx, log2_hashmap_size=T,
                               features_per_level=F, implementation="torch")
            gen = torch.Generator().manual_seed(100 + T + L)
            table = (torch.rand((L << T, F), generator=gen) * 2 - 1) * 0.1
            enc.hash_table.data = table.clone()
            N = 257
            u = torch.rand((N, 3), generator=gen)
            u[0] = 0.0
            u[1] = 1.0
            u[2] = torch.tensor([0.0, 1.0, 0.5])
            u[3] = torch.tensor([3.0, 5.0, 7.0]) / 16.0  # lattice point of level 0
            u[4] = torch.tensor([0.25, 0.75, 0.5])
            out = enc.pytorch_fwd(u)
            gy = torch.randn(out.shape, generator=gen)
            (out * gy).sum().backward()
            g_ref = enc.hash_table.grad.clone()
            t2 = table.clone().requires_grad_(True)
            sc = O.hash_scalings(L, mn, mx)
            check(f"scalings {name}", enc.scalings, sc)
            o2 = O.hashgrid_fwd(u, t2, sc, T)
            check(f"hashgrid fwd {name} T{T}", out, o2)
            (o2 * gy).sum().backward()
            check(f"hashgrid bwd {name} T{T}", g_ref, t2.grad, 1e-7)
            npz(f"hashgrid_{name}_T{T}", u=u, table=table, scalings=sc, log2_T=T, levels=L, features=F,
                out=out, grad_out=gy, grad_table=g_ref)


def fx_mlp():
    specs = {"prop": (10, 2, 16, 1, None), "base": (32, 2, 64, 16, None), "head": (31, 3, 64, 3, "sigmoid"),
             "sam": (192, 2, 256, 256, None), "clipseg": (192, 2, 256, 192, None)}
    for name, (ind, nl, width, outd, act) in specs.items():
        for bias in ((False, True) if width <= 64 else (False,)):
            torch.manual_seed(7)
            m = MLP(in_dim=ind, num_layers=nl, layer_width=width, out_dim=outd,
                    activation=torch.nn.ReLU(), out_activation=torch.nn.Sigmoid() if act else None)
            if not bias:
                for l in m.layers:
                    l.bias.data.zero_()
            gen = torch.Generator().manual_seed(8)
            x = (torch.randn((200, ind), generator=gen) * 0.5).requires_grad_(True)
            y = m(x)
            gy = torch.randn(y.shape, generator=gen)
            (y * gy).sum().backward()
            ws = [l.weight.detach().clone().requires_grad_(True) for l in m.layers]
            bs = [l.bias.detach().clone().requires_grad_(True) for l in m.layers]
            x2 = x.detach().clone().requires_grad_(True)
            y2 = O.mlp_fwd(x2, ws, bs if bias else None, act)
            check(f"mlp fwd {name}", y, y2)
            (y2 * gy).sum().backward()
            check(f"mlp gx {name}", x.grad, x2.grad, 1e-7)
            arrays = dict(x=x, y=y, grad_y=gy, grad_x=x.grad, n_layers=nl, out_act=act or "none")
            for i, l in enumerate(m.layers):
                check(f"mlp gw{i} {name}", l.weight.grad, ws[i].grad, 1e-6)
                arrays[f"w{i}"] = l.weight
                arrays[f"gw{i}"] = l.weight.grad
                if bias:
                    arrays[f"b{i}"] = l.bias
                    arrays[f"gb{i}"] = l.bias.grad
            npz(f"mlp_{name}_{'bias' if bias else 'nobias'}", **arrays)


def fx_sh():
    _, d = O.synthetic_rays(300, 9)
    ref = components_from_spherical_harmonics(4, d)
    check("sh16", ref, O.sh16(d))
    npz("sh16", directions=d, sh=ref)


def fx_weights():
    gen = torch.Generator().manual_seed(21)
    R, S = 48, 48
    rb, o, d = make_bundle(R, 4, True)
    torch.manual_seed(1)
    rs = UniformLinDispPiecewiseSampler(num_samples=S, single_jitter=True)(rb, num_samples=S)
    dens = torch.exp(torch.randn((R, S, 1), generator=gen) * 2.0)
    dens[0] = 0.0
    dens[1] = 1e30  # alpha = 1 on the first sample, T underflows to 0
    dens[2, :, 0] = float("inf")  # 0 * inf -> nan_to_num path
    dens[3, 10:] = 1e38
    dens = dens.clone().requires_grad_(True)
    w = rs.get_weights(dens)
    gw = torch.randn(w.shape, generator=gen)
    finite_rows = [i for i in range(R) if i != 2]
    (w[finite_rows] * gw[finite_rows]).sum().backward()
    d2 = dens.detach()[..., 0].clone().requires_grad_(True)
    w2 = O.weights_from_density(d2, rs.deltas[..., 0])
    check("weights", w[..., 0], w2)
    (w2[finite_rows] * gw[finite_rows, :, 0]).sum().backward()
    check("weights grad", torch.nan_to_num(dens.grad[..., 0]), torch.nan_to_num(d2.grad), 1e-6)
    # trunc_exp fwd/bwd
    x = (torch.randn((256,), generator=gen) * 10).requires_grad_(True)
    y = ref_trunc_exp(x)
    y.sum().backward()
    x2 = x.detach().clone().requires_grad_(True)
    y2 = O.trunc_exp(x2)
    y2.sum().backward()
    check("trunc_exp", y, y2)
    check("trunc_exp grad", x.grad, x2.grad)
    npz("weights", density=dens[..., 0], deltas=rs.deltas[..., 0], weights=w[..., 0], grad_w=gw[..., 0],
        grad_density=torch.nan_to_num(dens.grad[..., 0]), finite_rows=np.array(finite_rows),
        te_x=x, te_y=y, te_grad=x.grad)


def fx_pdf():
    R, P, S = 64, 64, 48
    for mode in ("train", "eval"):
        training = mode == "train"
        rb, o, d = make_bundle(R, 6, training)
        ini = UniformLinDispPiecewiseSampler(num_samples=P, single_jitter=True)
        ini.train(training)
        torch.manual_seed(2)
        rs = ini(rb, num_samples=P)
        gen = torch.Generator().manual_seed(33)
        w = torch.rand((R, P, 1), generator=gen) ** 4
        w[0] = 0.0  # all-zero ray -> eps padding path
        w[1, :, 0] = 0.0
        w[1, 17, 0] = 1.0  # a delta
        w[2] = 1e-9
        pdf = PDFSampler(include_original=False, single_jitter=True)
        pdf.train(training)
        torch.manual_seed(44)
        out = pdf(rb, rs, w, num_samples=S)
        torch.manual_seed(44)
        u_rand = torch.rand((R, 1)) if training else None
        sb_in = torch.cat([rs.spacing_starts[..., 0], rs.spacing_ends[..., -1:, 0]], -1)
        sb = torch.cat([out.spacing_starts[..., 0], out.spacing_ends[..., -1:, 0]], -1)
        eb = torch.cat([out.frustums.starts[..., 0], out.frustums.ends[..., -1:, 0]], -1)
        osb = O.pdf_resample(w[..., 0], sb_in, S, u_rand)
        check("pdf sbins " + mode, sb, osb)
        check("pdf ebins " + mode, eb, O.s_to_euclid(osb, rb.nears, rb.fars))
        npz(f"pdf_{mode}", weights=w[..., 0], sbins_in=sb_in, nears=rb.nears, fars=rb.fars,
            u_rand=u_rand if training else np.zeros((0,)), sbins=sb, ebins=eb, num_samples=S)


def fx_render():
    R, S = 64, 48
    gen = torch.Generator().manual_seed(55)
    rb, o, d = make_bundle(R, 8, True)
    torch.manual_seed(3)
    rs = UniformLinDispPiecewiseSampler(num_samples=S, single_jitter=True)(rb, num_samples=S)
    eb = torch.cat([rs.frustums.starts[..., 0], rs.frustums.ends[..., -1:, 0]], -1)
    dens = torch.exp(torch.randn((R, S, 1), generator=gen) * 3.0)
    dens[0] = 0.0
    dens[1] = 1e-3
    w = rs.get_weights(dens)
    rgb = torch.rand((R, S, 3), generator=gen) * 1.4 - 0.2
    rgb[5, 7, 1] = float("nan")
    arrays = dict(weights=w[..., 0], rgb_samples=rgb, ebins=eb)
    for mode in ("train", "eval"):
        r = RGBRenderer(background_color="last_sample")
        r.train(mode == "train")
        out = r(rgb=rgb.clone(), weights=w)
        check("rgb " + mode, out, O.render_rgb(rgb.clone(), w[..., 0], mode == "train"))
        arrays["rgb_" + mode] = out
    acc = AccumulationRenderer()(weights=w)
    dep = DepthRenderer()(weights=w, ray_samples=rs)
    check("acc", acc, O.render_accumulation(w[..., 0]))
    check("depth", dep, O.render_depth_median(w[..., 0], eb))
    npz("render", accumulation=acc, depth=dep, **arrays)


def ref_topk_mean(weights, k, temp, feats):
    """The selection / sharpening / mean lines of the reference model, behaviourally (sam_model.py:244-255,136)."""
    sam_weights, best_ids = torch.topk(weights, k, dim=-2, sorted=False)
    sam_weights = sam_weights**temp
    sam_weights = sam_weights / sam_weights.sum(dim=-2, keepdim=True)
    f = torch.gather(feats, -2, best_ids.expand(*best_ids.shape[:-1], feats.shape[-1]))
    return sam_weights, best_ids, torch.sum(sam_weights.detach() * f, dim=-2)


def fx_topk():
    R, S, K, C = 64, 48, 16, 24
    gen = torch.Generator().manual_seed(66)
    w = torch.rand((R, S, 1), generator=gen) ** 3
    w[0] = 0.0  # all-zero ray -> 0/0 = NaN row
    w[1] = 1e-8  # w^10 underflows -> NaN row
    feats = torch.randn((R, S, C), generator=gen)
    sw, ids, mean = ref_topk_mean(w, K, 10.0, feats)
    osw, oids = O.topk_sharpen(w[..., 0], K, 10.0)
    # order of topk(sorted=False) is unspecified: compare as sets via sort
    s_ref, _ = torch.sort(ids[..., 0], dim=-1)
    s_or, _ = torch.sort(oids, dim=-1)
    rows = list(range(2, R))
    check("topk ids", s_ref[rows], s_or[rows])
    omean = O.feature_mean(torch.gather(feats, 1, oids[..., None].expand(-1, -1, C)), osw)
    check("topk mean", mean, omean, 1e-6)
    tgt = torch.randn((R, C), generator=gen)
    un = torch.nn.functional.mse_loss(mean, tgt, reduction="none")
    loss = un.mean(dim=-1).nanmean()
    check("feature loss", loss, O.feature_loss(omean, tgt), 1e-6)
    npz("topk", weights=w[..., 0], feats=feats, k=K, temperature=10.0, sam_weights=sw[..., 0], ids=ids[..., 0],
        mean=mean, target=tgt, loss=loss, nan_rows=np.array([0, 1]))


class _RS:
    """Minimal stand-in for RaySamples in the loss fixtures: the reference losses only read spacing_starts/ends."""

    def __init__(self, sb):
        self.spacing_starts = sb[:, :-1, None]
        self.spacing_ends = sb[:, 1:, None]


def fx_losses():
    R, P, S = 64, 64, 48
    gen = torch.Generator().manual_seed(77)
    sb_p = torch.sort(torch.rand((R, P + 1), generator=gen), dim=-1)[0]
    sb_p[:, 0], sb_p[:, -1] = 0.0, 1.0
    sb_f = torch.sort(torch.rand((R, S + 1), generator=gen), dim=-1)[0]
    w_p = (torch.rand((R, P), generator=gen) ** 2 / P * 3).requires_grad_(True)
    w_f = (torch.rand((R, S), generator=gen) ** 2 / S * 3).requires_grad_(True)
    li = ref_interlevel([w_p[..., None], w_f[..., None]], [_RS(sb_p), _RS(sb_f)])
    ld = ref_distortion([w_p[..., None], w_f[..., None]], [_RS(sb_p), _RS(sb_f)])
    (li + ld).backward()
    wp2 = w_p.detach().clone().requires_grad_(True)
    wf2 = w_f.detach().clone().requires_grad_(True)
    oi = O.interlevel_loss(sb_f, wf2, sb_p, wp2)
    od = O.distortion_loss(sb_f, wf2)
    check("interlevel", li, oi, 1e-7)
    check("distortion", ld, od, 1e-7)
    (oi + od).backward()
    check("grad wp", w_p.grad, wp2.grad, 1e-7)
    check("grad wf", w_f.grad, wf2.grad, 1e-7)
    npz("losses", sbins_prop=sb_p, sbins_fine=sb_f, w_prop=w_p, w_fine=w_f, interlevel=li, distortion=ld,
        grad_w_prop=w_p.grad, grad_w_fine=w_f.grad)


# ---------------------------------------------------------------------------------------------
def fx_ministep():
    """One composed train step through the reference's own components."""
    _ministep("ministep", 64, 11, 48, 0.1, 200)


def _ministep(name, R, T, S, table_scale, anneal_step):
    cfg = O.PathConfig(num_proposal_samples=64, num_nerf_samples=S, num_sam_samples=16, patch_size=4).small(T)
    params = O.init_params(cfg, seed=0, table_scale=table_scale)
    o, d = O.synthetic_rays(R, 0)
    batch = O.synthetic_batch(cfg, R, 1)
    linf, l2 = SceneContraction(order=float("inf")), SceneContraction()

    def ref_enc(g, table):
        e = HashEncoding(num_levels=g.levels, min_res=g.min_res, max_res=g.max_res, log2_hashmap_size=g.log2_T,
                         features_per_level=g.features, implementation="torch")
        e.hash_table = torch.nn.Parameter(table.clone())
        return e

    def ref_mlp(ws, act=None):
        m = MLP(in_dim=ws[0].shape[1], num_layers=len(ws), layer_width=ws[0].shape[0], out_dim=ws[-1].shape[0],
                activation=torch.nn.ReLU(), out_activation=torch.nn.Sigmoid() if act else None)
        for l, w in zip(m.layers, ws):
            l.weight = torch.nn.Parameter(w.clone())
            l.bias.data.zero_()
            l.bias.requires_grad_(False)
        return m

    mods = {
        "prop_enc": ref_enc(cfg.prop_grid, params["prop_table"]),
        "prop_mlp": ref_mlp([params["prop_w0"], params["prop_w1"]]),
        "field_enc": ref_enc(cfg.field_grid, params["field_table"]),
        "base_mlp": ref_mlp([params["base_w0"], params["base_w1"]]),
        "head_mlp": ref_mlp([params["head_w0"], params["head_w1"], params["head_w2"]], "sigmoid"),
    }
    for head in ("sam", "clipseg"):
        for i, g in enumerate(cfg.feat_grids):
            mods[f"{head}_enc{i}"] = ref_enc(g, params[f"{head}_table{i}"])
        mods[f"{head}_mlp"] = ref_mlp([params[f"{head}_w0"], params[f"{head}_w1"]])
    conv = torch.nn.Sequential(torch.nn.Conv2d(256, 256, 3, padding=1), torch.nn.ReLU(inplace=True),
                               torch.nn.Conv2d(256, 256, 3, padding=1))
    conv[0].weight.data, conv[0].bias.data = params["conv0_w"].clone(), params["conv0_b"].clone()
    conv[2].weight.data, conv[2].bias.data = params["conv1_w"].clone(), params["conv1_b"].clone()

    def density_fn(positions):  # HashMLPDensityField.get_density dataflow, density_fields.py:102-125
        p = (linf(positions) + 2.0) / 4.0
        sel = ((p > 0.0) & (p < 1.0)).all(dim=-1)
        p = p * sel[..., None]
        raw = mods["prop_mlp"](mods["prop_enc"].pytorch_fwd(p.view(-1, 3))).view(*positions.shape[:-1], -1)
        return ref_trunc_exp(raw) * sel[..., None]

    rb = RayBundle(origins=o, directions=d, pixel_area=torch.full((R, 1), 1e-6),
                   camera_indices=torch.zeros((R, 1), dtype=torch.long))
    rb = NearFarCollider(near_plane=0.05, far_plane=1000.0)(rb)  # training mode default
    sampler = ProposalNetworkSampler(num_proposal_samples_per_ray=(cfg.num_proposal_samples,),
                                     num_nerf_samples_per_ray=cfg.num_nerf_samples,
                                     num_proposal_network_iterations=1, single_jitter=True)
    anneal = O.proposal_anneal(anneal_step)
    sampler.set_anneal(anneal)
    torch.manual_seed(123)
    ray_samples, weights_list, ray_samples_list = sampler(rb, density_fns=[density_fn])
    torch.manual_seed(123)
    t_rand, u_rand = torch.rand((R, 1)), torch.rand((R, 1))
    ray_samples_list.append(ray_samples)
    # main field (TCNNNerfactoField dataflow, nerfacto_field.py:242-351, appearance embedding off)
    pos = ray_samples.frustums.get_positions()
    p = (linf(pos) + 2.0) / 4.0
    sel = ((p > 0.0) & (p < 1.0)).all(dim=-1)
    p = p * sel[..., None]
    h = mods["base_mlp"](mods["field_enc"].pytorch_fwd(p.view(-1, 3))).view(*pos.shape[:-1], -1)
    raw, geo = torch.split(h, [1, cfg.geo_feat_dim], dim=-1)
    density = ref_trunc_exp(raw) * sel[..., None]
    sh = components_from_spherical_harmonics(4, ray_samples.frustums.directions.expand(*pos.shape[:-1], 3).reshape(-1, 3))
    rgb_s = mods["head_mlp"](torch.cat([sh, geo.reshape(-1, cfg.geo_feat_dim)], dim=-1)).view(*pos.shape[:-1], 3)
    weights = ray_samples.get_weights(density)
    weights_list.append(weights)
    rgb = RGBRenderer(background_color="last_sample")(rgb=rgb_s, weights=weights)
    depth = DepthRenderer()(weights=weights, ray_samples=ray_samples)
    acc = AccumulationRenderer()(weights=weights)
    prop_depth = DepthRenderer()(weights=weights_list[0], ray_samples=ray_samples_list[0])
    # feature branch (sam_model.py:243-277)
    sam_weights, best_ids = torch.topk(weights, cfg.num_sam_samples, dim=-2, sorted=False)
    sam_weights = sam_weights**cfg.sharpening_temperature
    sam_weights = sam_weights / sam_weights.sum(dim=-2, keepdim=True)

    def gather_fn(tens):
        return torch.gather(tens, -2, best_ids.expand(*best_ids.shape[:-1], tens.shape[-1]))

    dataclass_fn = lambda dc: dc._apply_fn_to_fields(gather_fn, dataclass_fn)  # noqa: E731
    sam_samples = ray_samples._apply_fn_to_fields(gather_fn, dataclass_fn)
    fpos = (l2(sam_samples.frustums.get_positions().detach()) + 2.0) / 4.0
    outs = {}
    for head in ("sam", "clipseg"):
        x = torch.cat([mods[f"{head}_enc{i}"].pytorch_fwd(fpos.view(-1, 3)) for i in range(2)], dim=-1)
        f = mods[f"{head}_mlp"](x).view(*fpos.shape[:-1], -1)
        outs[head] = torch.sum(sam_weights.detach() * f, dim=-2)
    sam_raw = outs["sam"]
    feat = sam_raw.reshape(-1, 4, 4, 256).permute(0, 3, 1, 2)
    outs["sam"] = conv(feat).mean(dim=[2, 3])
    # losses
    l_rgb = torch.nn.MSELoss()(batch["image"], rgb)
    l_inter = ref_interlevel(weights_list, ray_samples_list)
    l_dist = 0.002 * ref_distortion(weights_list, ray_samples_list)
    l_sam = torch.nn.functional.mse_loss(outs["sam"], batch["sam"], reduction="none").mean(dim=-1).nanmean()
    l_clip = torch.nn.functional.mse_loss(outs["clipseg"], batch["clipseg"], reduction="none").mean(dim=-1).nanmean()
    loss = l_rgb + l_inter + l_dist + l_sam + l_clip
    loss.backward()

    # ---- oracle on the same inputs
    op = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    oo = O.forward(op, cfg, o, d, True, t_rand, u_rand, anneal)
    old = O.loss_dict(oo, batch, cfg)
    oloss = sum(old.values())
    oloss.backward()
    sb_f = torch.cat([ray_samples.spacing_starts[..., 0], ray_samples.spacing_ends[..., -1:, 0]], -1)
    check("mini sbins_fine", sb_f, oo["sbins_fine"])
    check("mini w_prop", weights_list[0][..., 0], oo["weights_prop"], 1e-7)
    check("mini w_fine", weights[..., 0], oo["weights_fine"], 1e-7)
    check("mini rgb", rgb, oo["rgb"], 1e-6)
    check("mini depth", depth, oo["depth"])
    check("mini acc", acc, oo["accumulation"], 1e-6)
    check("mini sam", outs["sam"], oo["sam"], 1e-6)
    check("mini clipseg", outs["clipseg"], oo["clipseg"], 1e-6)
    check("mini loss", loss, oloss, 1e-6)
    ref_grads = {
        "prop_table": mods["prop_enc"].hash_table.grad, "field_table": mods["field_enc"].hash_table.grad,
        "prop_w0": mods["prop_mlp"].layers[0].weight.grad, "prop_w1": mods["prop_mlp"].layers[1].weight.grad,
        "base_w0": mods["base_mlp"].layers[0].weight.grad, "base_w1": mods["base_mlp"].layers[1].weight.grad,
        "head_w0": mods["head_mlp"].layers[0].weight.grad, "head_w1": mods["head_mlp"].layers[1].weight.grad,
        "head_w2": mods["head_mlp"].layers[2].weight.grad,
        "conv0_w": conv[0].weight.grad, "conv0_b": conv[0].bias.grad,
        "conv1_w": conv[2].weight.grad, "conv1_b": conv[2].bias.grad,
    }
    for head in ("sam", "clipseg"):
        for i in range(2):
            ref_grads[f"{head}_table{i}"] = mods[f"{head}_enc{i}"].hash_table.grad
            ref_grads[f"{head}_w{i}"] = mods[f"{head}_mlp"].layers[i].weight.grad
    arrays = {}
    for k, g in ref_grads.items():
        check("mini grad " + k, g, op[k].grad, 2e-6)
        arrays["grad_" + k] = g
    npz(name, seed_params=0, table_scale=table_scale, log2_T=T, num_rays=R, seed_rays=0, seed_batch=1,
        P=cfg.num_proposal_samples, S=cfg.num_nerf_samples, K=cfg.num_sam_samples, patch=cfg.patch_size,
        anneal=anneal, t_rand=t_rand, u_rand=u_rand, origins=o, directions=d,
        sbins_prop=oo["sbins_prop"], sbins_fine=sb_f, w_prop=weights_list[0][..., 0], w_fine=weights[..., 0],
        rgb=rgb, depth=depth, accumulation=acc, prop_depth_0=prop_depth, sam_raw=sam_raw, sam=outs["sam"],
        clipseg=outs["clipseg"], sam_ids=best_ids[..., 0], sam_weights=sam_weights[..., 0],
        rgb_loss=l_rgb, interlevel_loss=l_inter, distortion_loss=l_dist, sam_loss=l_sam, clipseg_loss=l_clip,
        loss=loss, **arrays)


# ---------------------------------------------------------------------------------------------
def fx_batch_builder():
    """PatchPixelSampler / PixelSampler, RayGenerator + Cameras (pinhole), FeatureDataloader.
    cameras.py imports cv2 / torchvision at module level only: empty stand-in modules satisfy the import."""
    import types
    for name in ("cv2", "torchvision", "torchvision.transforms", "torchvision.transforms.functional"):
        sys.modules.setdefault(name, types.ModuleType(name))
    from nerfstudio.cameras.cameras import Cameras, CameraType
    from nerfstudio.data.pixel_samplers import PatchPixelSampler, PixelSampler
    from samnerf.data.feature_loader import FeatureDataloader

    N, H, W, p, R = 3, 42, 65, 4, 256
    g = torch.Generator().manual_seed(3)
    # LLFF-style forward-facing poses: small rotations about x/y, translations in a 0.5 box, random intrinsics per camera
    ang = (torch.rand((N, 2), generator=g) - 0.5) * 0.6
    c2w = torch.zeros((N, 3, 4))
    for i in range(N):
        ca, sa, cb, sb = torch.cos(ang[i, 0]), torch.sin(ang[i, 0]), torch.cos(ang[i, 1]), torch.sin(ang[i, 1])
        rx = torch.tensor([[1, 0, 0], [0, ca, -sa], [0, sa, ca]])
        ry = torch.tensor([[cb, 0, sb], [0, 1, 0], [-sb, 0, cb]])
        c2w[i, :, :3] = ry @ rx
        c2w[i, :, 3] = (torch.rand((3,), generator=g) - 0.5)
    fx = 50.0 + 20.0 * torch.rand((N,), generator=g)
    fy = 50.0 + 20.0 * torch.rand((N,), generator=g)
    cx = W / 2 + torch.rand((N,), generator=g)
    cy = H / 2 + torch.rand((N,), generator=g)
    cams = Cameras(camera_to_worlds=c2w, fx=fx, fy=fy, cx=cx, cy=cy, width=W, height=H,
                   camera_type=CameraType.PERSPECTIVE)
    images = torch.rand((N, H, W, 3), generator=g)
    fh, fw = 42, 64  # SAM map of a 42x65-ish image (get_feature_size)
    sam = torch.randn((N, fh, fw, 8), generator=g)       # 8 channels stand in for 256: the index math is what is pinned
    clip = torch.randn((N, 32, 32, 6), generator=g)

    # --- samplers: seed, draw u ourselves, re-seed, let the reference draw the same numbers
    torch.manual_seed(11)
    u_patch = torch.rand((R // (p * p), 3))
    torch.manual_seed(11)
    ref_patch = PatchPixelSampler(R, patch_size=p).sample_method(R, N, H, W)
    check("patch_pixel_indices", O.patch_pixel_indices(u_patch, N, H, W, p), ref_patch)
    torch.manual_seed(12)
    u_pix = torch.rand((R, 3))
    torch.manual_seed(12)
    ref_pix = PixelSampler(R).sample_method(R, N, H, W)
    check("pixel_indices", O.pixel_indices(u_pix, N, H, W), ref_pix)
    # full collate path (image gather + absolute camera indices)
    torch.manual_seed(11)
    pb = PatchPixelSampler(R, patch_size=p).sample({"image": images, "image_idx": torch.arange(N)})
    check("collate indices", pb["indices"], ref_patch)
    # --- rays
    # RayGenerator.forward (ray_generators.py:44-63) spelled out -- the class itself imports the tyro-based config
    # stack; with camera_optimizer mode "off" (samconfigs.py:74,128) its pose correction is the identity
    coords = cams.get_image_coords()[ref_patch[:, 1], ref_patch[:, 2]]
    rb = cams.generate_rays(camera_indices=ref_patch[:, 0].unsqueeze(-1), coords=coords)
    o_o, o_d, o_pa, o_ci = O.generate_rays(ref_patch, c2w, fx, fy, cx, cy)
    check("ray origins", o_o, rb.origins)
    check("ray directions", o_d, rb.directions)
    check("ray pixel_area", o_pa, rb.pixel_area)
    check("ray camera_indices", o_ci, rb.camera_indices)
    # --- feature loaders (constructed without files: the loaded tensors are set directly)
    def loader(feat, patch):
        fl = FeatureDataloader.__new__(FeatureDataloader)
        fl.device, fl.features, fl.image_shape, fl.patch_size = "cpu", feat, [H, W], patch
        return fl
    centers = ref_patch.reshape(-1, p, p, 3)[:, p // 2, p // 2, :]
    ref_sam = loader(sam, p)(centers)
    ref_clip = loader(clip, 1)(ref_patch)
    check("sam gather", O.gather_features(sam, centers, (H, W)), ref_sam)
    check("clipseg gather", O.gather_features(clip, ref_patch, (H, W)), ref_clip)
    (bo, bd, bpa, bci), batch = O.build_batch(u_patch, images, c2w, fx, fy, cx, cy, p, sam, clip)
    check("batch image", batch["image"], pb["image"])
    check("batch sam", batch["sam"], ref_sam)
    check("batch clipseg", batch["clipseg"], ref_clip)
    check("batch dirs", bd, rb.directions)
    # --- dataparser pose normalisation (camera_utils.auto_orient_and_center_poses "up" / "none" + auto-scale)
    from nerfstudio.cameras import camera_utils
    poses4 = torch.cat([c2w, torch.tensor([0.0, 0.0, 0.0, 1.0]).expand(N, 1, 4)], dim=1)
    poses4[:, :3, 3] += torch.tensor([3.0, -2.0, 5.0])
    up_poses, up_tf = camera_utils.auto_orient_and_center_poses(poses4.clone(), method="up", center_poses=True)
    none_poses, none_tf = camera_utils.auto_orient_and_center_poses(poses4.clone(), method="none", center_poses=True)
    npz("batch_builder", poses4=poses4, up_poses=up_poses, up_tf=up_tf, none_poses=none_poses, none_tf=none_tf, N=N, H=H, W=W, p=p, R=R, c2w=c2w, fx=fx, fy=fy, cx=cx, cy=cy, images=images, sam=sam, clip=clip,
        u_patch=u_patch, u_pix=u_pix, patch_indices=ref_patch, pix_indices=ref_pix, origins=rb.origins,
        directions=rb.directions, pixel_area=rb.pixel_area, camera_indices=rb.camera_indices, batch_image=pb["image"],
        batch_sam=ref_sam, batch_clipseg=ref_clip)


# ---------------------------------------------------------------------------------------------
def fx_vit():
    """the SAM image encoder.  The reference's two model files are loaded as a synthetic package (its
    own package __init__ pulls in the predictor / torchvision stack); small configuration with one windowed block (window 5
    on a 14 x 14 grid: padded to 15) and one global block, random rel-pos tables."""
    import importlib.util
    import types
    from functools import partial
    from oracle import vit_oracle as V
    pkg = types.ModuleType("ref_sa_modeling")
    pkg.__path__ = [os.path.join(REF, "samnerf/segment_anything/modeling")]
    sys.modules["ref_sa_modeling"] = pkg
    for m in ("common", "image_encoder"):
        spec = importlib.util.spec_from_file_location(f"ref_sa_modeling.{m}", os.path.join(pkg.__path__[0], m + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"ref_sa_modeling.{m}"] = mod
        spec.loader.exec_module(mod)
    Ref = sys.modules["ref_sa_modeling.image_encoder"].ImageEncoderViT
    cfg = V.ViTConfig(img_size=224, patch_size=16, embed_dim=32, depth=2, num_heads=2, mlp_ratio=2.0, out_chans=16,
                      window_size=5, global_attn_indexes=(1,))
    ref = Ref(img_size=cfg.img_size, patch_size=cfg.patch_size, embed_dim=cfg.embed_dim, depth=cfg.depth,
              num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio, out_chans=cfg.out_chans, qkv_bias=True,
              norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), use_rel_pos=True, window_size=cfg.window_size,
              global_attn_indexes=cfg.global_attn_indexes)
    sd = V.init_weights(cfg, seed=4)
    missing = ref.load_state_dict(sd, strict=True)
    ref.eval()
    x = torch.randn((2, 3, 224, 224), generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        y_ref = ref(x)
        t0 = ref.patch_embed(x) + ref.pos_embed
        t1 = ref.blocks[0](t0)
        t2 = ref.blocks[1](t1)
        y, trace = V.forward(sd, x, cfg, return_tokens=True)
    check("vit tokens after patch embed", trace[0], t0, 1e-6)
    check("vit block 0 (windowed, padded)", trace[1], t1, 2e-6)
    check("vit block 1 (global)", trace[2], t2, 2e-6)
    check("vit output", y, y_ref, 2e-6)
    npz("vit_small", x=x, y=y_ref, t0=t0, t1=t1, t2=t2, **{"w:" + k: v for k, v in sd.items()})


def _reference_function(path, cls, name):
    """The source of one method of a reference class, compiled on its own (the module imports torchvision, which this
    container lacks; the method itself is plain torch)."""
    import ast
    src = open(os.path.join(REF, path)).read()
    tree = ast.parse(src)
    node = next(f for c in tree.body if isinstance(c, ast.ClassDef) and c.name == cls
                for f in c.body if isinstance(f, ast.FunctionDef) and f.name == name)
    mod = ast.Module(body=[node], type_ignores=[])
    ns = {"np": np, "torch": torch, "math": __import__("math"), "F": torch.nn.functional}
    exec(compile(mod, path, "exec"), ns)
    return ns[name]


def fx_sam_preprocess():
    """Sam.preprocess (segment_anything/modeling/sam.py:164-174: (x - pixel_mean) / pixel_std, zero-pad right / bottom to the
    encoder's square input) -- the step between SamPredictor.set_torch_image (predictor.py:70-97) and the image encoder -- run
    with the reference's own method body on a stand-in model (its module imports the prompt / mask stack).  uint8 and float
    inputs, landscape and portrait, the pixel statistics of build_sam.py:99-100."""
    from types import SimpleNamespace
    preprocess = _reference_function("samnerf/segment_anything/modeling/sam.py", "Sam", "preprocess")
    mean = torch.Tensor([123.675, 116.28, 103.53]).view(-1, 1, 1)
    std = torch.Tensor([58.395, 57.12, 57.375]).view(-1, 1, 1)
    me = SimpleNamespace(pixel_mean=mean, pixel_std=std, image_encoder=SimpleNamespace(img_size=64))
    g = torch.Generator().manual_seed(11)
    a = torch.randint(0, 256, (1, 3, 43, 64), generator=g, dtype=torch.uint8)
    b = torch.rand((2, 3, 64, 48), generator=g) * 255.0
    c = torch.rand((1, 3, 64, 64), generator=g) * 255.0
    npz("sam_preprocess", mean=mean.view(-1), std=std.view(-1), a=a, a_out=preprocess(me, a), b=b, b_out=preprocess(me, b), c=c,
        c_out=preprocess(me, c))


def fx_eval_regroup():
    """The eval-path ray regroup of samnerf/sam_model.py:371-398 run with the reference's own RayBundle / TensorDataclass
    (nerfstudio/utils/tensor_dataclass.py: __getitem__, reshape, _apply_fn_to_fields, get_row_major_sliced_ray_bundle) and
    get_feature_size; SamPredictor.set_feature's zero-pad (segment_anything/predictor.py:100-127)."""
    from types import SimpleNamespace
    from samnerf.sam_utils import get_feature_size
    arrays = {}
    cases = [(37, 53), (48, 31), (40, 64)]
    for ci, (H, W) in enumerate(cases):
        g = torch.Generator().manual_seed(100 + ci)
        ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
        origins = torch.stack([ys, xs, torch.zeros_like(ys)], -1).float()  # the pixel a ray came from, readable
        directions = torch.randn((H, W, 3), generator=g)
        cam = RayBundle(origins=origins, directions=directions, pixel_area=torch.rand((H, W, 1), generator=g),
                        camera_indices=torch.zeros((H, W, 1), dtype=torch.long))
        sz = cam.shape
        fh, fw = get_feature_size(H, W)
        p = 4
        h_indices = torch.linspace(0, sz[0] - 1, fh * p, dtype=torch.long)
        w_indices = torch.linspace(0, sz[1] - 1, fw * p, dtype=torch.long)
        hind, wind = torch.meshgrid(h_indices, w_indices)
        fb = cam[hind.flatten(), wind.flatten()]
        fb = fb.reshape((fh, p, fw, p))
        fb = fb._apply_fn_to_fields(lambda x: x.transpose(1, 2))
        chunk = 1000
        parts = [fb.get_row_major_sliced_ray_bundle(i, i + chunk) for i in range(0, len(fb), chunk)]
        fo = torch.cat([q.origins for q in parts])
        fd = torch.cat([q.directions for q in parts])
        fa = torch.cat([q.pixel_area for q in parts])
        check(f"feature regroup origins {H}x{W}", O.feature_ray_grid(origins, fh, fw, p), fo)
        check(f"feature regroup directions {H}x{W}", O.feature_ray_grid(directions, fh, fw, p), fd)
        h_indices = torch.linspace(0, sz[0] - 1, 32, dtype=torch.long)
        w_indices = torch.linspace(0, sz[1] - 1, 32, dtype=torch.long)
        hind, wind = torch.meshgrid(h_indices, w_indices)
        cb = cam[hind.flatten(), wind.flatten()].reshape((32, 32))
        co = torch.cat([cb.get_row_major_sliced_ray_bundle(i, i + chunk).origins for i in range(0, len(cb), chunk)])
        check(f"clipseg regroup {H}x{W}", O.clipseg_ray_grid(origins), co)
        arrays.update({f"c{ci}_hw": np.array([H, W, fh, fw, p]), f"c{ci}_directions": directions,
                       f"c{ci}_pixel_area": cam.pixel_area, f"c{ci}_feat_origins": fo, f"c{ci}_feat_directions": fd,
                       f"c{ci}_feat_pixel_area": fa, f"c{ci}_clip_origins": co})
    # set_feature: the reference's own function body on a stand-in predictor object
    set_feature = _reference_function("samnerf/segment_anything/predictor.py", "SamPredictor", "set_feature")
    for name, (C, fh, fw, hw) in {"land": (6, 45, 64, (360, 512)), "square": (6, 64, 64, (512, 512))}.items():
        feat = torch.randn((C, fh, fw), generator=torch.Generator().manual_seed(7))
        me = SimpleNamespace(model=SimpleNamespace(image_encoder=SimpleNamespace(img_size=1024), device="cpu"),
                             reset_image=lambda: None)
        set_feature(me, feat, hw)
        arrays.update({f"sf_{name}_in": feat, f"sf_{name}_out": me.features, f"sf_{name}_hw": np.array(hw),
                       f"sf_{name}_input_size": np.array(me.input_size)})
    # (portrait: the reference concatenates a [1,C,h,h-w] block along dim 2 and torch raises -- recorded, not reproduced)
    try:
        me = SimpleNamespace(model=SimpleNamespace(image_encoder=SimpleNamespace(img_size=1024), device="cpu"),
                             reset_image=lambda: None)
        set_feature(me, torch.zeros((6, 64, 40)), (512, 320))
        arrays["sf_portrait_reference_raises"] = np.array(0)
    except RuntimeError:
        arrays["sf_portrait_reference_raises"] = np.array(1)
    npz("eval_regroup", **arrays)


if __name__ == "__main__":
    only = set(sys.argv[1:])
    for fn in (fx_spacing, fx_contraction, fx_hashgrid, fx_mlp, fx_sh, fx_weights, fx_pdf, fx_render, fx_topk,
               fx_losses, fx_ministep, fx_batch_builder, fx_vit, fx_eval_regroup, fx_sam_preprocess):
        if only and fn.__name__ not in only:
            continue
        print(fn.__name__)
        fn()
    print("all reference outputs matched the oracle restatement; fixtures written.")

"""CPU ORACLE for the SAM-NeRF render-and-distill hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32) restatement of the reference algorithm for the path named
in BASELINE.json `north_star`.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg may import it -- never the product package (`segment-anything-in-nerf_amd/`),
which must fail loudly when its HIP extension is missing.

Every function cites the reference file:line (relative to the upstream repository root) whose
behaviour it restates.  It is written from the formulas, not copied.

Pinning: the reference ships no tests and no golden vectors for this path.  The restatement is
pinned by `tests/golden/make_golden.py`, which imports the reference's own torch components in
the build container (nerfstudio HashEncoding.pytorch_fwd, MLP, SceneContraction,
ProposalNetworkSampler/PDFSampler, RaySamples.get_weights, RGB/Depth/Accumulation renderers,
interlevel_loss, distortion_loss, components_from_spherical_harmonics, trunc_exp), asserts
agreement with these functions and writes the `.npz` fixtures that `tests/test_oracle_golden.py`
re-checks everywhere.  At the tiny-cuda-nn boundary (the library the reference calls on CUDA) the
reference pins nothing: there "parity unpinned" applies, and the torch-semantics hash grid
(nerfstudio/field_components/encodings.py:289-349) is the semantics implemented here.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

PRIME_Y = 2654435761
PRIME_Z = 805459861
EPS_OUTER = 1e-7  # nerfstudio/model_components/losses.py:34 (EPS)


# --------------------------------------------------------------------------------------------
# a2  NearFarCollider            nerfstudio/model_components/scene_colliders.py:183-189
# --------------------------------------------------------------------------------------------
def collider_near_far(num_rays: int, training: bool, near_plane: float = 0.05, far_plane: float = 1000.0):
    near = near_plane if training else 0.0
    nears = torch.full((num_rays, 1), near, dtype=torch.float32)
    fars = torch.full((num_rays, 1), far_plane, dtype=torch.float32)
    return nears, fars


# --------------------------------------------------------------------------------------------
# a3  UniformLinDispPiecewiseSampler / SpacedSampler   ray_samplers.py:79-126,223-246
# --------------------------------------------------------------------------------------------
def spacing_fn(x: torch.Tensor) -> torch.Tensor:
    return torch.where(x < 1, x / 2, 1 - 1 / (2 * x))  # ray_samplers.py:242


def spacing_fn_inv(y: torch.Tensor) -> torch.Tensor:
    return torch.where(y < 0.5, 2 * y, 1 / (2 - 2 * y))  # ray_samplers.py:243


def s_to_euclid(sbins: torch.Tensor, nears: torch.Tensor, fars: torch.Tensor) -> torch.Tensor:
    """spacing_to_euclidean_fn of ray_samplers.py:111-112."""
    s_near, s_far = spacing_fn(nears), spacing_fn(fars)
    return spacing_fn_inv(sbins * s_far + (1 - sbins) * s_near)


def sample_spacing(nears, fars, num_samples: int, t_rand: Optional[torch.Tensor]):
    """Initial proposal bins.  `t_rand` [R,1] is the single per-ray jitter (training) or None (eval).

    Returns (sbins [R,P+1], ebins [R,P+1]).   ray_samplers.py:101-124
    """
    R = nears.shape[0]
    bins = torch.linspace(0.0, 1.0, num_samples + 1)[None, :]
    if t_rand is not None:
        centers = (bins[..., 1:] + bins[..., :-1]) / 2.0
        upper = torch.cat([centers, bins[..., -1:]], -1)
        lower = torch.cat([bins[..., :1], centers], -1)
        bins = lower + (upper - lower) * t_rand
    else:
        bins = bins.expand(R, -1)
    return bins, s_to_euclid(bins, nears, fars)


# --------------------------------------------------------------------------------------------
# a1  Frustums.get_positions    nerfstudio/cameras/rays.py:48-57
# --------------------------------------------------------------------------------------------
def sample_positions(origins, directions, ebins):
    """[R,3],[R,3],[R,n+1] -> [R,n,3];  pos = o + d*(start+end)/2."""
    starts, ends = ebins[:, :-1, None], ebins[:, 1:, None]
    return origins[:, None, :] + directions[:, None, :] * (starts + ends) / 2


# --------------------------------------------------------------------------------------------
# a4  SceneContraction           nerfstudio/field_components/spatial_distortions.py:66-69
# --------------------------------------------------------------------------------------------
def contract(x: torch.Tensor, order: Optional[float]) -> torch.Tensor:
    mag = torch.linalg.norm(x, ord=order, dim=-1)[..., None]
    return torch.where(mag < 1, x, (2 - (1 / mag)) * (x / mag))


def normalize_positions(pos, order, use_selector: bool):
    """contraction -> (x+2)/4 -> optional (0,1) selector.  nerfacto_field.py:244-252, sam_field.py:116-118."""
    u = (contract(pos, order) + 2.0) / 4.0
    if use_selector:
        sel = ((u > 0.0) & (u < 1.0)).all(dim=-1)
        u = u * sel[..., None]
        return u, sel
    return u, None


# --------------------------------------------------------------------------------------------
# a6  HashEncoding (torch semantics)   nerfstudio/field_components/encodings.py:252-259,289-349
# --------------------------------------------------------------------------------------------
def hash_scalings(num_levels: int, min_res: int, max_res: int) -> torch.Tensor:
    """floor(min_res * g^l) evaluated the way the reference does (numpy f64 scalar ** int64 tensor -> fp32).

    encodings.py:252-254.  (128->512, L=12) gives 511 for the top level, not 512.
    """
    levels = torch.arange(num_levels)
    growth = np.exp((np.log(max_res) - np.log(min_res)) / (num_levels - 1))
    return torch.floor(min_res * growth**levels)


def hash_index(ix, iy, iz, log2_T: int, level_offsets):
    """(x ^ y*P1 ^ z*P2) mod 2^T + l*2^T in int64.   encodings.py:301-306."""
    h = ix.to(torch.int64) ^ (iy.to(torch.int64) * PRIME_Y) ^ (iz.to(torch.int64) * PRIME_Z)
    h = h % (1 << log2_T)
    return h + level_offsets


def hashgrid_fwd(u: torch.Tensor, table: torch.Tensor, scalings: torch.Tensor, log2_T: int) -> torch.Tensor:
    """u [N,3] in [0,1], table [L*2^T, F] -> [N, L*F] (level-major, feature-minor).   encodings.py:308-349."""
    L = scalings.numel()
    scaled = u[:, None, :] * scalings.view(-1, 1)  # [N,L,3]
    c = torch.ceil(scaled).to(torch.int32)
    f = torch.floor(scaled).to(torch.int32)
    o = scaled - f
    off = (torch.arange(L) * (1 << log2_T)).to(torch.int64)

    def g(a, b, cc):
        return table[hash_index(a, b, cc, log2_T, off)]

    cx, cy, cz = c[..., 0], c[..., 1], c[..., 2]
    fx, fy, fz = f[..., 0], f[..., 1], f[..., 2]
    f0 = g(cx, cy, cz)
    f1 = g(cx, fy, cz)
    f2 = g(fx, fy, cz)
    f3 = g(fx, cy, cz)
    f4 = g(cx, cy, fz)
    f5 = g(cx, fy, fz)
    f6 = g(fx, fy, fz)
    f7 = g(fx, cy, fz)
    ox, oy, oz = o[..., 0:1], o[..., 1:2], o[..., 2:3]
    f03 = f0 * ox + f3 * (1 - ox)
    f12 = f1 * ox + f2 * (1 - ox)
    f56 = f5 * ox + f6 * (1 - ox)
    f47 = f4 * ox + f7 * (1 - ox)
    f0312 = f03 * oy + f12 * (1 - oy)
    f4756 = f47 * oy + f56 * (1 - oy)
    out = f0312 * oz + f4756 * (1 - oz)
    return out.flatten(-2, -1)


# --------------------------------------------------------------------------------------------
# a7  MLP (bias optional; tcnn nets are bias-free)    nerfstudio/field_components/mlp.py:80-99
# --------------------------------------------------------------------------------------------
def mlp_fwd(x, weights: Sequence[torch.Tensor], biases: Optional[Sequence[Optional[torch.Tensor]]] = None,
            out_act: Optional[str] = None):
    """weights[i] is [out_i, in_i] (torch Linear layout); ReLU between layers."""
    n = len(weights)
    for i, w in enumerate(weights):
        b = None if biases is None else biases[i]
        x = torch.nn.functional.linear(x, w, b)
        if i < n - 1:
            x = torch.relu(x)
    if out_act == "sigmoid":
        x = torch.sigmoid(x)
    return x


# --------------------------------------------------------------------------------------------
# a13 SH basis, degree 4 (16 comps) on raw unit dirs   nerfstudio/utils/math.py:27-73
# --------------------------------------------------------------------------------------------
def sh16(d: torch.Tensor) -> torch.Tensor:
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    xx, yy, zz = x**2, y**2, z**2
    c = torch.zeros((*d.shape[:-1], 16), dtype=d.dtype)
    c[..., 0] = 0.28209479177387814
    c[..., 1] = 0.4886025119029199 * y
    c[..., 2] = 0.4886025119029199 * z
    c[..., 3] = 0.4886025119029199 * x
    c[..., 4] = 1.0925484305920792 * x * y
    c[..., 5] = 1.0925484305920792 * y * z
    c[..., 6] = 0.9461746957575601 * zz - 0.31539156525251999
    c[..., 7] = 1.0925484305920792 * x * z
    c[..., 8] = 0.5462742152960396 * (xx - yy)
    c[..., 9] = 0.5900435899266435 * y * (3 * xx - yy)
    c[..., 10] = 2.890611442640554 * x * y * z
    c[..., 11] = 0.4570457994644658 * y * (5 * zz - 1)
    c[..., 12] = 0.3731763325901154 * z * (5 * zz - 3)
    c[..., 13] = 0.4570457994644658 * x * (5 * zz - 1)
    c[..., 14] = 1.445305721320277 * z * (xx - yy)
    c[..., 15] = 0.5900435899266435 * x * (xx - 3 * yy)
    return c


# --------------------------------------------------------------------------------------------
# a8  trunc_exp     nerfstudio/field_components/activations.py:24-40
# --------------------------------------------------------------------------------------------
class _TruncExp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _TruncExp.apply


# --------------------------------------------------------------------------------------------
# a9  RaySamples.get_weights    nerfstudio/cameras/rays.py:141-163
# --------------------------------------------------------------------------------------------
def weights_from_density(density: torch.Tensor, deltas: torch.Tensor) -> torch.Tensor:
    """density, deltas [R,n] -> weights [R,n]."""
    dd = deltas * density
    alphas = 1 - torch.exp(-dd)
    acc = torch.cumsum(dd[:, :-1], dim=-1)
    acc = torch.cat([torch.zeros_like(acc[:, :1]), acc], dim=-1)
    trans = torch.exp(-acc)
    return torch.nan_to_num(alphas * trans)


# --------------------------------------------------------------------------------------------
# a10 PDFSampler (include_original=False, histogram_padding=0.01, single_jitter)  ray_samplers.py:298-367
# --------------------------------------------------------------------------------------------
def pdf_resample(weights: torch.Tensor, sbins: torch.Tensor, num_samples: int,
                 u_rand: Optional[torch.Tensor], histogram_padding: float = 0.01, eps: float = 1e-5):
    """weights [R,P] (already annealed), sbins [R,P+1] existing spacing bins, u_rand [R,1] or None (eval).

    Returns new spacing bins [R, num_samples+1] (detached).
    """
    num_bins = num_samples + 1
    w = weights + histogram_padding
    wsum = torch.sum(w, dim=-1, keepdim=True)
    padding = torch.relu(eps - wsum)
    w = w + padding / w.shape[-1]
    wsum = wsum + padding
    pdf = w / wsum
    cdf = torch.min(torch.ones_like(pdf), torch.cumsum(pdf, dim=-1))
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], dim=-1)
    u = torch.linspace(0.0, 1.0 - (1.0 / num_bins), steps=num_bins)
    u = u.expand(cdf.shape[0], num_bins)
    if u_rand is not None:
        u = u + u_rand / num_bins
    else:
        u = u + 1.0 / (2 * num_bins)
    u = u.contiguous()
    inds = torch.searchsorted(cdf.contiguous(), u, side="right")
    below = torch.clamp(inds - 1, 0, sbins.shape[-1] - 1)
    above = torch.clamp(inds, 0, sbins.shape[-1] - 1)
    cdf_g0 = torch.gather(cdf, -1, below)
    bins_g0 = torch.gather(sbins, -1, below)
    cdf_g1 = torch.gather(cdf, -1, above)
    bins_g1 = torch.gather(sbins, -1, above)
    t = torch.clip(torch.nan_to_num((u - cdf_g0) / (cdf_g1 - cdf_g0), 0), 0, 1)
    return (bins_g0 + t * (bins_g1 - bins_g0)).detach()


def proposal_anneal(step: int, max_iters: int = 1000, slope: float = 10.0) -> float:
    """nerfstudio/models/nerfacto.py:248-255."""
    frac = float(np.clip(step / max_iters, 0, 1))
    return (slope * frac) / ((slope - 1) * frac + 1)


# --------------------------------------------------------------------------------------------
# a14/a15 renderers   nerfstudio/model_components/renderers.py:97-112,133-139,222,260-270
# --------------------------------------------------------------------------------------------
def render_rgb(rgb: torch.Tensor, weights: torch.Tensor, training: bool) -> torch.Tensor:
    """rgb [R,S,3], weights [R,S] ; background 'last_sample'."""
    if not training:
        rgb = torch.nan_to_num(rgb)
    comp = torch.sum(weights[..., None] * rgb, dim=-2)
    acc = torch.sum(weights[..., None], dim=-2)
    comp = comp + rgb[..., -1, :] * (1.0 - acc)
    if not training:
        comp = torch.clamp(comp, 0.0, 1.0)
    return comp


def render_accumulation(weights: torch.Tensor) -> torch.Tensor:
    return torch.sum(weights, dim=-1, keepdim=True)


def render_depth_median(weights: torch.Tensor, ebins: torch.Tensor) -> torch.Tensor:
    steps = (ebins[:, :-1] + ebins[:, 1:]) / 2
    cw = torch.cumsum(weights, dim=-1)
    split = torch.full((weights.shape[0], 1), 0.5)
    idx = torch.searchsorted(cw.contiguous(), split, side="left")
    idx = torch.clamp(idx, 0, steps.shape[-1] - 1)
    return torch.gather(steps, -1, idx)


# --------------------------------------------------------------------------------------------
# a16 top-K select + sharpen    samnerf/sam_model.py:244-255
# --------------------------------------------------------------------------------------------
def topk_sharpen(weights: torch.Tensor, k: int, temperature: float = 10.0, ids: Optional[torch.Tensor] = None):
    """weights [R,S] -> (sam_weights [R,K] (may contain NaN rows), ids [R,K]).
    ids (test infrastructure only): a selection to use INSTEAD of torch.topk's, for rays whose K-th and (K+1)-th weights agree to
    fp32 rounding -- there both choices are answers of the reference's arithmetic and a comparison has to fix one of them."""
    if ids is None:
        w_k, ids = torch.topk(weights, k, dim=-1, sorted=False)
    else:
        w_k = torch.gather(weights, -1, ids)
    w_k = w_k**temperature
    w_k = w_k / w_k.sum(dim=-1, keepdim=True)
    return w_k, ids


def gather_samples(ebins: torch.Tensor, ids: torch.Tensor):
    """starts/ends of the selected samples: [R,K] each (the _apply_fn_to_fields gather of sam_model.py:250-255)."""
    return torch.gather(ebins[:, :-1], -1, ids), torch.gather(ebins[:, 1:], -1, ids)


# --------------------------------------------------------------------------------------------
# a18 MeanRenderer    samnerf/sam_model.py:126-137
# --------------------------------------------------------------------------------------------
def feature_mean(embeds: torch.Tensor, weights: torch.Tensor) -> torch.Tensor:
    """embeds [R,K,C], weights [R,K] (detached by the caller) -> [R,C]."""
    return torch.sum(weights[..., None] * embeds, dim=-2)


# --------------------------------------------------------------------------------------------
# a19-a21 losses     samnerf/sam_model.py:316-328, nerfacto.py:324-333, losses.py:46-143
# --------------------------------------------------------------------------------------------
def mse(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return torch.mean((a - b) ** 2)


def feature_loss(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    return ((pred - target) ** 2).mean(dim=-1).nanmean()


def _outer(t0_starts, t0_ends, t1_starts, t1_ends, y1):
    cy1 = torch.cat([torch.zeros_like(y1[..., :1]), torch.cumsum(y1, dim=-1)], dim=-1)
    idx_lo = torch.searchsorted(t1_starts.contiguous(), t0_starts.contiguous(), side="right") - 1
    idx_lo = torch.clamp(idx_lo, min=0, max=y1.shape[-1] - 1)
    idx_hi = torch.searchsorted(t1_ends.contiguous(), t0_ends.contiguous(), side="right")
    idx_hi = torch.clamp(idx_hi, min=0, max=y1.shape[-1] - 1)
    cy1_lo = torch.take_along_dim(cy1[..., :-1], idx_lo, dim=-1)
    cy1_hi = torch.take_along_dim(cy1[..., 1:], idx_hi, dim=-1)
    return cy1_hi - cy1_lo


def interlevel_loss(sbins_fine, w_fine, sbins_prop, w_prop) -> torch.Tensor:
    """One proposal level.  losses.py:78-120."""
    c = sbins_fine.detach()
    w = w_fine.detach()
    w_outer = _outer(c[..., :-1], c[..., 1:], sbins_prop[..., :-1], sbins_prop[..., 1:], w_prop)
    return torch.mean(torch.clip(w - w_outer, min=0) ** 2 / (w + EPS_OUTER))


def distortion_loss(sbins_fine, w_fine) -> torch.Tensor:
    """O(S^2) form exactly as losses.py:124-143."""
    t, w = sbins_fine, w_fine
    ut = (t[..., 1:] + t[..., :-1]) / 2
    dut = torch.abs(ut[..., :, None] - ut[..., None, :])
    inter = torch.sum(w * torch.sum(w[..., None, :] * dut, dim=-1), dim=-1)
    intra = torch.sum(w**2 * (t[..., 1:] - t[..., :-1]), dim=-1) / 3
    return torch.mean(inter + intra)


# --------------------------------------------------------------------------------------------
# a22 composed step (orchestration of sam_model.py:226-328 with torch-semantics components)
# --------------------------------------------------------------------------------------------
@dataclass
class GridSpec:
    levels: int
    features: int
    log2_T: int
    min_res: int
    max_res: int

    @property
    def out_dim(self) -> int:
        return self.levels * self.features

    @property
    def rows(self) -> int:
        return self.levels << self.log2_T

    def scalings(self) -> torch.Tensor:
        return hash_scalings(self.levels, self.min_res, self.max_res)


@dataclass
class PathConfig:
    """Shapes of the hot path.  Defaults = shipped samnerf_distill (samconfigs.py:104-164, sam_model.py:140-162)."""
    num_proposal_samples: int = 64
    num_nerf_samples: int = 32
    num_sam_samples: int = 16
    patch_size: int = 4
    distill_sam: bool = True
    use_clipseg: bool = True
    sharpening_temperature: float = 10.0
    geo_feat_dim: int = 15
    prop_grid: GridSpec = field(default_factory=lambda: GridSpec(5, 2, 17, 16, 128))
    field_grid: GridSpec = field(default_factory=lambda: GridSpec(16, 2, 19, 16, 2048))
    feat_grids: Tuple[GridSpec, ...] = field(
        default_factory=lambda: (GridSpec(12, 8, 19, 16, 128), GridSpec(12, 8, 19, 128, 512)))
    prop_hidden: int = 16
    base_hidden: int = 64
    head_hidden: int = 64
    feat_hidden: int = 256
    feat_hidden_layers: int = 1
    sam_dim: int = 256
    clipseg_dim: int = 192
    interlevel_loss_mult: float = 1.0
    distortion_loss_mult: float = 0.002
    sam_loss_weight: float = 1.0
    clipseg_loss_weight: float = 1.0
    near_plane: float = 0.05
    far_plane: float = 1000.0

    def small(self, log2_T: int) -> "PathConfig":
        """Same architecture with smaller tables (for fixtures / fast parity)."""
        import copy
        c = copy.deepcopy(self)
        c.prop_grid.log2_T = min(c.prop_grid.log2_T, log2_T)
        c.field_grid.log2_T = min(c.field_grid.log2_T, log2_T)
        for g in c.feat_grids:
            g.log2_T = min(g.log2_T, log2_T)
        return c


def _linear_init(out_dim: int, in_dim: int, gen: torch.Generator) -> torch.Tensor:
    """torch.nn.Linear default (kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(in), 1/sqrt(in)))."""
    bound = 1.0 / math.sqrt(in_dim)
    return (torch.rand((out_dim, in_dim), generator=gen) * 2 - 1) * bound


def init_params(cfg: PathConfig, seed: int = 0, table_scale: float = 1e-3) -> Dict[str, torch.Tensor]:
    """Random-init parameters: tables U(-1,1)*scale (encodings.py:257-258), MLPs Linear-default, no bias."""
    gen = torch.Generator().manual_seed(seed)
    p: Dict[str, torch.Tensor] = {}

    def table(g: GridSpec):
        return (torch.rand((g.rows, g.features), generator=gen) * 2 - 1) * table_scale

    p["prop_table"] = table(cfg.prop_grid)
    p["prop_w0"] = _linear_init(cfg.prop_hidden, cfg.prop_grid.out_dim, gen)
    p["prop_w1"] = _linear_init(1, cfg.prop_hidden, gen)
    p["field_table"] = table(cfg.field_grid)
    p["base_w0"] = _linear_init(cfg.base_hidden, cfg.field_grid.out_dim, gen)
    p["base_w1"] = _linear_init(1 + cfg.geo_feat_dim, cfg.base_hidden, gen)
    p["head_w0"] = _linear_init(cfg.head_hidden, 16 + cfg.geo_feat_dim, gen)
    p["head_w1"] = _linear_init(cfg.head_hidden, cfg.head_hidden, gen)
    p["head_w2"] = _linear_init(3, cfg.head_hidden, gen)
    if cfg.distill_sam:
        feat_in = sum(g.out_dim for g in cfg.feat_grids)
        for head, odim in (("sam", cfg.sam_dim),) + ((("clipseg", cfg.clipseg_dim),) if cfg.use_clipseg else ()):
            for i, g in enumerate(cfg.feat_grids):
                p[f"{head}_table{i}"] = table(g)
            dims = [feat_in] + [cfg.feat_hidden] * cfg.feat_hidden_layers + [odim]
            for i in range(len(dims) - 1):
                p[f"{head}_w{i}"] = _linear_init(dims[i + 1], dims[i], gen)
        if cfg.patch_size > 1:
            k = 3
            bound = 1.0 / math.sqrt(cfg.sam_dim * k * k)
            for i in (0, 1):
                p[f"conv{i}_w"] = (torch.rand((cfg.sam_dim, cfg.sam_dim, k, k), generator=gen) * 2 - 1) * bound
                p[f"conv{i}_b"] = (torch.rand((cfg.sam_dim,), generator=gen) * 2 - 1) * bound
    return p


def _mlp_weights(params, prefix: str) -> List[torch.Tensor]:
    ws, i = [], 0
    while f"{prefix}_w{i}" in params:
        ws.append(params[f"{prefix}_w{i}"])
        i += 1
    return ws


def proposal_density(params, cfg: PathConfig, pos: torch.Tensor) -> torch.Tensor:
    """a5 HashMLPDensityField.get_density.  density_fields.py:102-125.  pos [R,n,3] -> density [R,n]."""
    shape = pos.shape[:-1]
    u, sel = normalize_positions(pos.reshape(-1, 3), float("inf"), True)
    g = cfg.prop_grid
    enc = hashgrid_fwd(u, params["prop_table"], g.scalings(), g.log2_T)
    raw = mlp_fwd(enc, [params["prop_w0"], params["prop_w1"]])
    dens = trunc_exp(raw) * sel[..., None]
    return dens.reshape(*shape)


def field_density_rgb(params, cfg: PathConfig, pos: torch.Tensor, dirs: torch.Tensor):
    """a12+a13 TCNNNerfactoField.get_density/get_outputs (appearance embedding off).  nerfacto_field.py:242-351.

    pos [R,S,3], dirs [R,3] -> density [R,S], rgb [R,S,3].
    """
    R, S = pos.shape[:2]
    u, sel = normalize_positions(pos.reshape(-1, 3), float("inf"), True)
    g = cfg.field_grid
    enc = hashgrid_fwd(u, params["field_table"], g.scalings(), g.log2_T)
    h = mlp_fwd(enc, [params["base_w0"], params["base_w1"]])
    raw, geo = h[:, :1], h[:, 1:]
    dens = trunc_exp(raw) * sel[..., None]
    sh = sh16(dirs)[:, None, :].expand(R, S, 16).reshape(-1, 16)
    rgb = mlp_fwd(torch.cat([sh, geo], dim=-1), _mlp_weights(params, "head"), out_act="sigmoid")
    return dens.reshape(R, S), rgb.reshape(R, S, 3)


def feature_field(params, cfg: PathConfig, pos: torch.Tensor, head: str) -> torch.Tensor:
    """a17 SAMField.get_outputs for one head.  sam_field.py:112-140.  pos [R,K,3] -> [R,K,C]."""
    shape = pos.shape[:-1]
    u, _ = normalize_positions(pos.detach().reshape(-1, 3), None, False)
    xs = [hashgrid_fwd(u, params[f"{head}_table{i}"], g.scalings(), g.log2_T) for i, g in enumerate(cfg.feat_grids)]
    x = torch.cat(xs, dim=-1)
    out = mlp_fwd(x, _mlp_weights(params, head))
    return out.reshape(*shape, -1)


def conv_head(params, feat: torch.Tensor, patch: int) -> torch.Tensor:
    """sam_model.py:259-264: [R,256] -> [R/p^2,256] via conv3x3 -> ReLU -> conv3x3 -> mean(H,W)."""
    x = feat.reshape(-1, patch, patch, feat.shape[-1]).permute(0, 3, 1, 2)
    x = torch.nn.functional.conv2d(x, params["conv0_w"], params["conv0_b"], padding=1)
    x = torch.relu(x)
    x = torch.nn.functional.conv2d(x, params["conv1_w"], params["conv1_b"], padding=1)
    return x.mean(dim=[2, 3])


def forward(params: Dict[str, torch.Tensor], cfg: PathConfig, origins, directions, training: bool,
            t_rand: Optional[torch.Tensor] = None, u_rand: Optional[torch.Tensor] = None,
            anneal: float = 1.0, prop_requires_grad: bool = True,
            get_feature: Sequence[str] = ("sam", "clipseg"), topk_ids: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """SAMModel.forward/get_outputs.  samnerf/sam_model.py:226-314.

    t_rand/u_rand: the two per-ray jitters ([R,1]) the reference draws with torch.rand in training
    (ray_samplers.py:105-106,318-319); pass None for eval.
    """
    R = origins.shape[0]
    nears, fars = collider_near_far(R, training, cfg.near_plane, cfg.far_plane)
    out: Dict[str, torch.Tensor] = {}
    # --- proposal sampler (ray_samplers.py:558-599), one proposal iteration
    sb_p, eb_p = sample_spacing(nears, fars, cfg.num_proposal_samples, t_rand if training else None)
    pos_p = sample_positions(origins, directions, eb_p)
    if prop_requires_grad:
        dens_p = proposal_density(params, cfg, pos_p)
    else:
        with torch.no_grad():
            dens_p = proposal_density(params, cfg, pos_p)
    w_p = weights_from_density(dens_p, eb_p[:, 1:] - eb_p[:, :-1])
    sb_f = pdf_resample(torch.pow(w_p, anneal), sb_p, cfg.num_nerf_samples, u_rand if training else None)
    eb_f = s_to_euclid(sb_f, nears, fars)
    # --- main field
    pos_f = sample_positions(origins, directions, eb_f)
    dens_f, rgb_f = field_density_rgb(params, cfg, pos_f, directions)
    w_f = weights_from_density(dens_f, eb_f[:, 1:] - eb_f[:, :-1])
    out["rgb"] = render_rgb(rgb_f, w_f, training)
    out["accumulation"] = render_accumulation(w_f)
    out["depth"] = render_depth_median(w_f, eb_f)
    out["prop_depth_0"] = render_depth_median(w_p, eb_p)
    out["weights_prop"], out["weights_fine"] = w_p, w_f
    out["sbins_prop"], out["sbins_fine"] = sb_p, sb_f
    out["ebins_prop"], out["ebins_fine"] = eb_p, eb_f
    out["density_fine"], out["rgb_samples"] = dens_f, rgb_f
    # --- feature branch
    if cfg.distill_sam and len(get_feature) > 0:
        w_k, ids = topk_sharpen(w_f, cfg.num_sam_samples, cfg.sharpening_temperature, topk_ids)
        st, en = gather_samples(eb_f, ids)
        pos_k = origins[:, None, :] + directions[:, None, :] * ((st + en) / 2)[..., None]
        out["sam_weights"], out["sam_ids"] = w_k, ids
        if "sam" in get_feature:
            f = feature_mean(feature_field(params, cfg, pos_k, "sam"), w_k.detach())
            out["sam_raw"] = f
            out["sam"] = conv_head(params, f, cfg.patch_size) if cfg.patch_size > 1 else f
        if "clipseg" in get_feature and cfg.use_clipseg:
            out["clipseg"] = feature_mean(feature_field(params, cfg, pos_k, "clipseg"), w_k.detach())
    return out


def loss_dict(out: Dict[str, torch.Tensor], batch: Dict[str, torch.Tensor], cfg: PathConfig) -> Dict[str, torch.Tensor]:
    """SAMModel.get_loss_dict + NerfactoModel.get_loss_dict (training).  sam_model.py:316-328, nerfacto.py:324-333."""
    ld = {"rgb_loss": mse(batch["image"], out["rgb"])}
    ld["interlevel_loss"] = cfg.interlevel_loss_mult * interlevel_loss(
        out["sbins_fine"], out["weights_fine"], out["sbins_prop"], out["weights_prop"])
    ld["distortion_loss"] = cfg.distortion_loss_mult * distortion_loss(out["sbins_fine"], out["weights_fine"])
    if cfg.distill_sam and "sam" in out:
        ld["sam_loss"] = cfg.sam_loss_weight * feature_loss(out["sam"], batch["sam"])
        if cfg.use_clipseg and "clipseg" in out:
            ld["clipseg_loss"] = cfg.clipseg_loss_weight * feature_loss(out["clipseg"], batch["clipseg"])
    return ld


def psnr(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """torchmetrics PeakSignalNoiseRatio(data_range=1.0): -10 log10(mse).  nerfacto.py:232,319."""
    return -10.0 * torch.log10(mse(pred, target))


def synthetic_rays(num_rays: int, seed: int = 0):
    """Synthetic ray bundle of SURVEY.md 8(d): origins U(-.5,.5)^3, unit-normal directions."""
    gen = torch.Generator().manual_seed(seed)
    origins = torch.rand((num_rays, 3), generator=gen) - 0.5
    d = torch.randn((num_rays, 3), generator=gen)
    directions = d / torch.linalg.norm(d, dim=-1, keepdim=True)
    return origins, directions


def synthetic_batch(cfg: PathConfig, num_rays: int, seed: int = 1) -> Dict[str, torch.Tensor]:
    gen = torch.Generator().manual_seed(seed)
    b = {"image": torch.rand((num_rays, 3), generator=gen)}
    if cfg.distill_sam:
        n_sam = num_rays // (cfg.patch_size**2) if cfg.patch_size > 1 else num_rays
        b["sam"] = torch.randn((n_sam, cfg.sam_dim), generator=gen)
        if cfg.use_clipseg:
            b["clipseg"] = torch.randn((num_rays, cfg.clipseg_dim), generator=gen)
    return b


# --------------------------------------------------------------------------------------------
# patch-render eval path   samnerf/sam_model.py:337-419, samnerf/sam_utils.py:7-14
# --------------------------------------------------------------------------------------------
def get_feature_size(h: int, w: int, largesize: int = 64):
    """samnerf/sam_utils.py:7-14 (the reference leaves h == w undefined; square images map to largesize x largesize)."""
    if h < w:
        return int(math.ceil((h / w) * largesize)), largesize
    if h > w:
        return largesize, int(math.ceil((w / h) * largesize))
    return largesize, largesize


def feature_ray_grid(field: torch.Tensor, fh: int, fw: int, p: int) -> torch.Tensor:
    """sam_model.py:371-380 for one RayBundle field [H,W,C]: linspace sub-sampling to [fh*p, fw*p], regroup into p x p
    patches (reshape (fh, p, fw, p) then transpose(1, 2)), row-major flatten -> [fh*fw*p*p, C] in the order the chunk loop
    (get_row_major_sliced_ray_bundle) walks it: patch-major, rows of a patch, columns of a patch."""
    H, W = field.shape[:2]
    hi = torch.linspace(0, H - 1, fh * p, dtype=torch.long)
    wi = torch.linspace(0, W - 1, fw * p, dtype=torch.long)
    hind, wind = torch.meshgrid(hi, wi, indexing="ij")
    return field[hind.flatten(), wind.flatten()].reshape(fh, p, fw, p, -1).transpose(1, 2).reshape(fh * fw * p * p, -1)


def clipseg_ray_grid(field: torch.Tensor, n: int = 32) -> torch.Tensor:
    """sam_model.py:389-398: the n x n linspace sub-sampling of a RayBundle field [H,W,C] -> [n*n, C]."""
    H, W = field.shape[:2]
    hi = torch.linspace(0, H - 1, n, dtype=torch.long)
    wi = torch.linspace(0, W - 1, n, dtype=torch.long)
    hind, wind = torch.meshgrid(hi, wi, indexing="ij")
    return field[hind.flatten(), wind.flatten()]


def render_camera(params, cfg: PathConfig, origins: torch.Tensor, directions: torch.Tensor, chunk: int = 1 << 15):
    """get_outputs_for_camera_ray_bundle passes 1-3 (eval mode, no grad): origins/directions [H,W,3] ->
    rgb/depth/accumulation [H,W,*], sam [fh,fw,256], clipseg [32,32,192]."""
    H, W = origins.shape[:2]
    out: Dict[str, torch.Tensor] = {}
    with torch.no_grad():
        o, d = origins.reshape(-1, 3), directions.reshape(-1, 3)
        parts = [forward(params, cfg, o[i:i + chunk], d[i:i + chunk], False, get_feature=()) for i in range(0, H * W, chunk)]
        for k in ("rgb", "depth", "accumulation", "prop_depth_0"):
            out[k] = torch.cat([p[k] for p in parts]).view(H, W, -1)
        if cfg.distill_sam:
            fh, fw = get_feature_size(H, W)
            p = cfg.patch_size
            chunk = max(p * p, chunk - chunk % (p * p))  # whole patches per chunk (the reference's 1<<15 already is)
            fo, fd = feature_ray_grid(origins, fh, fw, p), feature_ray_grid(directions, fh, fw, p)
            parts = [forward(params, cfg, fo[i:i + chunk], fd[i:i + chunk], False, get_feature=("sam",))["sam"]
                     for i in range(0, fo.shape[0], chunk)]
            out["sam"] = torch.cat(parts).view(fh, fw, -1)
            if cfg.use_clipseg:
                co, cd = clipseg_ray_grid(origins), clipseg_ray_grid(directions)
                parts = [forward(params, cfg, co[i:i + chunk], cd[i:i + chunk], False, get_feature=("clipseg",))["clipseg"]
                         for i in range(0, co.shape[0], chunk)]
                out["clipseg"] = torch.cat(parts).view(32, 32, -1)
    return out


# --------------------------------------------------------------------------------------------
# the batch builder in front of the hot path
#   nerfstudio/data/pixel_samplers.py:50-75,246-300 ; nerfstudio/model_components/ray_generators.py:44-63 ;
#   nerfstudio/cameras/cameras.py:284-311,576-722 (pinhole branch) ; samnerf/data/feature_loader.py:49-56 ;
#   samnerf/datamanager.py:97-117
# --------------------------------------------------------------------------------------------
def pixel_indices(u: torch.Tensor, num_images: int, H: int, W: int) -> torch.Tensor:
    """PixelSampler.sample_method without a mask: u [B,3] ~ U[0,1) -> (camera, row, col) int64."""
    return torch.floor(u * torch.tensor([num_images, H, W])).long()


def patch_pixel_indices(u: torch.Tensor, num_images: int, H: int, W: int, p: int) -> torch.Tensor:
    """PatchPixelSampler.sample_method: u [B/p^2,3] -> top-left corners U * (n, H-p, W-p), + (yy, xx), floor."""
    n = u.shape[0]
    ind = u * torch.tensor([num_images, H - p, W - p])
    ind = ind.view(n, 1, 1, 3).broadcast_to(n, p, p, 3).clone()
    yys, xxs = torch.meshgrid(torch.arange(p), torch.arange(p), indexing="ij")
    ind[:, ..., 1] += yys
    ind[:, ..., 2] += xxs
    return torch.floor(ind).long().flatten(0, 2)


def generate_rays(indices: torch.Tensor, c2w: torch.Tensor, fx, fy, cx, cy):
    """RayGenerator.forward + the PERSPECTIVE branch of Cameras._generate_rays_from_coords (no distortion, no camera
    optimizer).  indices [R,3] (camera, row, col); c2w [N,3,4]; fx, fy, cx, cy [N].
    -> origins [R,3], directions [R,3] (unit), pixel_area [R,1], camera_indices [R,1]."""
    c, yi, xi = indices[:, 0], indices[:, 1], indices[:, 2]
    y, x = yi + 0.5, xi + 0.5  # image_coords: pixel centres (cameras.py:303-304)
    fx_, fy_, cx_, cy_ = fx[c], fy[c], cx[c], cy[c]
    coord = torch.stack([(x - cx_) / fx_, -(y - cy_) / fy_], -1)
    coord_x = torch.stack([(x - cx_ + 1) / fx_, -(y - cy_) / fy_], -1)
    coord_y = torch.stack([(x - cx_) / fx_, -(y - cy_ + 1) / fy_], -1)
    cs = torch.stack([coord, coord_x, coord_y], 0)  # [3,R,2]
    d = torch.empty(cs.shape[:-1] + (3,))
    d[..., 0], d[..., 1], d[..., 2] = cs[..., 0], cs[..., 1], -1.0
    rot = c2w[c][..., :3, :3]
    d = torch.sum(d[..., None, :] * rot, dim=-1)
    eps = torch.tensor([np.finfo(float).eps * 4.0]).to(d)
    d = d / torch.maximum(torch.linalg.vector_norm(d, dim=-1, keepdims=True), eps)  # normalize_with_norm
    dx = torch.sqrt(torch.sum((d[0] - d[1]) ** 2, dim=-1))
    dy = torch.sqrt(torch.sum((d[0] - d[2]) ** 2, dim=-1))
    return c2w[c][..., :3, 3], d[0], (dx * dy)[..., None], c[:, None]


def gather_features(features: torch.Tensor, img_points: torch.Tensor, image_shape) -> torch.Tensor:
    """FeatureDataloader.__call__: nearest lookup features[cam, long(row * fh/H), long(col * fw/W)]."""
    scale = (features.shape[1] / image_shape[0], features.shape[2] / image_shape[1])
    xi, yi = (img_points[:, 1] * scale[0]).long(), (img_points[:, 2] * scale[1]).long()
    return features[img_points[:, 0].long(), xi, yi]


def build_batch(u, images, c2w, fx, fy, cx, cy, patch: int, sam_features=None, clipseg_features=None):
    """SAMDataManager.next_train (samnerf/datamanager.py:97-117) from the random draws `u`:
    images [N,H,W,3]; returns (origins, directions, pixel_area, camera_indices), batch{image, indices, sam, clipseg}."""
    N, H, W = images.shape[:3]
    ind = patch_pixel_indices(u, N, H, W, patch) if patch > 1 else pixel_indices(u, N, H, W)
    batch = {"image": images[ind[:, 0], ind[:, 1], ind[:, 2]], "indices": ind}
    if sam_features is not None:
        centers = ind.reshape(-1, patch, patch, 3)[:, patch // 2, patch // 2, :]
        batch["sam"] = gather_features(sam_features, centers, (H, W))
    if clipseg_features is not None:
        batch["clipseg"] = gather_features(clipseg_features, ind, (H, W))
    return generate_rays(ind, c2w, fx, fy, cx, cy), batch

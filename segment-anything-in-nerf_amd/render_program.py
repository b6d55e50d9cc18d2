"""The eval / render path as a STATIC NO-GRAD LAUNCH SCHEDULE (samnerf/sam_model.py:337-419, passes 1-3).

`SAMModel.get_outputs_for_camera_ray_bundle` renders an image chunk by chunk through the same forward as a train step
(sam_model.py:226-301), three times over: every pixel for RGB / depth / accumulation, the [fh*p, fw*p] feature ray grid for the
SAM map, 32 x 32 rays for the ClipSeg map.  Driving each chunk through the plugin classes costs the host ~1.5 ms and leaves
work in the kernels nobody reads.  Here a chunk is a recorded list of C-ABI launches (like `StepProgram` for the train step):

  * every intermediate lives in a buffer allocated once per (chunk size, pass);
  * the chunk's ray origins / directions are read IN PLACE from the camera bundle and its outputs are written in place into the
    image-sized result tensors -- the two pointer arguments that change from chunk to chunk are patched into the recorded
    argument lists (no slicing, no `torch.cat`);
  * the feature passes evaluate what their one output depends on and nothing else: the proposal sampler, the field's DENSITY
    (hash grid + base MLP; no SH, no colour network, no RGB / depth compositing -- the reference computes and drops them,
    sam_model.py:379-384,399-404), top-K + sharpen, the head;
  * a head's hidden activations are rendered inside the GEMM epilogue (snf_linear_fwd_mean: the weighted mean over the K
    samples commutes with the linear last layer, sam_model.py:126-137) -- the [R*K, 256] activations never reach HBM.

Same arithmetic as the eager eval path kernel for kernel (tests/test_model_gpu.py runs both against the oracle's
`render_camera`); only the order of the heads' last layer and the mean differs (fp32 rounding, 1e-7).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

from . import _lib, ops


FUSED_GRID_HEAD = True  # feature passes: grids + first head layer + mean in one kernel (module constant: tests flip it)


class RenderProgram:
    """Built lazily by `SAMModel.get_outputs_for_camera_ray_bundle`; `render(origins, directions, mode)` renders [n] rays."""

    MODES = ("rgb", "sam", "clipseg")

    @staticmethod
    def unsupported_reason(model) -> Optional[str]:
        c = model.config
        if not torch.cuda.is_available():
            return "no GPU"
        if c.num_proposal_iterations != 1 or c.use_same_proposal_network:
            return "more than one proposal iteration"
        if not model.field._fusable():
            return "nerfacto field is not the fused 32-64-16 / 31-64-64-3 shape"
        if not ops.PLANAR_FIELD_ENCODING:
            return "level-major field encoding switched off"
        prop = model.proposal_networks[0].mlp_base
        if not ops.mlp_tiny_supported(prop.network.n_input_dims, prop.network.weights(), prop.network.output_activation):
            return "proposal network is not the 10-16-1 shape"
        if getattr(c, "distill_sam", False) and c.use_dino_feature:
            return "dino head"
        return None

    def __init__(self, model) -> None:
        self.model = model
        self.cfg = model.config
        self.dev = model.device
        self.lib = _lib.load()
        self.bufs: Dict[str, torch.Tensor] = {}
        self.plans: Dict[tuple, tuple] = {}
        self._keep: list = []

    # ------------------------------------------------------------------------------------------------------------
    def buf(self, name: str, shape, dtype=torch.float32) -> torch.Tensor:
        shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        t = self.bufs.get(name)
        if t is None or tuple(t.shape) != shape or t.dtype != dtype:
            t = self.bufs[name] = torch.empty(shape, device=self.dev, dtype=dtype)
        return t

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.bufs.values())

    # ------------------------------------------------------------------------------------------------------------
    def _build(self, R: int, mode: str, fast: bool):
        """-> (entries [[fn, args]], slots {name: [(args, index)]}, outputs {name: channels}) for one chunk of R rays."""
        model, cfg = self.model, self.cfg
        P, S = cfg.num_proposal_samples_per_ray[0], cfg.num_nerf_samples_per_ray
        N0, N1 = R * P, R * S
        pre = f"{mode}{R}_"
        b = lambda name, shape, dtype=torch.float32: self.buf(pre + name, shape, dtype)  # noqa: E731
        entries: list = []
        slots: Dict[str, list] = {}
        st = torch.cuda.current_stream().cuda_stream

        def k(name: str, *args, dyn: Optional[dict] = None) -> None:
            a = [x.data_ptr() if isinstance(x, torch.Tensor) else x for x in args]
            a.append(st)
            entries.append([getattr(self.lib, name), a, name])
            for slot, idx in (dyn or {}).items():
                slots.setdefault(slot, []).append((a, idx))

        # eval: near plane 0 (scene_colliders.py:170-189), no jitter (ray_samplers.py:105,318)
        nears, fars = b("nears", (R,)), b("fars", (R,))
        nears.zero_()
        fars.fill_(float(model.collider.far_plane))
        # ---- proposal sampler (ray_samplers.py:549-599)
        prop = model.proposal_networks[0]
        penc, pnet = prop.mlp_base.encoding, prop.mlp_base.network
        pw0, pw1 = pnet.weights()
        PL, PF, PT = penc.n_levels, penc.n_features_per_level, penc.log2_hashmap_size
        sb0, eb0 = b("sb0", (R, P + 1)), b("eb0", (R, P + 1))
        k("snf_sample_spacing", nears, fars, None, R, P, sb0, eb0)
        u0, sel0 = b("u0", (N0, 3)), b("sel0", (N0,), torch.uint8)
        k("snf_positions", 0, 0, eb0, None, R, P, P, ops.CONTRACT_LINF, 1, u0, sel0, dyn={"o": 0, "d": 1})
        enc0 = b("enc0", (N0, PL * PF))
        k("snf_hashgrid_fwd", u0, penc.params, penc.scalings, N0, PL, PF, PT, enc0, PL * PF, 0)
        I0, H0 = pnet.n_input_dims, pw0.shape[0]
        raw0 = b("raw0", (N0, 1))
        k("snf_mlp_tiny_fwd", enc0, I0, pw0, pw1, I0, H0, N0, None, raw0)
        dens0 = b("dens0", (N0,))
        k("snf_trunc_exp_fwd", raw0, 1, sel0, N0, dens0)
        w0 = b("w0", (R, P))
        k("snf_weights_fwd", dens0, 1, 1, None, eb0, R, P, w0, None)
        sb1, eb1 = b("sb1", (R, S + 1)), b("eb1", (R, S + 1))
        k("snf_pdf_resample", w0, sb0, None, nears, fars, R, P, S, float(model.proposal_sampler._anneal),
          float(model.proposal_sampler.pdf_sampler.histogram_padding), sb1, eb1, dyn={"anneal": 8})
        # ---- nerfacto field: density always, colour only for the RGB pass (nerfacto_field.py:242-351)
        fenc, fbase, fhead = model.field.mlp_base.encoding, model.field.mlp_base.network, model.field.mlp_head
        bw0, bw1 = fbase.weights()
        hw0, hw1, hw2 = fhead.weights()
        FL, FF, FT = fenc.n_levels, fenc.n_features_per_level, fenc.log2_hashmap_size
        u1, sel1 = b("u1", (N1, 3)), b("sel1", (N1,), torch.uint8)
        k("snf_positions", 0, 0, eb1, None, R, S, S, ops.CONTRACT_LINF, 1, u1, sel1, dyn={"o": 0, "d": 1})
        enc1 = b("enc1", (FL * FF * N1,))
        k("snf_hashgrid_fwd", u1, fenc.params, fenc.scalings, N1, FL, FF, FT, enc1, 0, 0)
        C = bw1.shape[0]
        h = b("h", (N1, C))
        k("snf_mlp64_fwd", enc1, 0, bw0, FL * FF, None, bw1, 1, C, ops.ACT_NONE, N1, None, None, h, C)
        density1 = b("density1", (N1,))
        k("snf_trunc_exp_fwd", h, C, sel1, N1, density1)
        w1 = b("w1", (R, S))
        k("snf_weights_fwd", density1, 1, 1, None, eb1, R, S, w1, None)
        outputs: Dict[str, int] = {}
        if mode == "rgb":
            n_geo = C - 1
            x2 = b("x2", (N1, 32))
            k("snf_head_input", 0, h.data_ptr() + 4, R, S, n_geo, C, x2, 32, dyn={"d": 0})
            rgb = b("rgb", (N1, 3))
            k("snf_mlp64_fwd", x2, 32, hw0, 16 + n_geo, hw1, hw2, 2, 3, ops.ACT_SIGMOID, N1, None, None, rgb, 3)
            k("snf_composite_fwd", rgb, w1, None, R, S, 0, 0, None, None, dyn={"out:rgb": 6})
            outputs["rgb"] = 3
            if fast:
                k("snf_composite_fwd", None, w1, eb1, R, S, 0, None, None, 0, dyn={"out:depth": 8})
                outputs["depth"] = 1
            else:
                k("snf_composite_fwd", None, w1, eb1, R, S, 0, None, 0, 0, dyn={"out:accumulation": 7, "out:depth": 8})
                k("snf_composite_fwd", None, w0, eb0, R, P, 0, None, None, 0, dyn={"out:prop_depth_0": 8})
                outputs.update({"accumulation": 1, "depth": 1, "prop_depth_0": 1})
            return entries, slots, outputs
        # ---- feature pass: top-K + sharpen (sam_model.py:243-255), one head (sam_field.py:112-140), MeanRenderer, conv head
        K = cfg.num_sam_samples
        NK = R * K
        sf = model.sam_field
        encs = list(sf.clip_encs if mode == "sam" else sf.clipseg_encs)
        net = sf.sam_net if mode == "sam" else sf.clipseg_net
        ws_ = net.weights()
        ids, wk = b("ids", (R, K), torch.int32), b("wk", (R, K))
        k("snf_topk_sharpen", w1, R, S, K, float(cfg.sharpening_temperature), ids, wk)
        uk = b("uk", (NK, 3))
        k("snf_positions", 0, 0, eb1, ids, R, S, K, ops.CONTRACT_L2, 0, uk, None, dyn={"o": 0, "d": 1})
        total = sum(e.n_output_dims for e in encs)
        gemm_b3 = int(self.lib.snf_get_gemm_mode()) >= 1
        planar = gemm_b3 and all(e.n_features_per_level == 8 for e in encs) and total % 16 == 0 and 64 <= total <= 256
        ld_enc = -8 if planar else total
        n_lay = len(ws_)
        commute = n_lay >= 2 and net.output_activation == ops.ACT_NONE and ws_[-1].shape[1] % 4 == 0
        wp = ws_[n_lay - 2] if commute else None
        fuse_mean = (commute and gemm_b3 and K == 16 and NK >= 8192 and wp.shape[0] % 32 == 0
                     and 64 <= wp.shape[0] <= 256 and 64 <= wp.shape[1] <= 256 and wp.shape[1] % 16 == 0
                     and int(self.lib.snf_linear_bwd_weight_workspace_bytes(NK, wp.shape[1], wp.shape[0])) > 0)
        # The render pass has no backward, so the encoding itself is not needed: grids -> LDS -> first layer -> ReLU -> weighted mean
        # in one kernel (csrc/fused_head.hip, north_star's "LDS staging of per-sample features") when the hidden layer that is
        # rendered is the FIRST layer of the head (the samnerf heads: 192 -> 256 -> out).
        fused = (FUSED_GRID_HEAD and commute and gemm_b3 and K == 16 and n_lay == 2 and len(encs) == 2 and NK % 64 == 0
                 and wp.shape[0] in (128, 256) and wp.shape[1] == total and total <= 256
                 and all(e.n_features_per_level == 8 for e in encs) and encs[0].log2_hashmap_size == encs[1].log2_hashmap_size
                 and (encs[0].n_levels + encs[1].n_levels) % 2 == 0)
        x, hbar = None, None
        if fused:
            O, I = wp.shape
            whi, wlo = b("w0_hi", (O * I,), torch.int16), b("w0_lo", (O * I,), torch.int16)
            k("snf_split_weights_b3", wp, O, I, whi, wlo)  # (per chunk: the weights may have been trained since the last render)
            hbar = b("hbar", (R, O))
            ea, eb_ = encs
            k("snf_grid_head_fused_fwd", uk, ea.params, ea.scalings, ea.n_levels, eb_.params, eb_.scalings, eb_.n_levels,
              ea.log2_hashmap_size, whi, wlo, O, wk, K, hbar, NK)
        else:
            enc_out = b("enc", (NK * total,) if planar else (NK, total))
            col = 0
            for e in encs:
                L, F, T = e.n_levels, e.n_features_per_level, e.log2_hashmap_size
                if planar:
                    k("snf_hashgrid_fwd", uk, e.params, e.scalings, NK, L, F, T, enc_out.data_ptr() + col * NK * 4, 0, 0)
                else:
                    k("snf_hashgrid_fwd", uk, e.params, e.scalings, NK, L, F, T, enc_out, total, col)
                col += L * F
            x = enc_out
        for i, w in enumerate(() if fused else (ws_[:n_lay - 1] if commute else ws_)):
            O, I = w.shape
            act = ops.ACT_RELU if i < n_lay - 1 else net.output_activation
            ldx = ld_enc if i == 0 else I
            if fuse_mean and i == n_lay - 2:
                hbar = b("hbar", (R, O))
                k("snf_linear_fwd_mean", x, w, NK, I, O, ldx, wk, K, hbar, b("mask", (NK, O // 8), torch.uint8), None, O)
                x = None
                continue
            y = b(f"a{i}", (NK, O))
            k("snf_linear_fwd", x, w, None, NK, I, O, ldx, O, act, y)
            x = y
        if commute:
            w_last = ws_[-1]
            Cf, Ih = w_last.shape
            if hbar is None:
                hbar = b("hbar", (R, Ih))
                k("snf_feature_mean_fwd", x, wk, R, K, Ih, hbar)
            conv = mode == "sam" and cfg.patch_size > 1
            if conv:
                fm = b("fm", (R, Cf))
                k("snf_linear_fwd", hbar, w_last, None, R, Ih, Cf, Ih, Cf, ops.ACT_NONE, fm)
            else:
                k("snf_linear_fwd", hbar, w_last, None, R, Ih, Cf, Ih, Cf, ops.ACT_NONE, 0, dyn={"out:" + mode: 9})
        else:
            Cf = ws_[-1].shape[0]
            conv = mode == "sam" and cfg.patch_size > 1
            if conv:
                fm = b("fm", (R, Cf))
                k("snf_feature_mean_fwd", x, wk, R, K, Cf, fm)
            else:
                k("snf_feature_mean_fwd", x, wk, R, K, Cf, 0, dyn={"out:" + mode: 5})
        if conv:
            c0, c1 = model.conv_head[0], model.conv_head[2]
            p, ks = cfg.patch_size, c0.weight.shape[-1]
            kk = ks * ks
            O0, O1 = c0.weight.shape[0], c1.weight.shape[0]
            npatch = R // (p * p)
            colb = b("cv_col", (R, Cf * kk))
            k("snf_patch_unfold", fm, R, p, Cf, ks, colb)
            hc = b("cv_h", (R, O0))
            nb0 = int(self.lib.snf_linear_fwd_workspace_bytes(R, Cf * kk, O0))
            k("snf_linear_fwd_ws", colb, c0.weight, c0.bias, R, Cf * kk, O0, Cf * kk, O0, ops.ACT_RELU, hc,
              b("cv_ws0", (max(nb0, 16) // 4,)), nb0)
            cm = b("cv_cm", (npatch, O0 * kk))
            k("snf_patch_unfold_mean", hc, R, p, O0, ks, cm)
            nb1 = int(self.lib.snf_linear_fwd_workspace_bytes(npatch, O0 * kk, O1))
            k("snf_linear_fwd_ws", cm, c1.weight, c1.bias, npatch, O0 * kk, O1, O0 * kk, O1, ops.ACT_NONE, 0,
              b("cv_ws1", (max(nb1, 16) // 4,)), nb1, dyn={"out:sam": 9})
            outputs["sam"] = O1
        else:
            outputs[mode] = Cf
        return entries, slots, outputs

    # ------------------------------------------------------------------------------------------------------------
    def rows_out(self, n_rays: int, mode: str) -> int:
        p = self.cfg.patch_size
        return n_rays // (p * p) if (mode == "sam" and p > 1) else n_rays

    @torch.no_grad()
    def render(self, origins: torch.Tensor, directions: torch.Tensor, mode: str = "rgb", fast: bool = False,
               chunk: Optional[int] = None) -> Dict[str, torch.Tensor]:
        """Render the rays `origins`, `directions` ([n, 3], contiguous fp32 on the device) in chunks of `chunk` rays
        (config.eval_num_rays_per_chunk).  mode 'rgb': {'rgb', 'depth'[, 'accumulation', 'prop_depth_0']} [n, C];
        'sam' / 'clipseg': that head's rendered feature rows ([n / p^2, 256] behind the conv head, [n, C] otherwise)."""
        assert mode in self.MODES
        n = origins.shape[0]
        chunk = int(chunk or self.cfg.eval_num_rays_per_chunk)
        p = self.cfg.patch_size
        if mode == "sam" and p > 1:
            assert n % (p * p) == 0 and chunk % (p * p) == 0, "feature rays come in whole p x p patches"
        assert origins.is_cuda and origins.is_contiguous() and directions.is_contiguous() and origins.dtype == torch.float32
        if torch.cuda.current_stream().cuda_stream != getattr(self, "_stream", torch.cuda.current_stream().cuda_stream):
            self.plans.clear()  # (recorded for another stream)
        self._stream = torch.cuda.current_stream().cuda_stream
        results: Dict[str, torch.Tensor] = {}
        anneal = float(self.model.proposal_sampler._anneal)
        if mode != "rgb":  # tables trained table-parallel are made whole before a replicated evaluation (collective)
            sf = self.model.sam_field
            ops._tp_refresh([e.params for e in (sf.clip_encs if mode == "sam" else sf.clipseg_encs)])
        for i in range(0, n, chunk):
            R = min(chunk, n - i)
            key = (R, mode, bool(fast), int(self.lib.snf_get_gemm_mode()))
            plan = self.plans.get(key)
            if plan is None:
                plan = self.plans[key] = self._build(R, mode, bool(fast))
            entries, slots, outputs = plan
            if not results:
                results = {name: torch.empty((self.rows_out(n, mode) if name == mode else n, ch), device=self.dev)
                           for name, ch in outputs.items()}
            r0 = self.rows_out(i, mode)
            vals = {"o": origins.data_ptr() + i * 12, "d": directions.data_ptr() + i * 12, "anneal": anneal}
            for name, t in results.items():
                row = r0 if name == mode else i
                vals["out:" + name] = t.data_ptr() + row * t.shape[1] * 4
            for slot, sites in slots.items():
                v = vals[slot]
                for a, idx in sites:
                    a[idx] = v
            for fn, args, name in entries:
                rc = fn(*args)
                if rc:
                    _lib.check(rc, name)
        return results

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

  * the feature passes' rays are an INDEX SUBSET of the camera's rays (sam_model.py:371-377,392-398:
    `camera_ray_bundle[hind.flatten(), wind.flatten()]` with `linspace(...).long()` indices) and eval sampling is deterministic, so
    pass 1 has already computed their samples and weights: while a chunk of pass 1 is resident, the selected samples (top-K ids,
    sharpened weights, contracted positions) of the feature rays among its pixels are kept (`collect`), and passes 2-3 run the
    heads only (`render_heads`) -- the same numbers as the reference's recomputation, bit for bit, without 20 % of the image's
    proposal / field-grid / density work.

Same arithmetic as the eager eval path kernel for kernel (tests/test_model_gpu.py runs both against the oracle's
`render_camera`); only the order of the heads' last layer and the mean differs (fp32 rounding, 1e-7).
"""
from __future__ import annotations

import os as _os
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib, ops


FUSED_GRID_HEAD = True  # feature passes: grids + first head layer + mean in one kernel (module constant: tests flip it)
# feature passes take the selected samples of their rays from pass 1 instead of sampling them again (SNF_RENDER_REUSE_PASS1=0: as the
# reference does, every pass samples its own rays)
REUSE_PASS1 = _os.environ.get("SNF_RENDER_REUSE_PASS1", "1") == "1"
FUSED_SH_INPUT = _os.environ.get("SNF_FUSED_SH_INPUT", "1") == "1"  # colour net input cat(SH16(d), geo) formed in its loader
FUSED_DENSITY = _os.environ.get("SNF_FUSED_DENSITY", "1") == "1"  # trunc_exp of the base net's output 0 in its epilogue
FUSED_PROP = _os.environ.get("SNF_FUSED_PROP", "1") == "1"  # proposal grid + density net + trunc_exp in one launch (eval)
# pass 1: consecutive chunks on this many alternating streams, each with its own intermediates -- a chunk's grid forwards (bound by the
# texture path) beside the previous chunk's MLP chains (matrix cores + streaming).  Measured SLOWER: 16.6 ms on one stream, 17.7 on two,
# 19.0 on three (the level-at-a-time forward lives on its 4 MB slab staying in L2; the chains' 800 MB per chunk stream through it)
LANES = max(1, int(_os.environ.get("SNF_RENDER_LANES", "1")))


class _Dyn:
    """An argument patched per chunk (a pointer into an image-sized tensor): recorded as 0, registered under `slot`."""
    __slots__ = ("slot",)

    def __init__(self, slot: str) -> None:
        self.slot = slot


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
        self._sig: Optional[tuple] = None
        self._maps: Dict[tuple, tuple] = {}

    def _signature(self) -> tuple:
        """What a recorded argument list bakes in besides its own buffers: the device address (and shape) of every parameter
        and buffer of the model, the far plane.  `build_arenas()` (p.data = arena view), `model.to()` or a changed collider move
        them; plans recorded before are then dropped instead of reading freed memory."""
        m = self.model
        ptrs = tuple((t.data_ptr(), tuple(t.shape)) for t in list(m.parameters()) + list(m.buffers()))
        return ptrs + (float(m.collider.far_plane),)

    # ------------------------------------------------------------------------------------------------------------
    def buf(self, name: str, shape, dtype=torch.float32) -> torch.Tensor:
        shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        t = self.bufs.get(name)
        if t is None:
            t = self.bufs[name] = torch.empty(shape, device=self.dev, dtype=dtype)
        elif tuple(t.shape) != shape or t.dtype != dtype:
            # (a silent reallocation would leave the plans that recorded the old address with a dangling pointer)
            raise RuntimeError(f"render buffer {name!r} exists as {tuple(t.shape)} {t.dtype}, requested {shape} {dtype}")
        return t

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.bufs.values())

    # ------------------------------------------------------------------------------------------------------------
    def _recorder(self, pre: str):
        """(k, b, entries, slots): `k(name, *args)` records one C-ABI launch on the current stream (tensors -> device addresses,
        `_Dyn` arguments -> per-chunk slots), `b(name, shape)` is a buffer private to this plan family."""
        entries: list = []
        slots: Dict[str, list] = {}
        st = torch.cuda.current_stream().cuda_stream
        b = lambda name, shape, dtype=torch.float32: self.buf(pre + name, shape, dtype)  # noqa: E731

        def k(name: str, *args, dyn: Optional[dict] = None) -> None:
            a = []
            for idx, x in enumerate(args):
                if isinstance(x, _Dyn):
                    slots.setdefault(x.slot, []).append((a, idx))
                    a.append(0)
                else:
                    a.append(x.data_ptr() if isinstance(x, torch.Tensor) else x)
            a.append(st)
            entries.append([getattr(self.lib, name), a, name])
            for slot, idx in (dyn or {}).items():
                slots.setdefault(slot, []).append((a, idx))

        return k, b, entries, slots

    def _prefix(self, R: int, mode: str, tag: str = "") -> str:
        # (buffers of different GEMM modes / head paths / lanes never share a name: a plan of the other kind keeps its own addresses)
        lane = getattr(self, "_lane", 0)
        return f"{mode}{tag}{R}_g{int(self.lib.snf_get_gemm_mode())}{'f' if FUSED_GRID_HEAD else 'u'}{'L%d' % lane if lane else ''}_"

    def _build(self, R: int, mode: str, fast: bool):
        """-> (entries [[fn, args, name]], slots {name: [(args, index)]}, outputs {name: channels}, handles) for one chunk of R
        rays sampled from scratch; handles = the chunk's weights / bin edges (what `collect` reads)."""
        model, cfg = self.model, self.cfg
        P, S = cfg.num_proposal_samples_per_ray[0], cfg.num_nerf_samples_per_ray
        N0, N1 = R * P, R * S
        k, b, entries, slots = self._recorder(self._prefix(R, mode))
        o_, d_ = _Dyn("o"), _Dyn("d")
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
        k("snf_positions", o_, d_, eb0, None, R, P, P, ops.CONTRACT_LINF, 1, u0, sel0)
        I0, H0 = pnet.n_input_dims, pw0.shape[0]
        dens0 = b("dens0", (N0,))
        if FUSED_PROP and PL == 5 and PF == 2 and H0 == 16 and I0 == 10 and pw1.shape[0] == 1:
            # grid + density net + trunc_exp per sample in registers: the [N, 10] encoding is never written
            k("snf_prop_density_fwd", u0, penc.params, penc.scalings, N0, PL, PF, PT, pw0, pw1, H0, sel0, dens0)
        else:
            enc0 = b("enc0", (N0, PL * PF))
            k("snf_hashgrid_fwd", u0, penc.params, penc.scalings, N0, PL, PF, PT, enc0, PL * PF, 0)
            raw0 = b("raw0", (N0, 1))
            k("snf_mlp_tiny_fwd", enc0, I0, pw0, pw1, I0, H0, N0, None, raw0)
            k("snf_trunc_exp_fwd", raw0, 1, sel0, N0, dens0)
        w0 = b("w0", (R, P))
        k("snf_weights_fwd", dens0, 1, 1, None, eb0, R, P, w0, None)
        sb1, eb1 = b("sb1", (R, S + 1)), b("eb1", (R, S + 1))
        k("snf_pdf_resample", w0, sb0, None, nears, fars, R, P, S, _Dyn("anneal"),
          float(model.proposal_sampler.pdf_sampler.histogram_padding), sb1, eb1)
        # ---- nerfacto field: density always, colour only for the RGB pass (nerfacto_field.py:242-351)
        fenc, fbase, fhead = model.field.mlp_base.encoding, model.field.mlp_base.network, model.field.mlp_head
        bw0, bw1 = fbase.weights()
        hw0, hw1, hw2 = fhead.weights()
        FL, FF, FT = fenc.n_levels, fenc.n_features_per_level, fenc.log2_hashmap_size
        u1, sel1 = b("u1", (N1, 3)), b("sel1", (N1,), torch.uint8)
        k("snf_positions", o_, d_, eb1, None, R, S, S, ops.CONTRACT_LINF, 1, u1, sel1)
        enc1 = b("enc1", (FL * FF * N1,))
        k("snf_hashgrid_fwd", u1, fenc.params, fenc.scalings, N1, FL, FF, FT, enc1, 0, 0)
        C = bw1.shape[0]
        h = b("h", (N1, C))
        density1 = b("density1", (N1,))
        if FUSED_DENSITY:  # (trunc_exp of output 0 from the base net's epilogue)
            k("snf_mlp64_fwd_density", enc1, 0, bw0, FL * FF, None, bw1, 1, C, ops.ACT_NONE, N1, None, None, h, C, sel1, density1)
        else:
            k("snf_mlp64_fwd", enc1, 0, bw0, FL * FF, None, bw1, 1, C, ops.ACT_NONE, N1, None, None, h, C)
            k("snf_trunc_exp_fwd", h, C, sel1, N1, density1)
        w1 = b("w1", (R, S))
        k("snf_weights_fwd", density1, 1, 1, None, eb1, R, S, w1, None)
        handles = {"w1": w1, "eb1": eb1}
        outputs: Dict[str, int] = {}
        if mode == "rgb":
            n_geo = C - 1
            rgb = b("rgb", (N1, 3))
            if FUSED_SH_INPUT and int(self.lib.snf_get_gemm_mode()) == 1 and n_geo == 15 and C % 4 == 0:
                # the colour net forms cat(SH16(d), geo) in its loader: the [N, 32] input (537 MB per 32 768-ray chunk) is never written
                k("snf_mlp64_fwd_sh", d_, R, S, h, C, n_geo, hw0, hw1, hw2, 2, 3, ops.ACT_SIGMOID, None, None, rgb, 3)
            else:
                x2 = b("x2", (N1, 32))
                k("snf_head_input", d_, h.data_ptr() + 4, R, S, n_geo, C, x2, 32)
                k("snf_mlp64_fwd", x2, 32, hw0, 16 + n_geo, hw1, hw2, 2, 3, ops.ACT_SIGMOID, N1, None, None, rgb, 3)
            # (colour, accumulation and median depth of a ray in ONE pass over its weights)
            outputs["rgb"] = 3
            if fast:
                k("snf_composite_fwd", rgb, w1, eb1, R, S, 0, _Dyn("out:rgb"), None, _Dyn("out:depth"))
                outputs["depth"] = 1
            else:
                k("snf_composite_fwd", rgb, w1, eb1, R, S, 0, _Dyn("out:rgb"), _Dyn("out:accumulation"), _Dyn("out:depth"))
                k("snf_composite_fwd", None, w0, eb0, R, P, 0, None, None, _Dyn("out:prop_depth_0"))
                outputs.update({"accumulation": 1, "depth": 1, "prop_depth_0": 1})
            return entries, slots, outputs, handles
        # ---- feature pass: top-K + sharpen (sam_model.py:243-255), then the head
        K = cfg.num_sam_samples
        ids, wk = b("ids", (R, K), torch.int32), b("wk", (R, K))
        k("snf_topk_sharpen", w1, R, S, K, float(cfg.sharpening_temperature), ids, wk)
        uk = b("uk", (R * K, 3))
        k("snf_positions", o_, d_, eb1, ids, R, S, K, ops.CONTRACT_L2, 0, uk, None)
        self._head(k, b, R, mode, uk, wk, outputs)
        return entries, slots, outputs, handles

    def _build_heads(self, R: int, mode: str):
        """One chunk of R feature rays whose selected samples pass 1 has kept: the head only (uk / wk are per-chunk slots)."""
        k, b, entries, slots = self._recorder(self._prefix(R, mode, "H"))
        outputs: Dict[str, int] = {}
        self._head(k, b, R, mode, _Dyn("uk"), _Dyn("wk"), outputs)
        return entries, slots, outputs, {}

    def _head(self, k, b, R: int, mode: str, uk, wk, outputs: Dict[str, int]) -> None:
        """One head (sam_field.py:112-140) on the R*K selected samples at normalised positions uk [R*K,3] with render weights
        wk [R,K], MeanRenderer (sam_model.py:126-137), conv head for 'sam' (sam_model.py:196-200,259-264)."""
        model, cfg = self.model, self.cfg
        K = cfg.num_sam_samples
        NK = R * K
        sf = model.sam_field
        encs = list(sf.clip_encs if mode == "sam" else sf.clipseg_encs)
        net = sf.sam_net if mode == "sam" else sf.clipseg_net
        ws_ = net.weights()
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
        out_slot = _Dyn("out:" + mode)
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
                k("snf_linear_fwd", hbar, w_last, None, R, Ih, Cf, Ih, Cf, ops.ACT_NONE, out_slot)
        else:
            Cf = ws_[-1].shape[0]
            conv = mode == "sam" and cfg.patch_size > 1
            if conv:
                fm = b("fm", (R, Cf))
                k("snf_feature_mean_fwd", x, wk, R, K, Cf, fm)
            else:
                k("snf_feature_mean_fwd", x, wk, R, K, Cf, out_slot)
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
            k("snf_linear_fwd_ws", cm, c1.weight, c1.bias, npatch, O0 * kk, O1, O0 * kk, O1, ops.ACT_NONE, out_slot,
              b("cv_ws1", (max(nb1, 16) // 4,)), nb1)
            outputs["sam"] = O1
        else:
            outputs[mode] = Cf

    # ------------------------------------------------------------------------------------------------------------
    def rows_out(self, n_rays: int, mode: str) -> int:
        p = self.cfg.patch_size
        return n_rays // (p * p) if (mode == "sam" and p > 1) else n_rays

    def _enter(self, mode: str) -> None:
        """Checks common to every render call: eval semantics, the stream and the addresses the plans were recorded with."""
        assert mode in self.MODES
        if self.model.training:
            # (near plane 0, no jitter: the train-mode chunk loop goes through `forward` and the collider)
            raise RuntimeError("RenderProgram renders with eval semantics: call model.eval() first")
        st = torch.cuda.current_stream().cuda_stream
        sig = self._signature()
        if st != getattr(self, "_stream", st) or sig != self._sig:
            self.plans.clear()  # recorded for another stream, or before the parameters / the collider moved
        self._stream, self._sig = st, sig

    def _plan(self, kind: str, R: int, mode: str, fast: bool, lane: int = 0):
        """The recorded launches of one chunk.  lane > 0: recorded on that lane's stream (`_lane_stream`) with its own buffers."""
        key = (kind, R, mode, bool(fast), int(self.lib.snf_get_gemm_mode()), bool(FUSED_GRID_HEAD), bool(FUSED_SH_INPUT), lane)
        plan = self.plans.get(key)
        if plan is None:
            self._lane = lane
            try:
                if lane:
                    with torch.cuda.stream(self._lane_stream(lane)):
                        plan = self._build(R, mode, bool(fast)) if kind == "full" else self._build_heads(R, mode)
                else:
                    plan = self._build(R, mode, bool(fast)) if kind == "full" else self._build_heads(R, mode)
            finally:
                self._lane = 0
            self.plans[key] = plan
        return plan

    def _lane_stream(self, lane: int) -> "torch.cuda.Stream":
        return ops.make_stream(f"render{lane}")

    @staticmethod
    def _replay(plan, vals: dict) -> None:
        entries, slots = plan[0], plan[1]
        for slot, sites in slots.items():
            v = vals[slot]
            for a, idx in sites:
                a[idx] = v
        for fn, args, name in entries:
            rc = fn(*args)
            if rc:
                _lib.check(rc, name)

    def feature_maps(self, key, pixel_ids, n_pixels: int, chunk: int):
        """Per chunk of pass 1 the feature rays among its pixels: (src, dst, bounds, m) with src = chunk-local ray index, dst = the
        feature ray's own row (its position in the index list), both int32 on the device and ordered by pixel; bounds[c] ..
        bounds[c+1] is chunk c's run.  `key` (hashable) names the index list, `pixel_ids` is the list or a callable that makes it;
        cached per (key, n_pixels, chunk)."""
        ck = (key, int(n_pixels), int(chunk))
        m = self._maps.get(ck)
        if m is None:
            pix = (pixel_ids() if callable(pixel_ids) else pixel_ids).reshape(-1).to(device=self.dev, dtype=torch.int64)
            assert pix.numel() > 0 and int(pix.min()) >= 0 and int(pix.max()) < n_pixels
            order = torch.argsort(pix, stable=True)
            sp = pix[order]
            edges = torch.arange(0, n_pixels + chunk, chunk, device=self.dev, dtype=torch.int64)
            bounds = torch.searchsorted(sp, edges).tolist()
            src = (sp % chunk).to(torch.int32).contiguous()
            dst = order.to(torch.int32).contiguous()
            m = self._maps[ck] = (src, dst, bounds, int(pix.numel()))
        return m

    @torch.no_grad()
    def render(self, origins: torch.Tensor, directions: torch.Tensor, mode: str = "rgb", fast: bool = False,
               chunk: Optional[int] = None, collect: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
        """Render the rays `origins`, `directions` ([n, 3], contiguous fp32 on the device) in chunks of `chunk` rays
        (config.eval_num_rays_per_chunk).  mode 'rgb': {'rgb', 'depth'[, 'accumulation', 'prop_depth_0']} [n, C];
        'sam' / 'clipseg': that head's rendered feature rows ([n / p^2, 256] behind the conv head, [n, C] otherwise).
        collect = {name: (key, ray indices [m] into these n rays, or a callable making them)}: the selected samples of those rays
        are kept for `render_heads(name)` (`key` names the index list, see `feature_maps`)."""
        self._enter(mode)
        n = origins.shape[0]
        chunk = int(chunk or self.cfg.eval_num_rays_per_chunk)
        p = self.cfg.patch_size
        if mode == "sam" and p > 1:
            assert n % (p * p) == 0 and chunk % (p * p) == 0, "feature rays come in whole p x p patches"
        assert origins.is_cuda and origins.is_contiguous() and directions.is_contiguous() and origins.dtype == torch.float32
        results: Dict[str, torch.Tensor] = {}
        anneal = float(self.model.proposal_sampler._anneal)
        if mode != "rgb":  # tables trained table-parallel are made whole before a replicated evaluation (collective)
            sf = self.model.sam_field
            ops._tp_refresh([e.params for e in (sf.clip_encs if mode == "sam" else sf.clipseg_encs)])
        K, S = self.cfg.num_sam_samples, self.cfg.num_nerf_samples_per_ray
        kept = {}
        for name, (mkey, pix) in (collect or {}).items():
            maps = self.feature_maps(mkey, pix, n, chunk)
            m = maps[3]
            kept[name] = {"maps": maps, "m": m,
                          "ids": torch.empty((m, K), device=self.dev, dtype=torch.int32),
                          "wk": torch.empty((m, K), device=self.dev), "uk": torch.empty((m * K, 3), device=self.dev)}
        self._kept = kept
        cur = torch.cuda.current_stream()
        n_chunks = (n + chunk - 1) // chunk
        lanes = LANES if n_chunks > 1 else 1
        if lanes > 1:  # lane l = chunks l, l + lanes, ...: lane 0 is the caller's stream, the others start behind what it has queued
            ev = torch.cuda.Event()
            ev.record(cur)
            for l in range(1, lanes):
                self._lane_stream(l).wait_event(ev)
        for ci, i in enumerate(range(0, n, chunk)):
            R = min(chunk, n - i)
            lane = ci % lanes
            st = (self._lane_stream(lane) if lane else cur).cuda_stream
            plan = self._plan("full", R, mode, fast, lane)
            outputs, handles = plan[2], plan[3]
            if not results:
                results = {name: torch.empty((self.rows_out(n, mode) if name == mode else n, ch), device=self.dev)
                           for name, ch in outputs.items()}
            r0 = self.rows_out(i, mode)
            o_ptr, d_ptr = origins.data_ptr() + i * 12, directions.data_ptr() + i * 12
            vals = {"o": o_ptr, "d": d_ptr, "anneal": anneal}
            for name, t in results.items():
                row = r0 if name == mode else i
                vals["out:" + name] = t.data_ptr() + row * t.shape[1] * 4
            self._replay(plan, vals)
            for name, kp in kept.items():
                # the feature rays among this chunk's pixels: top-K + sharpen on their weights, positions of the selected samples
                # (sam_model.py:243-255 on rows of THIS chunk), written to the feature rays' own rows
                src, dst, bounds, _ = kp["maps"]
                a, e = bounds[ci], bounds[ci + 1]
                if e > a:
                    w1, eb1 = handles["w1"], handles["eb1"]
                    rc = self.lib.snf_topk_sharpen_rows(w1.data_ptr(), src.data_ptr() + a * 4, dst.data_ptr() + a * 4, e - a, S, K,
                                                        float(self.cfg.sharpening_temperature), kp["ids"].data_ptr(),
                                                        kp["wk"].data_ptr(), st)
                    _lib.check(rc, "snf_topk_sharpen_rows")
                    rc = self.lib.snf_positions_rows(o_ptr, d_ptr, eb1.data_ptr(), kp["ids"].data_ptr(), src.data_ptr() + a * 4,
                                                     dst.data_ptr() + a * 4, e - a, S, K, ops.CONTRACT_L2, kp["uk"].data_ptr(), st)
                    _lib.check(rc, "snf_positions_rows")
        for l in range(1, lanes):  # the caller's stream continues behind every lane
            ev = torch.cuda.Event()
            ev.record(self._lane_stream(l))
            cur.wait_event(ev)
        return results

    @torch.no_grad()
    def render_heads(self, mode: str, chunk: Optional[int] = None) -> Dict[str, torch.Tensor]:
        """Passes 2 / 3 on the samples `render(..., collect={mode: ...})` kept: the head, the MeanRenderer, the conv head."""
        self._enter(mode)
        kp = self._kept[mode]
        n = kp["m"]
        chunk = int(chunk or self.cfg.eval_num_rays_per_chunk)
        p = self.cfg.patch_size
        K = self.cfg.num_sam_samples
        if mode == "sam" and p > 1:
            assert n % (p * p) == 0 and chunk % (p * p) == 0, "feature rays come in whole p x p patches"
        sf = self.model.sam_field
        ops._tp_refresh([e.params for e in (sf.clip_encs if mode == "sam" else sf.clipseg_encs)])
        results: Dict[str, torch.Tensor] = {}
        for i in range(0, n, chunk):
            R = min(chunk, n - i)
            plan = self._plan("heads", R, mode, False)
            if not results:
                results = {name: torch.empty((self.rows_out(n, mode), ch), device=self.dev) for name, ch in plan[2].items()}
            t = results[mode]
            self._replay(plan, {"uk": kp["uk"].data_ptr() + i * K * 12, "wk": kp["wk"].data_ptr() + i * K * 4,
                                "out:" + mode: t.data_ptr() + self.rows_out(i, mode) * t.shape[1] * 4})
        return results

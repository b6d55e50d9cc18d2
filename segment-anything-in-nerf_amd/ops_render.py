"""Colour-net input, trunc_exp / get_weights, RGB composite, MeanRenderer (split out of ops.py; `samnerf_amd.ops` re-exports everything here)."""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

import torch

from ._opcore import (ACT_BY_NAME, ACT_GELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, CONTRACT_L2, CONTRACT_LINF, CONTRACT_NONE, _L, _chk,
                      _launch, _linear_fwd_ws, _p, _stream)

# ---------------------------------------------------------------------------------------------
# colour-MLP input: SH16(dir) ++ geo features (columns 1.. of the base-MLP output)
# ---------------------------------------------------------------------------------------------
class _HeadInput(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dirs, h, R: int, S: int):
        dirs, h = _chk(dirs, "dirs"), _chk(h, "h")
        n_geo = h.shape[1] - 1
        out = torch.empty((R * S, 16 + n_geo), device=h.device, dtype=torch.float32)
        geo = ctypes.c_void_p(h.data_ptr() + 4)
        _launch("snf_head_input", _p(dirs), geo, R, S, n_geo, h.shape[1], _p(out), 16 + n_geo, _stream())
        ctx.n_geo = n_geo
        return out

    @staticmethod
    def backward(ctx, g):
        # d h[:, 1:] = g[:, 16:], d h[:, 0] = 0 (the density column gets its gradient from weights_from_raw)
        gh = torch.zeros((g.shape[0], 1 + ctx.n_geo), device=g.device, dtype=g.dtype)
        gh[:, 1:] = g[:, 16:]
        return None, gh, None, None


def head_input(dirs, h, R: int, S: int) -> torch.Tensor:
    return _HeadInput.apply(dirs, h, R, S)


# ---------------------------------------------------------------------------------------------
# trunc_exp * selector + get_weights
# ---------------------------------------------------------------------------------------------
class _Weights(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, selector, ebins, R: int, n: int, is_density: bool):
        """h [R*n, C]: column 0 is the pre-activation density (or the density itself when is_density)."""
        h, ebins = _chk(h, "h"), _chk(ebins, "ebins")
        C = h.shape[1]
        w = torch.empty((R, n), device=h.device, dtype=torch.float32)
        _launch("snf_weights_fwd", _p(h), C, int(is_density), _p(selector), _p(ebins), R, n, _p(w), _p(None),
                                        _stream())
        ctx.save_for_backward(h, ebins)
        ctx.selector = selector
        ctx.dims = (R, n, C, int(is_density))
        return w

    @staticmethod
    def backward(ctx, gw):
        h, ebins = ctx.saved_tensors
        R, n, C, is_density = ctx.dims
        gw = _chk(gw, "grad_w")
        gh = torch.zeros_like(h) if C > 1 else torch.empty_like(h)
        _launch("snf_weights_bwd", _p(h), C, is_density, _p(ctx.selector), _p(ebins), _p(gw), R, n, _p(gh),
                                        _stream())
        return gh, None, None, None, None, None


def weights_from_raw(h, selector, ebins, R: int, n: int) -> torch.Tensor:
    """fused trunc_exp * selector + get_weights from the pre-activation density column h[:, 0]."""
    return _Weights.apply(h, selector, ebins, R, n, False)


def weights_from_density(density, ebins) -> torch.Tensor:
    """RaySamples.get_weights(densities): density [R,n] (or [R,n,1]) -> weights [R,n]."""
    R, n = ebins.shape[0], ebins.shape[1] - 1
    return _Weights.apply(density.reshape(R * n, 1), None, ebins, R, n, True)


@torch.no_grad()
def density_from_raw(h, selector, ebins, R: int, n: int) -> torch.Tensor:
    """trunc_exp(h[:,0]) * selector as [R,n] (inspection / API parity; the train path uses weights_from_raw)."""
    h, ebins = _chk(h, "h"), _chk(ebins, "ebins")
    w = torch.empty((R, n), device=h.device, dtype=torch.float32)
    d = torch.empty((R, n), device=h.device, dtype=torch.float32)
    _launch("snf_weights_fwd", _p(h), h.shape[1], 0, _p(selector), _p(ebins), R, n, _p(w), _p(d), _stream())
    return d


class _TruncExpSel(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, selector):
        h = _chk(h, "h")
        N, C = h.shape
        d = torch.empty((N,), device=h.device, dtype=torch.float32)
        _launch("snf_trunc_exp_fwd", _p(h), C, _p(selector), N, _p(d), _stream())
        ctx.save_for_backward(h)
        ctx.selector = selector
        return d

    @staticmethod
    def backward(ctx, gd):
        (h,) = ctx.saved_tensors
        N, C = h.shape
        gd = _chk(gd, "grad_density")
        gh = torch.zeros_like(h) if C > 1 else torch.empty_like(h)
        _launch("snf_trunc_exp_bwd", _p(h), C, _p(ctx.selector), _p(gd), N, _p(gh), C, _stream())
        return gh, None


def trunc_exp_sel(h, selector=None) -> torch.Tensor:
    """density [N] = trunc_exp(h[:, 0]) * selector (h is the [N, C] base-MLP output; column 0 = raw density)."""
    return _TruncExpSel.apply(h, selector)


# ---------------------------------------------------------------------------------------------
# RGB composite ('last_sample' background)
# ---------------------------------------------------------------------------------------------
class _CompositeRGB(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb, weights, training: bool):
        rgb, weights = _chk(rgb, "rgb"), _chk(weights, "weights")
        R, S = weights.shape
        out = torch.empty((R, 3), device=rgb.device, dtype=torch.float32)
        _launch("snf_composite_fwd", _p(rgb), _p(weights), _p(None), R, S, int(training), _p(out), _p(None),
                                          _p(None), _stream())
        ctx.save_for_backward(rgb, weights)
        ctx.training = training
        return out

    @staticmethod
    def backward(ctx, g):
        if not ctx.training:
            raise RuntimeError("composite_rgb backward is defined for training mode only (eval renders under no_grad)")
        rgb, weights = ctx.saved_tensors
        R, S = weights.shape
        g = _chk(g, "grad_rgb")
        grgb = torch.empty_like(rgb)
        gw = torch.empty_like(weights)
        _launch("snf_composite_bwd", _p(rgb), _p(weights), _p(g), R, S, _p(grgb), _p(gw), _stream())
        return grgb, gw, None


def composite_rgb(rgb, weights, training: bool) -> torch.Tensor:
    """rgb [R,S,3] (or [R*S,3]), weights [R,S] -> [R,3]."""
    return _CompositeRGB.apply(rgb, weights, training)


# ---------------------------------------------------------------------------------------------
# MeanRenderer
# ---------------------------------------------------------------------------------------------
class _FeatureMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, embeds, w, R: int, K: int):
        embeds, w = _chk(embeds, "embeds"), _chk(w, "w")
        C = embeds.shape[-1]
        out = torch.empty((R, C), device=embeds.device, dtype=torch.float32)
        _launch("snf_feature_mean_fwd", _p(embeds), _p(w), R, K, C, _p(out), _stream())
        ctx.save_for_backward(w)
        ctx.dims = (R, K, C)
        return out

    @staticmethod
    def backward(ctx, g):
        (w,) = ctx.saved_tensors
        R, K, C = ctx.dims
        g = _chk(g, "grad_out")
        ge = torch.empty((R * K, C), device=g.device, dtype=torch.float32)
        _launch("snf_feature_mean_bwd", _p(g), _p(w), R, K, C, _p(ge), _stream())
        return ge, None, None, None


def feature_mean(embeds, w, R: int, K: int) -> torch.Tensor:
    """embeds [R*K, C]; w [R,K] (treated as a constant, as the reference detaches it)."""
    return _FeatureMean.apply(embeds, w.detach(), R, K)

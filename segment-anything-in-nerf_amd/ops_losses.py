"""L2 / interlevel / distortion losses: value + gradient in one kernel pass each (split out of ops.py; `samnerf_amd.ops` re-exports everything here)."""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

import torch

from ._opcore import (ACT_BY_NAME, ACT_GELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, CONTRACT_L2, CONTRACT_LINF, CONTRACT_NONE, _L, _chk,
                      _launch, _linear_fwd_ws, _p, _stream)

# ---------------------------------------------------------------------------------------------
# regularisers: value + gradient in one kernel pass each
# ---------------------------------------------------------------------------------------------
class _RowMSELoss(torch.autograd.Function):
    """weight * mean_r mean_c (pred - target)^2, optionally skipping NaN rows (nanmean); one launch per direction."""

    @staticmethod
    def forward(ctx, pred, target, weight: float, nan_skip: bool):
        pred, target = _chk(pred, "pred"), _chk(target, "target")
        assert pred.shape == target.shape
        C = pred.shape[-1]
        R = pred.numel() // C
        acc = torch.zeros((516,), device=pred.device, dtype=torch.float32)  # SNF_ROWMSE_SCRATCH_WORDS
        out = torch.empty((2,), device=pred.device, dtype=torch.float32)
        _launch("snf_rowmse_loss_fwd", _p(pred), _p(target), R, C, float(weight), int(nan_skip), _p(acc), _p(out), _stream())
        ctx.args = (R, C, float(weight), int(nan_skip))
        ctx.save_for_backward(pred, target, out)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        pred, target, out = ctx.saved_tensors
        R, C, weight, nan_skip = ctx.args
        g = g.contiguous()
        dpred = torch.empty_like(pred)
        _launch("snf_rowmse_loss_bwd", _p(pred), _p(target), R, C, weight, nan_skip, _p(g), _p(out), _p(dpred), _stream())
        return dpred, None, None, None


def mse_loss(pred, target, weight: float = 1.0) -> torch.Tensor:
    """weight * nn.MSELoss()(pred, target) (mean over all elements; NaN propagates).  A plain mean does not care about the
    row shape: [R, 3] colours are viewed as 64-wide rows so that one wave covers 64 elements instead of 3."""
    if pred.is_contiguous() and target.is_contiguous() and pred.shape[-1] < 64 and pred.numel() % 64 == 0:
        out = _RowMSELoss.apply(pred.view(-1, 64), target.detach().view(-1, 64), weight, False)
        return out
    return _RowMSELoss.apply(pred, target.detach(), weight, False)


def rowmse_nanmean_loss(pred, target, weight: float = 1.0) -> torch.Tensor:
    """weight * mse_loss(pred, target, reduction='none').mean(-1).nanmean() (samnerf/sam_model.py:316-328)."""
    return _RowMSELoss.apply(pred, target.detach(), weight, True)


class _Interlevel(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w_prop, sbins_prop, sbins_fine, w_fine):
        w_prop, sbins_prop = _chk(w_prop, "w_prop"), _chk(sbins_prop, "sbins_prop")
        sbins_fine, w_fine = _chk(sbins_fine, "sbins_fine"), _chk(w_fine, "w_fine")
        R, Pn = w_prop.shape
        S = w_fine.shape[1]
        rows = torch.empty((R,), device=w_prop.device, dtype=torch.float32)
        need = ctx.needs_input_grad[0]
        gwp = torch.empty_like(w_prop) if need else None
        _launch("snf_interlevel", _p(sbins_fine), _p(w_fine), _p(sbins_prop), _p(w_prop), R, S, Pn,
                                       1.0 / float(R * S), _p(rows), _p(gwp), _stream())
        ctx.gwp = gwp
        return rows.sum() / float(R * S)

    @staticmethod
    def backward(ctx, g):
        return (ctx.gwp * g if ctx.gwp is not None else None), None, None, None


def interlevel_loss(w_prop, sbins_prop, sbins_fine, w_fine) -> torch.Tensor:
    """mean(clip(w - w_outer, 0)^2 / (w + 1e-7)); gradient flows to w_prop only (fine side is detached)."""
    return _Interlevel.apply(w_prop, sbins_prop, sbins_fine.detach(), w_fine.detach())


class _Distortion(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, sbins):
        w, sbins = _chk(w, "w"), _chk(sbins, "sbins")
        R, S = w.shape
        rows = torch.empty((R,), device=w.device, dtype=torch.float32)
        need = ctx.needs_input_grad[0]
        gw = torch.empty_like(w) if need else None
        _launch("snf_distortion", _p(sbins), _p(w), R, S, 1.0 / float(R), _p(rows), _p(gw), _stream())
        ctx.gw = gw
        return rows.sum() / float(R)

    @staticmethod
    def backward(ctx, g):
        return (ctx.gw * g if ctx.gw is not None else None), None


def distortion_loss(w, sbins) -> torch.Tensor:
    return _Distortion.apply(w, sbins)

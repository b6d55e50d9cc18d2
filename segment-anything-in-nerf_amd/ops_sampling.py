"""No-grad sampling ops: bin spacing, positions, PDF resampling, top-K, depth / accumulation renders (split out of ops.py; `samnerf_amd.ops` re-exports everything here)."""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

import torch

from ._opcore import (ACT_BY_NAME, ACT_GELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, CONTRACT_L2, CONTRACT_LINF, CONTRACT_NONE, _L, _chk,
                      _launch, _linear_fwd_ws, _p, _stream)

# ---------------------------------------------------------------------------------------------
# no-grad sampling ops
# ---------------------------------------------------------------------------------------------
@torch.no_grad()
def sample_spacing(nears, fars, num_samples: int, t_rand=None):
    nears, fars = _chk(nears.reshape(-1), "nears"), _chk(fars.reshape(-1), "fars")
    R = nears.numel()
    t = None if t_rand is None else _chk(t_rand.reshape(-1), "t_rand")
    sb = torch.empty((R, num_samples + 1), device=nears.device, dtype=torch.float32)
    eb = torch.empty_like(sb)
    _launch("snf_sample_spacing", _p(nears), _p(fars), _p(t), R, num_samples, _p(sb), _p(eb), _stream())
    return sb, eb


@torch.no_grad()
def positions(origins, directions, ebins, ids=None, contraction: int = CONTRACT_LINF, use_selector: bool = True):
    """-> (u [R*K,3] normalised positions, selector [R*K] uint8 or None)."""
    origins, directions, ebins = _chk(origins, "origins"), _chk(directions, "directions"), _chk(ebins, "ebins")
    R, n = ebins.shape[0], ebins.shape[1] - 1
    if ids is not None:
        ids = _chk(ids, "ids", torch.int32)
        K = ids.shape[1]
    else:
        K = n
    u = torch.empty((R * K, 3), device=ebins.device, dtype=torch.float32)
    sel = torch.empty((R * K,), device=ebins.device, dtype=torch.uint8) if use_selector else None
    _launch("snf_positions", _p(origins), _p(directions), _p(ebins), _p(ids), R, n, K, contraction,
                                  int(use_selector), _p(u), _p(sel), _stream())
    return u, sel


@torch.no_grad()
def pdf_resample(weights, sbins_in, nears, fars, num_samples: int, u_rand=None, anneal: float = 1.0,
                 histogram_padding: float = 0.01):
    weights, sbins_in = _chk(weights, "weights"), _chk(sbins_in, "sbins_in")
    nears, fars = _chk(nears.reshape(-1), "nears"), _chk(fars.reshape(-1), "fars")
    R, Pn = weights.shape
    u = None if u_rand is None else _chk(u_rand.reshape(-1), "u_rand")
    sb = torch.empty((R, num_samples + 1), device=weights.device, dtype=torch.float32)
    eb = torch.empty_like(sb)
    _launch("snf_pdf_resample", _p(weights), _p(sbins_in), _p(u), _p(nears), _p(fars), R, Pn, num_samples,
                                     float(anneal), float(histogram_padding), _p(sb), _p(eb), _stream())
    return sb, eb


@torch.no_grad()
def topk_sharpen(weights, k: int, temperature: float = 10.0):
    weights = _chk(weights, "weights")
    R, S = weights.shape
    ids = torch.empty((R, k), device=weights.device, dtype=torch.int32)
    w = torch.empty((R, k), device=weights.device, dtype=torch.float32)
    _launch("snf_topk_sharpen", _p(weights), R, S, k, float(temperature), _p(ids), _p(w), _stream())
    return w, ids


@torch.no_grad()
def render_depth_acc(weights, ebins, want_acc: bool = True):
    """median depth [R,1] (+ accumulation [R,1]); no gradient (as in the reference's use)."""
    weights, ebins = _chk(weights, "weights"), _chk(ebins, "ebins")
    R, S = weights.shape
    depth = torch.empty((R, 1), device=weights.device, dtype=torch.float32)
    acc = torch.empty((R, 1), device=weights.device, dtype=torch.float32) if want_acc else None
    _launch("snf_composite_fwd", _p(None), _p(weights), _p(ebins), R, S, 1, _p(None), _p(acc), _p(depth),
                                      _stream())
    return depth, acc


@torch.no_grad()
def accumulation(weights) -> torch.Tensor:
    """AccumulationRenderer: sum_s w [R,1] from the compositing kernel (no colours, no depth)."""
    weights = _chk(weights, "weights")
    R, S = weights.shape
    acc = torch.empty((R, 1), device=weights.device, dtype=torch.float32)
    _launch("snf_composite_fwd", _p(None), _p(weights), _p(None), R, S, 1, _p(None), _p(acc), _p(None), _stream())
    return acc

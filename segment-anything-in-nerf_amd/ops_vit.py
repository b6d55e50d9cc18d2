"""Image-encoder pieces (csrc/vit.hip): inference only (split out of ops.py; `samnerf_amd.ops` re-exports everything here)."""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

import torch

from ._opcore import (ACT_BY_NAME, ACT_GELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, CONTRACT_L2, CONTRACT_LINF, CONTRACT_NONE, _L, _chk,
                      _launch, _linear_fwd_ws, _p, _stream)

# ---------------------------------------------------------------------------------------------
# image-encoder pieces (csrc/vit.hip) -- inference only
# ---------------------------------------------------------------------------------------------
@torch.no_grad()
def linear_nograd(x, w, b=None, act: int = ACT_NONE) -> torch.Tensor:
    """Y = act(X W^T + b) without autograd bookkeeping (split-K scratch when the shape asks for it)."""
    x, w = _chk(x, "x"), _chk(w, "w")
    N, I = x.shape
    O = w.shape[0]
    y = torch.empty((N, O), device=x.device, dtype=torch.float32)
    _linear_fwd_ws(x, w, b, N, I, O, act, y, _stream(), f"{I}x{O}")
    return y


@torch.no_grad()
def sam_preprocess(img, mean, std, size: int) -> torch.Tensor:
    """Sam.preprocess (modeling/sam.py:164-174): [B,C,h,w] uint8 / float -> normalised, zero-padded [B,C,size,size] fp32."""
    if not img.is_cuda:
        raise RuntimeError("sam_preprocess runs on the HIP kernels only (no CPU path)")
    img = img.contiguous()
    if img.dtype != torch.uint8:
        img = img.float()
    B, C, h, w = img.shape
    out = torch.empty((B, C, size, size), device=img.device, dtype=torch.float32)
    _launch("snf_sam_preprocess", _p(img), int(img.dtype == torch.uint8), B, C, h, w, size, _p(_chk(mean, "mean")),
            _p(_chk(std, "std")), _p(out), _stream())
    return out


@torch.no_grad()
def patchify(img, P: int) -> torch.Tensor:
    img = _chk(img, "img")
    B, Cin, S, _ = img.shape
    rows = torch.empty((B * (S // P) ** 2, Cin * P * P), device=img.device, dtype=torch.float32)
    _launch("snf_patchify", _p(img), B, Cin, S, P, _p(rows), _stream())
    return rows


@torch.no_grad()
def layernorm(x, weight, bias, eps: float, residual=None, want_sum: bool = False):
    """LayerNorm(x + residual) over the last axis of [N, C]; with want_sum also returns x + residual."""
    x = _chk(x, "x")
    N, C = x.shape
    y = torch.empty_like(x)
    s = torch.empty_like(x) if want_sum else None
    _launch("snf_layernorm", _p(x), _p(residual), N, C, _p(weight), _p(bias), float(eps), _p(s), _p(y), _stream())
    return (y, s) if want_sum else y


@torch.no_grad()
def window_partition(x, B: int, H: int, W: int, ws: int) -> torch.Tensor:
    C = x.shape[-1]
    nW = ((H + ws - 1) // ws) * ((W + ws - 1) // ws)
    out = torch.empty((B * nW * ws * ws, C), device=x.device, dtype=torch.float32)
    _launch("snf_window_partition", _p(x), B, H, W, C, ws, _p(out), _stream())
    return out


@torch.no_grad()
def window_merge_add(windows, shortcut, B: int, H: int, W: int, ws: int) -> torch.Tensor:
    C = shortcut.shape[-1]
    out = torch.empty((B * H * W, C), device=shortcut.device, dtype=torch.float32)
    _launch("snf_window_merge_add", _p(windows), _p(shortcut), B, H, W, C, ws, _p(out), _stream())
    return out


@torch.no_grad()
def attention(qkv, Bw: int, T: int, heads: int, n: int, rel_pos_h=None, rel_pos_w=None) -> torch.Tensor:
    """qkv [Bw*T, 3*C] -> [Bw*T, C]: softmax(hd^-0.5 q k^T + decomposed rel-pos) v per (window, head)."""
    qkv = _chk(qkv, "qkv")
    C = qkv.shape[1] // 3
    hd = C // heads
    rel = None
    if rel_pos_h is not None:
        assert rel_pos_h.shape == (2 * n - 1, hd) and rel_pos_w.shape == (2 * n - 1, hd), "rel-pos tables must have 2n-1 rows"
        rel = torch.empty((Bw * heads * T, 2 * n), device=qkv.device, dtype=torch.float32)
        _launch("snf_relpos", _p(qkv), Bw, T, heads, hd, n, _p(_chk(rel_pos_h, "rel_pos_h")), _p(_chk(rel_pos_w, "rel_pos_w")),
                _p(rel), _stream())
    out = torch.empty((Bw * T, C), device=qkv.device, dtype=torch.float32)
    _launch("snf_attention", _p(qkv), _p(rel), Bw, T, heads, hd, n, float(hd ** -0.5), _p(out), _stream(),
            units=4.0 * Bw * heads * T * T * hd)
    return out


@torch.no_grad()
def patch_unfold(x, p: int, k: int) -> torch.Tensor:
    x = _chk(x, "x")
    R, C = x.shape
    col = torch.empty((R, C * k * k), device=x.device, dtype=torch.float32)
    _launch("snf_patch_unfold", _p(x), R, p, C, k, _p(col), _stream())
    return col

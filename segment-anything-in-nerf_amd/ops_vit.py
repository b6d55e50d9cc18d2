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


FUSED_WINDOW_RELPOS = True  # attention_planes on small grids: snf_relpos inside the attention kernel (tests compare both)

# ---- the block GEMMs on pre-split operands (csrc/gemm_planes.hip) ------------------------------------------------------------
class Planes:
    """An activation matrix [M, K] as the GEMM operand: bf16 hi / lo k-blocked planes [K/8, M, 8] (x = hi + lo)."""
    __slots__ = ("hi", "lo", "M", "K")

    def __init__(self, hi: torch.Tensor, lo: torch.Tensor):
        assert hi.dtype == torch.bfloat16 and hi.shape == lo.shape and hi.dim() == 3 and hi.shape[2] == 8
        self.hi, self.lo, self.M, self.K = hi, lo, int(hi.shape[1]), int(hi.shape[0]) * 8

    @staticmethod
    def empty(M: int, K: int, device, zero: bool = False) -> "Planes":
        mk = torch.zeros if zero else torch.empty
        return Planes(mk((K // 8, M, 8), device=device, dtype=torch.bfloat16), mk((K // 8, M, 8), device=device, dtype=torch.bfloat16))

    def float(self) -> torch.Tensor:
        """[M, K] fp32 (tests)."""
        return (self.hi.float() + self.lo.float()).permute(1, 0, 2).reshape(self.M, self.K)


@torch.no_grad()
def split_weight_planes(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """A constant weight [Nc, K] -> its bf16 hi / lo planes [Nc, K] (once per checkpoint)."""
    w = _chk(w, "w")
    hi, lo = torch.empty(w.shape, device=w.device, dtype=torch.bfloat16), torch.empty(w.shape, device=w.device, dtype=torch.bfloat16)
    _launch("snf_split_planes", _p(w), w.numel(), hi.data_ptr(), lo.data_ptr(), _stream())
    return hi, lo


def split_weight_planes_kb(w: torch.Tensor) -> "Planes":
    """A constant weight [Nc, K] -> k-blocked planes [K/8, Nc, 8] (the layout of the activations): linear_planes then takes the
    kernel that stages both operands through LDS (Nc % 320 == 0 or Nc % 256 == 0)."""
    return split_planes_kb(w)


@torch.no_grad()
def split_planes_kb(x: torch.Tensor) -> Planes:
    x = _chk(x, "x")
    out = Planes.empty(x.shape[0], x.shape[1], x.device)
    _launch("snf_split_planes_kb", _p(x), x.shape[0], x.shape[1], out.hi.data_ptr(), out.lo.data_ptr(), _stream())
    return out


@torch.no_grad()
def layernorm_planes(x, weight, bias, eps: float, out: Planes, residual=None, want_sum: bool = False, grid=None):
    """LayerNorm(x + residual) written as GEMM operand planes; grid = (H, W, ws): rows go where window_partition puts the tokens
    (`out` must then have been allocated zeroed: the padded rows are never written)."""
    x = _chk(x, "x")
    N, C = x.shape
    s = torch.empty_like(x) if want_sum else None
    H, W, ws = grid if grid is not None else (0, 0, 0)
    _launch("snf_layernorm_planes", _p(x), _p(residual), N, C, _p(weight), _p(bias), float(eps), _p(s), out.hi.data_ptr(),
            out.lo.data_ptr(), out.M, H, W, ws, _stream())
    return (out, s) if want_sum else out


@torch.no_grad()
def layernorm_planes_merge(shortcut, windows, weight, bias, eps: float, out: Planes, B: int, H: int, W: int, ws: int):
    """x = shortcut + window_unpartition(windows); LayerNorm(x) as operand planes: `window_merge_add` + `layernorm_planes` in one
    launch.  Returns (planes, x)."""
    shortcut, windows = _chk(shortcut, "shortcut"), _chk(windows, "windows")
    N, C = shortcut.shape
    assert N == B * H * W and out.M == N and ws > 0
    s = torch.empty_like(shortcut)
    _launch("snf_layernorm_planes_merge", _p(shortcut), _p(windows), N, C, _p(weight), _p(bias), float(eps), _p(s), out.hi.data_ptr(),
            out.lo.data_ptr(), H, W, ws, _stream())
    return out, s


@torch.no_grad()
def attention_planes(qkv, Bw: int, T: int, heads: int, n: int, out: Planes, rel_pos_h=None, rel_pos_w=None) -> Planes:
    """`attention` with the result written as the projection GEMM's operand planes."""
    qkv = _chk(qkv, "qkv")
    C = qkv.shape[1] // 3
    hd = C // heads
    rel = None
    assert out.M == Bw * T and out.K == C
    if rel_pos_h is not None:
        assert rel_pos_h.shape == (2 * n - 1, hd) and rel_pos_w.shape == (2 * n - 1, hd), "rel-pos tables must have 2n-1 rows"
        if FUSED_WINDOW_RELPOS and 2 * n - 1 <= 32:  # the windowed blocks: the position terms are formed inside the attention kernel
            _launch("snf_attention_planes_rp", _p(qkv), _p(_chk(rel_pos_h, "rel_pos_h")), _p(_chk(rel_pos_w, "rel_pos_w")), Bw, T, heads,
                    hd, n, float(hd ** -0.5), out.hi.data_ptr(), out.lo.data_ptr(), _stream(), units=4.0 * Bw * heads * T * T * hd)
            return out
        rel = torch.empty((Bw * heads * T, 2 * n), device=qkv.device, dtype=torch.float32)
        _launch("snf_relpos", _p(qkv), Bw, T, heads, hd, n, _p(_chk(rel_pos_h, "rel_pos_h")), _p(_chk(rel_pos_w, "rel_pos_w")),
                _p(rel), _stream())
    _launch("snf_attention_planes", _p(qkv), _p(rel), Bw, T, heads, hd, n, float(hd ** -0.5), out.hi.data_ptr(), out.lo.data_ptr(),
            _stream(), units=4.0 * Bw * heads * T * T * hd)
    return out


@torch.no_grad()
def linear_planes(a: Planes, w_planes, bias, act: int = ACT_NONE, out: Optional[Planes] = None, shape=None) -> torch.Tensor:
    """act(A W^T + b) from operand planes; returns the fp32 [M, Nc] result, or writes `out` planes (and returns it) when given.
    shape = (rb, nb): force the (128 rb) x (32 nb) tile (benchmarks)."""
    kb = isinstance(w_planes, Planes)  # k-blocked weights (split_weight_planes_kb): the kernel with both operands through LDS
    if kb:
        wh, wl, Nc, K = w_planes.hi, w_planes.lo, w_planes.M, w_planes.K
        assert shape is None
    else:
        wh, wl = w_planes
        Nc, K = wh.shape
    assert K == a.K, (K, a.K)
    y = None
    if out is None:
        y = torch.empty((a.M, Nc), device=wh.device, dtype=torch.float32)
    else:
        assert out.M == a.M and out.K == Nc
    oh, ol = (out.hi.data_ptr(), out.lo.data_ptr()) if out is not None else (None, None)
    if kb:
        _launch("snf_linear_planes_kb_fwd", a.hi.data_ptr(), a.lo.data_ptr(), wh.data_ptr(), wl.data_ptr(), _p(bias), a.M, K, Nc, act,
                _p(y), oh, ol, _stream(), tag=f"{K}x{Nc}", units=2.0 * a.M * K * Nc)
    elif shape is not None:
        _launch("snf_linear_planes_fwd_shape", a.hi.data_ptr(), a.lo.data_ptr(), wh.data_ptr(), wl.data_ptr(), _p(bias), a.M, K, Nc,
                act, _p(y), oh, ol, int(shape[0]), int(shape[1]), _stream(), tag=f"{K}x{Nc}", units=2.0 * a.M * K * Nc)
    else:
        _launch("snf_linear_planes_fwd", a.hi.data_ptr(), a.lo.data_ptr(), wh.data_ptr(), wl.data_ptr(), _p(bias), a.M, K, Nc, act,
                _p(y), oh, ol, _stream(), tag=f"{K}x{Nc}", units=2.0 * a.M * K * Nc)
    return out if out is not None else y


@torch.no_grad()
def patch_unfold(x, p: int, k: int) -> torch.Tensor:
    x = _chk(x, "x")
    R, C = x.shape
    col = torch.empty((R, C * k * k), device=x.device, dtype=torch.float32)
    _launch("snf_patch_unfold", _p(x), R, p, C, k, _p(col), _stream())
    return col

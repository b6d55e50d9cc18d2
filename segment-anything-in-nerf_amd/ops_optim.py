"""Arena kernels: fused Adam, reachable-row Adam, counter-based fill (split out of ops.py; `samnerf_amd.ops` re-exports everything here)."""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

import torch

from ._opcore import (ACT_BY_NAME, ACT_GELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, CONTRACT_L2, CONTRACT_LINF, CONTRACT_NONE, _L, _chk,
                      _launch, _linear_fwd_ws, _p, _stream)

# ---------------------------------------------------------------------------------------------
# arena kernels
# ---------------------------------------------------------------------------------------------
@torch.no_grad()
def adam_step_(p, g, m, v, lr: float, beta1: float, beta2: float, eps: float, step: int, grad_scale: float = 1.0,
               zero_grad: bool = True) -> None:
    for t in (p, g, m, v):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
    _launch("snf_adam_step", _p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(beta1), float(beta2),
                                  float(eps), int(step), float(grad_scale), int(zero_grad), _stream(),
            units=32.0 * p.numel())  # p, g, m, v read + p, m, v, g(zero) written


@torch.no_grad()
def adam_step_rows_(p, g, m, v, rows, F: int, lr: float, beta1: float, beta2: float, eps: float, step: int,
                    grad_scale: float = 1.0, zero_grad: bool = True) -> None:
    """Adam on the listed rows only: rows int32 = element offsets (from the arena bases p, g, m, v) of F-float rows."""
    for t in (p, g, m, v):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
    assert rows.is_cuda and rows.dtype == torch.int32 and rows.is_contiguous()
    n = rows.numel()
    if n == 0:
        return
    _launch("snf_adam_step_rows", _p(p), _p(g), _p(m), _p(v), _p(rows), n, int(F), float(lr), float(beta1), float(beta2),
            float(eps), int(step), float(grad_scale), int(zero_grad), _stream(), units=32.0 * n * F)


@torch.no_grad()
def fill_uniform_(x, seed: int, lo: float, hi: float) -> None:
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
    _launch("snf_fill_uniform", _p(x), x.numel(), int(seed), float(lo), float(hi), _stream())

"""Flat fp32 parameter / gradient / Adam-state arenas.

All trainable tensors of the hot path live in ONE contiguous fp32 buffer per parameter group
(`proposal_networks`, `fields`, `sam_field`, `conv` -- the groups of samnerf/sam_model.py:330-335 and
nerfstudio/models/nerfacto.py:236-240), laid out for 288 GB of HBM:
  * `nn.Parameter`s are views into the arena, so `state_dict()` stays a plain tensor map (checkpoint
    contract of nerfstudio/engine/trainer.py:379-406);
  * the gradient arena is written in place by the backward kernels (`param.main_grad`), is the single
    RCCL all-reduce buffer of the data-parallel step, and is re-zeroed by the fused Adam pass itself;
  * every slice starts on a 256-byte boundary (dwordx4 vector access, no partial cache lines between tensors).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

ALIGN = 64  # floats (256 B)


def fill_uniform_reference(n: int, seed: int, lo: float, hi: float) -> np.ndarray:
    """numpy twin of the `snf_fill_uniform` kernel (splitmix64 counter hash -> 24-bit mantissa uniform)."""
    with np.errstate(over="ignore"):
        i = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return (np.float32(lo) + (np.float32(hi) - np.float32(lo)) * u).astype(np.float32)


class ParamGroupArena:
    """One parameter group = one contiguous (param, grad, exp_avg, exp_avg_sq) quadruple."""

    def __init__(self, name: str, shapes: Sequence[Tuple[str, Tuple[int, ...]]], device, with_optimizer_state=True):
        self.name = name
        self.offsets: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        off = 0
        for pname, shape in shapes:
            n = int(np.prod(shape))
            self.offsets[pname] = (off, tuple(shape))
            off += (n + ALIGN - 1) // ALIGN * ALIGN
        self.numel = max(off, ALIGN)
        self.param = torch.zeros((self.numel,), device=device, dtype=torch.float32)
        self.grad = torch.zeros_like(self.param)
        self.exp_avg = torch.zeros_like(self.param) if with_optimizer_state else None
        self.exp_avg_sq = torch.zeros_like(self.param) if with_optimizer_state else None
        self.tables: Dict[str, object] = {}  # pname -> hash-grid Encoding owning that slice (see engine.Optimizers._plan)

    def view(self, buf: torch.Tensor, pname: str) -> torch.Tensor:
        off, shape = self.offsets[pname]
        n = int(np.prod(shape))
        return buf[off:off + n].view(shape)

    def make_parameter(self, pname: str) -> torch.nn.Parameter:
        p = torch.nn.Parameter(self.view(self.param, pname))
        p.main_grad = self.view(self.grad, pname)  # backward kernels accumulate here
        p.arena_name = (self.name, pname)
        return p

    def nbytes(self) -> int:
        return self.param.numel() * 4

"""Optimiser / scheduler / trainer contract of the reference (nerfstudio/engine/optimizers.py:30-179,
schedulers.py:60-109, trainer.py:409-440) on the flat arenas: one fused Adam launch per parameter group,
gradient zeroing folded into it, learning rates from the reference's exponential-decay formula."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Type

import numpy as np
import torch

from . import ops
from .arena import ParamGroupArena


@dataclass
class AdamOptimizerConfig:
    lr: float = 0.0005
    eps: float = 1e-08
    max_norm: Optional[float] = None
    weight_decay: float = 0
    betas: tuple = (0.9, 0.999)


@dataclass
class ExponentialDecaySchedulerConfig:
    lr_pre_warmup: float = 1e-8
    lr_final: Optional[float] = None
    warmup_steps: int = 0
    max_steps: int = 100000
    ramp: str = "cosine"

    def lr_at(self, step: int, lr_init: float) -> float:
        """schedulers.py:86-105 (returns the learning rate, not the LambdaLR multiplier)."""
        lr_final = lr_init if self.lr_final is None else self.lr_final
        if step < self.warmup_steps:
            if self.ramp == "cosine":
                return (self.lr_pre_warmup + (1 - self.lr_pre_warmup)
                        * np.sin(0.5 * np.pi * np.clip(step / self.warmup_steps, 0, 1)))
            return self.lr_pre_warmup + (lr_init - self.lr_pre_warmup) * step / self.warmup_steps
        t = np.clip((step - self.warmup_steps) / (self.max_steps - self.warmup_steps), 0, 1)
        return float(np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t))


class Optimizers:
    """optimizers.py:92-179 with arenas instead of torch.optim objects."""

    def __init__(self, config: Dict[str, Any], arenas: Dict[str, ParamGroupArena]) -> None:
        missing = [k for k in arenas if k not in config]
        if missing:
            raise KeyError(f"no optimizer configured for parameter groups {missing}")
        self.config = config
        self.arenas = arenas
        import os
        self.sharded = os.environ.get("SNF_SHARDED_OPTIMIZER", "1") == "1"
        self.enabled = True
        self.shard_slices: Dict[str, List] = {}  # group -> [(lo, hi)] arena slices stepped separately (set by the trainer)
        self.step_count = {k: 0 for k in arenas}
        self.sched_step = {k: 0 for k in arenas}
        for k, a in arenas.items():
            if config[k]["optimizer"].weight_decay or config[k]["optimizer"].max_norm is not None:
                raise NotImplementedError("weight decay / clipping are not used by the samnerf configs")
            if a.exp_avg is None:
                raise ValueError("arena built without optimizer state")

    def lr(self, group: str) -> float:
        oc, sc = self.config[group]["optimizer"], self.config[group].get("scheduler")
        return oc.lr if sc is None else sc.lr_at(self.sched_step[group], oc.lr)

    def zero_grad_all(self) -> None:
        for a in self.arenas.values():
            a.grad.zero_()

    def optimizer_step(self, k: str, grad_scale: float = 1.0, zero_grad: bool = True) -> None:
        """One fused Adam launch over the whole arena of group `k` (on the current stream)."""
        a, oc = self.arenas[k], self.config[k]["optimizer"]
        self.step_count[k] += 1
        ops.adam_step_(a.param, a.grad, a.exp_avg, a.exp_avg_sq, self.lr(k), oc.betas[0], oc.betas[1], oc.eps,
                       self.step_count[k], grad_scale, zero_grad)

    def optimizer_step_params(self, k: str, first: int, last: int, grad_scale: float = 1.0, zero_grad: bool = True,
                              count_step: bool = True) -> None:
        """Adam over the contiguous arena slice holding parameters [first, last) of group `k` (registration order)."""
        a, oc = self.arenas[k], self.config[k]["optimizer"]
        names = list(a.offsets)
        lo = a.offsets[names[first]][0]
        hi = a.offsets[names[last]][0] if last < len(names) else a.numel
        if count_step:
            self.step_count[k] += 1
        ops.adam_step_(a.param[lo:hi], a.grad[lo:hi], a.exp_avg[lo:hi], a.exp_avg_sq[lo:hi], self.lr(k), oc.betas[0],
                       oc.betas[1], oc.eps, self.step_count[k], grad_scale, zero_grad)

    # -- data-parallel exchange + step (distributed.sharded_step) ---------------------------------------------------
    def exchange_and_step(self, k: str, first: Optional[int] = None, last: Optional[int] = None, count_step: bool = True,
                          extra: Optional[List[str]] = None) -> None:
        """Gradient mean over the ranks + Adam for group `k` (or its parameters [first, last)), on the current stream.
        world == 1: plain fused Adam.  world > 1: reduce-scatter, Adam on this rank's shard, all-gather of the parameters
        (or all-reduce + replicated Adam when `self.sharded` is off)."""
        from . import distributed as D
        if not self.enabled:  # measurement of the forward+backward alone (bench.py): gradients keep accumulating
            return
        a, oc = self.arenas[k], self.config[k]["optimizer"]
        if first is None:
            lo, hi = 0, a.numel
        else:
            names = list(a.offsets)
            lo = a.offsets[names[first]][0]
            hi = a.offsets[names[last]][0] if last < len(names) else a.numel
        if count_step:
            self.step_count[k] += 1
        scale, lr, t = 1.0 / D.world_size(), self.lr(k), self.step_count[k]
        p, g, m, v = a.param[lo:hi], a.grad[lo:hi], a.exp_avg[lo:hi], a.exp_avg_sq[lo:hi]

        def step_fn(s0: int, s1: int) -> None:
            if s1 > s0:
                ops.adam_step_(p[s0:s1], g[s0:s1], m[s0:s1], v[s0:s1], lr, oc.betas[0], oc.betas[1], oc.eps, t, scale, True)

        if self.sharded:
            D.sharded_step(p, g, step_fn)
        else:
            D.allreduce_gradients([g])
            step_fn(0, hi - lo)

    def consolidate_state(self) -> None:
        """Sharded runs keep each rank's Adam moments only for its shards: gather them before saving a checkpoint.
        (Slices stepped through `exchange_and_step(first, last)` are sharded per slice; the trainer passes the same slices.)"""
        from . import distributed as D
        if not self.sharded or D.world_size() == 1:
            return
        for k, a in self.arenas.items():
            for lo, hi in self.shard_slices.get(k, [(0, a.numel)]):
                D.gather_sharded_state(a.exp_avg[lo:hi])
                D.gather_sharded_state(a.exp_avg_sq[lo:hi])

    def optimizer_step_all(self, grad_scale: float = 1.0, zero_grad: bool = True) -> None:
        for k in self.arenas:
            self.optimizer_step(k, grad_scale, zero_grad)

    def scheduler_step_all(self, step: int) -> None:
        for k in self.sched_step:
            self.sched_step[k] += 1

    def state_dict(self) -> Dict[str, Any]:
        return {k: {"exp_avg": a.exp_avg, "exp_avg_sq": a.exp_avg_sq, "step": self.step_count[k],
                    "sched_step": self.sched_step[k]} for k, a in self.arenas.items()}

    def load_optimizers(self, loaded_state: Dict[str, Any]) -> None:
        for k, v in loaded_state.items():
            self.arenas[k].exp_avg.copy_(v["exp_avg"])
            self.arenas[k].exp_avg_sq.copy_(v["exp_avg_sq"])
            self.step_count[k], self.sched_step[k] = v["step"], v["sched_step"]

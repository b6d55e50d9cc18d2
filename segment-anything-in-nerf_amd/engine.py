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
        # Adam skips hash-table rows no input can ever address (exact: their g = m = v stay 0); SNF_ADAM_ALL_ROWS=1 disables
        self.skip_unreachable_rows = os.environ.get("SNF_ADAM_ALL_ROWS", "0") != "1"
        # the hash-grid backward of a table IS its Adam step where it sees the table's whole gradient (see arm_fused_adam)
        self.fuse_table_adam = os.environ.get("SNF_FUSED_TABLE_ADAM", "1") == "1"
        self._plans: Dict[str, list] = {}
        self._row_cuts: Dict[tuple, tuple] = {}
        self._tp_cache: Dict[str, list] = {}
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
        if zero_grad:
            self._adam_range(k, 0, a.numel, self.lr(k), oc.betas[0], oc.betas[1], oc.eps, self.step_count[k], grad_scale)
        else:
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

    # -- optimizer step folded into the hash-grid backward ---------------------------------------------------------------
    def arm_fused_adam(self) -> None:
        """Start of a train step: hand every hash table whose backward will see its WHOLE gradient (one rank, or
        table-parallel levels owned by this rank) Adam's hyper-parameters of this step.  ops._hashgrid_bwd_launch then
        applies the update inside the reduce pass of the backward for the dense levels (24 B per parameter instead of the
        gradient read-modify-write + the 32 B Adam pass) and reports the stepped levels; exchange_and_step skips them.
        The coarse reachable-row levels keep their own small launch."""
        from . import distributed as D
        for k, a in self.arenas.items():
            oc = self.config[k]["optimizer"]
            owned = {t[0] for t in self._tp_tables(k)}
            for pname, (off, shape) in a.offsets.items():
                enc = a.tables.get(pname)
                if enc is None:
                    continue
                # (a table evaluated more than once per step -- use_same_proposal_network with several proposal iterations --
                # gets its gradient from several backward launches: each would apply a full Adam step with a partial gradient)
                ok = (self.enabled and self.fuse_table_adam and a.param.is_cuda
                      and not getattr(enc.params, "no_fused_adam", False)
                      and (not D.collectives_on() or off in owned))
                if not ok:
                    enc.params._fused_adam = None
                    continue
                n, stride = enc.params.numel(), (1 << enc.log2_hashmap_size) * enc.n_features_per_level
                n_sparse = next(((seg[2] - off) // stride for seg in self._plan(k) if seg[0] == "rows" and seg[1] == off), 0)
                enc.params._fused_adam = ops.FusedAdam(
                    a.param[off:off + n], a.exp_avg[off:off + n], a.exp_avg_sq[off:off + n], self.lr(k), oc.betas[0],
                    oc.betas[1], oc.eps, self.step_count[k] + 1, 1.0 / D.world_size(), n_sparse)

    def disarm_fused_adam(self) -> None:
        """End of a train step (everything is enqueued): a backward run outside train_iteration must not step the tables
        with this step's hyper-parameters."""
        for a in self.arenas.values():
            for enc in a.tables.values():
                enc.params._fused_adam = None

    def _take_fused(self, k: str, lo: int, hi: int, t: int) -> list:
        """Arena element ranges of group `k` inside [lo, hi) that this step's backward has already stepped."""
        a, out = self.arenas[k], []
        for pname, (off, shape) in a.offsets.items():
            enc = a.tables.get(pname)
            fa = getattr(enc.params, "_fused_adam", None) if enc is not None else None
            if fa is None or fa.done is None or off >= hi or off + enc.params.numel() <= lo:
                continue
            if fa.step != t:
                raise RuntimeError(f"hash table {k}/{pname} was stepped by its backward for step {fa.step}, optimizer is at {t}")
            stride = (1 << enc.log2_hashmap_size) * enc.n_features_per_level
            out.append((off + fa.done[0] * stride, off + fa.done[1] * stride))
            fa.done = None
        return out

    # -- data-parallel exchange + step (distributed.sharded_step) ---------------------------------------------------
    def exchange_and_step(self, k: str, first: Optional[int] = None, last: Optional[int] = None, count_step: bool = True,
                          extra: Optional[List[str]] = None, done: Optional[list] = None) -> None:
        """Gradient mean over the ranks + Adam for group `k` (or its parameters [first, last)), on the current stream.
        world == 1: plain fused Adam.  world > 1: reduce-scatter, Adam on this rank's shard, all-gather of the parameters
        (or all-reduce + replicated Adam when `self.sharded` is off).
        done: arena element ranges this step's backward has already stepped, when the caller knows them (the static launch
        schedule, which does not arm `FusedAdam` objects per step); default: ask the armed tables."""
        from . import distributed as D
        ops.join_wgrad_stream()  # weight gradients of this task may still be running on the companion stream
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
        b1, b2, eps = oc.betas[0], oc.betas[1], oc.eps
        done = self._take_fused(k, lo, hi, t) if done is None else list(done)
        if not D.collectives_on():
            self._adam_range(k, lo, hi, lr, b1, b2, eps, t, scale, done)
            return
        # data-parallel: walk the plan.  Dense segments: sharded exchange + step (or all-reduce + replicated step);
        # row segments (reachable rows of coarse hash levels): only those rows travel, and every rank steps them.
        owned = self._tp_tables(k)
        for seg in self._plan(k):
            s0, s1 = max(lo, seg[1]), min(hi, seg[2])
            if s1 <= s0:
                continue
            tp = next((t for t in owned if t[0] <= seg[1] < t[1]), None)
            if tp is not None:
                # table-parallel hash table: the backward already summed every rank's samples into the levels this rank
                # owns, nothing else was touched -- step the owned levels, no exchange
                x0, x1 = max(s0, tp[2]), min(s1, tp[3])
                if x1 > x0:
                    self._adam_range(k, x0, x1, lr, b1, b2, eps, t, scale, done)
                continue
            if seg[0] == "dense":
                def step_fn(x0: int, x1: int, s0=s0) -> None:
                    if x1 > x0:
                        self._adam_range(k, s0 + x0, s0 + x1, lr, b1, b2, eps, t, scale)
                if self.sharded:
                    D.sharded_step(a.param[s0:s1], a.grad[s0:s1], step_fn)
                else:
                    D.allreduce_gradients([a.grad[s0:s1]])
                    step_fn(0, s1 - s0)
            else:
                F = seg[4]
                i0, i1 = self._cut(k, seg, s0, s1)
                if i1 > i0:
                    key = ("rowidx", k, seg[1], i0, i1)
                    if key not in self._row_cuts:
                        self._row_cuts[key] = torch.div(seg[3][i0:i1].long(), F, rounding_mode="floor")
                    D.exchange_rows(a.grad.view(-1, F), self._row_cuts[key])
                    self._adam_range(k, s0, s1, lr, b1, b2, eps, t, scale)

    def _cut(self, k: str, seg, s0: int, s1: int):
        """Index range of a row segment's sorted offset list that falls inside the arena elements [s0, s1)."""
        key = (k, seg[1], s0, s1)
        if key not in self._row_cuts:
            host = seg[5]
            self._row_cuts[key] = (int(np.searchsorted(host, s0, "left")), int(np.searchsorted(host, s1, "left")))
        return self._row_cuts[key]

    # -- reachable-row plan -------------------------------------------------------------------------------------------
    def _plan(self, k: str):
        """Arena `k` as an ordered list of segments: ("rows", start, end, offsets int32, F) for the coarse levels of a
        hash table (only the rows their lattice can reach are ever updated -- exact, see Encoding.active_rows) and
        ("dense", start, end) for everything else.  Built once."""
        if k in self._plans:
            return self._plans[k]
        a = self.arenas[k]
        segs, cur = [], 0
        tp_offs = {t[0] for t in self._tp_tables(k)}
        for pname, (off, shape) in a.offsets.items():
            enc = a.tables.get(pname)
            if enc is None:
                continue
            n_sparse, rows = enc.active_rows() if self.skip_unreachable_rows else (0, None)
            if n_sparse == 0 and off not in tp_offs:
                continue
            if off > cur:
                segs.append(("dense", cur, off))
            cur = off
            if n_sparse:
                F = enc.n_features_per_level
                cur = off + (n_sparse << enc.log2_hashmap_size) * F
                offsets = (rows * F + off).to(torch.int32).contiguous()
                segs.append(("rows", off, cur, offsets, F, (rows * F + off).cpu().numpy()))
            if off in tp_offs:  # segments never straddle the end of a table-parallel table
                end = off + enc.params.numel()
                if end > cur:
                    segs.append(("dense", cur, end))
                cur = end
        if cur < a.numel:
            segs.append(("dense", cur, a.numel))
        self._plans[k] = segs
        return segs

    def _tp_tables(self, k: str):
        """Hash tables of group `k` that are trained table-parallel (distributed.TableParallelLayout), as arena element
        ranges (table_start, table_end, owned_start, owned_end); [] on one rank.  Also hands each such parameter the
        collective that makes it whole again (ops.hashgrid calls it before a replicated evaluation)."""
        if k in self._tp_cache:
            return self._tp_cache[k]
        from . import distributed as D
        out = []
        a = self.arenas[k]
        if D.collectives_on():
            import torch.distributed as dist
            for pname, (off, shape) in a.offsets.items():
                enc = a.tables.get(pname)
                head = getattr(enc, "tp_head", None)
                if head is None:
                    continue
                encs, gi = head
                layout = ops.table_parallel_layout(tuple(e.spec for e in encs))
                if layout is None:
                    continue
                lo, hi = layout.owned_elements(dist.get_rank(), gi)
                n = enc.params.numel()
                out.append((off, off + n, off + lo, off + hi, layout, gi))
                enc.params._tp_refresh = (lambda flat=a.param[off:off + n], layout=layout, gi=gi:
                                          D.tp_refresh_table(flat, layout, gi))
        self._tp_cache[k] = out
        return out

    def adam_pieces(self, k: str, lo: int, hi: int, done=()):
        """The launches of a fused Adam pass over the arena elements [lo, hi) of group `k`: ("dense", x0, x1) element ranges
        and ("rows", offsets int32, F) reachable-row lists, following the plan and leaving out the ranges in `done` (already
        stepped by the hash-grid backward).  Shared by the eager pass below and the static step schedule (step_program.py)."""
        out = []
        for seg in self._plan(k):
            s0, s1 = max(lo, seg[1]), min(hi, seg[2])
            if s1 <= s0:
                continue
            if seg[0] == "dense":
                pieces = [(s0, s1)]
                for d0, d1 in done:
                    pieces = [q for x0, x1 in pieces for q in ((x0, min(x1, d0)), (max(x0, d1), x1)) if q[1] > q[0]]
                out.extend(("dense", x0, x1) for x0, x1 in pieces)
            else:
                # (reachable rows the backward stepped itself -- k_hg_reduce_sparse inside snf_hashgrid_bwd_presorted_adam_sp / _pair,
                #  whole leading levels -- are left out; the rows are sorted, so what remains of [s0, s1) is slices of the list)
                pieces = [(s0, s1)]
                for d0, d1 in done:
                    pieces = [q for x0, x1 in pieces for q in ((x0, min(x1, d0)), (max(x0, d1), x1)) if q[1] > q[0]]
                for x0, x1 in pieces:
                    i0, i1 = self._cut(k, seg, x0, x1)
                    if i1 > i0:
                        out.append(("rows", seg[3][i0:i1], seg[4]))
        return out

    def _adam_range(self, k: str, lo: int, hi: int, lr, b1, b2, eps, t, scale, done=()) -> None:
        """Fused Adam (+ gradient re-zeroing) on the arena elements [lo, hi) of group `k`, following the plan and leaving out
        the ranges in `done` (already stepped by the hash-grid backward)."""
        a = self.arenas[k]
        for piece in self.adam_pieces(k, lo, hi, done):
            if piece[0] == "dense":
                x0, x1 = piece[1], piece[2]
                ops.adam_step_(a.param[x0:x1], a.grad[x0:x1], a.exp_avg[x0:x1], a.exp_avg_sq[x0:x1], lr, b1, b2, eps, t,
                               scale, True)
            else:
                ops.adam_step_rows_(a.param, a.grad, a.exp_avg, a.exp_avg_sq, piece[1], piece[2], lr, b1, b2, eps, t,
                                    scale, True)

    def consolidate_state(self) -> None:
        """Sharded runs keep each rank's Adam moments only for its shards of the dense segments: gather them before saving
        a checkpoint (the layout is the one exchange_and_step uses: trainer slices x dense plan segments)."""
        from . import distributed as D
        if not D.collectives_on():
            return
        for k, a in self.arenas.items():
            owned = self._tp_tables(k)
            for toff, tend, _, _, layout, gi in owned:  # table-parallel tables: every owner broadcasts its levels
                for buf in (a.param, a.exp_avg, a.exp_avg_sq):
                    D.tp_refresh_table(buf[toff:tend], layout, gi)
            for pname in a.offsets:
                enc = a.tables.get(pname)
                if enc is not None and getattr(enc.params, "_tp_stale", False):
                    enc.params._tp_stale = False
            if not self.sharded:
                continue
            for lo, hi in self.shard_slices.get(k, [(0, a.numel)]):
                for seg in self._plan(k):
                    s0, s1 = max(lo, seg[1]), min(hi, seg[2])
                    if seg[0] == "dense" and s1 > s0 and not any(t[0] <= seg[1] < t[1] for t in owned):
                        D.gather_sharded_state(a.exp_avg[s0:s1])
                        D.gather_sharded_state(a.exp_avg_sq[s0:s1])

    def optimizer_step_all(self, grad_scale: float = 1.0, zero_grad: bool = True) -> None:
        for k in self.arenas:
            self.optimizer_step(k, grad_scale, zero_grad)

    def scheduler_step_all(self, step: int) -> None:
        for k in self.sched_step:
            self.sched_step[k] += 1

    def state_dict(self) -> Dict[str, Any]:
        """Per group: Adam's step count, the scheduler position and the moments PER PARAMETER, keyed by the parameter's name in
        registration order -- the order torch.optim numbers its `state` entries in, so entry i of the reference's
        `{group: optimizer.state_dict()}` (trainer.py:399-401) is the i-th name here.  The tensors are views of the arenas (no
        padding travels).  Multi-rank: call `consolidate_state()` on EVERY rank first (it is a collective)."""
        out = {}
        for k, a in self.arenas.items():
            state = {}
            for name, (off, shape) in a.offsets.items():
                n = int(np.prod(shape)) if len(shape) else 1
                state[name] = {"exp_avg": a.exp_avg[off:off + n].view(shape), "exp_avg_sq": a.exp_avg_sq[off:off + n].view(shape)}
            out[k] = {"step": self.step_count[k], "sched_step": self.sched_step[k], "state": state}
        return out

    def load_optimizers(self, loaded_state: Dict[str, Any]) -> None:
        for k, v in loaded_state.items():
            a = self.arenas[k]
            if "state" in v:
                missing = set(a.offsets) - set(v["state"])
                if missing:
                    raise KeyError(f"optimizer state of group {k!r} lacks {sorted(missing)}")
                for name, (off, shape) in a.offsets.items():
                    n = int(np.prod(shape)) if len(shape) else 1
                    a.exp_avg[off:off + n].copy_(v["state"][name]["exp_avg"].reshape(-1))
                    a.exp_avg_sq[off:off + n].copy_(v["state"][name]["exp_avg_sq"].reshape(-1))
            else:  # flat arenas (checkpoints written before the per-parameter layout)
                a.exp_avg.copy_(v["exp_avg"])
                a.exp_avg_sq.copy_(v["exp_avg_sq"])
            self.step_count[k], self.sched_step[k] = v["step"], v["sched_step"]

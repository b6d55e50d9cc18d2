"""The samnerf train step as a STATIC LAUNCH SCHEDULE (one process per GPU, three HIP streams, no autograd graph).

`Trainer.train_iteration` of the reference (nerfstudio/engine/trainer.py:409-440) runs, per step, the same fixed sequence
of ~75 kernels: samnerf/sam_model.py:226-328 forward, the losses of nerfstudio/models/nerfacto.py:316-344 and
sam_model.py:316-328, their backward, and Adam for four parameter groups.  Driving that sequence through
`torch.autograd.Function`s costs ~3.8 ms of Python per step (allocations, graph bookkeeping, stream contexts) -- as long as
the GPU needs for the step.  Here the sequence is written down ONCE:

  * every intermediate lives in a buffer allocated at build time (the schedule owns its memory, like a captured graph);
  * every C-ABI launch is recorded with its final arguments -- raw device pointers, sizes, the HIP stream it goes to;
    replaying a step is `for fn, args in plan: fn(*args)`;
  * the few values that change from step to step (learning rates, Adam step counts, the proposal-weight anneal) are
    patched into the recorded argument lists before the replay;
  * cross-stream edges are HIP events recorded / awaited at fixed points of the schedule; tensors shared between the main
    stream and a feature-head stream are double-buffered by step parity, so the next step's forward can run under the tails
    of this step's head tasks (what `Trainer.pipeline_steps` does with `record_stream` in the eager path).

The arithmetic is the eager path's, launch for launch (`ops.py` is the reference for every argument list below); the two
torch reductions of the eager loss dict (`rows.sum()`, `sum(losses)`) are `snf_nerf_loss_summary`, and autograd's
"scale one branch's gradient, add the branches" is `snf_add_scaled`.  tests/test_model_gpu.py runs both paths from the
same state and compares parameters and Adam moments after several steps.

Scope: `num_proposal_iterations == 1`, the fused nerfacto field, and the sorted hash-grid backward -- i.e. the
samnerf_distill / samnerf_no_distill method configs -- on one rank or many.

Multi-rank (one process per GPU, RCCL): the schedule is the same list with the collectives of the eager path recorded at fixed
points of it (every rank replays the same list from one host thread, so the ranks issue their collectives in the same order):
  * table-parallel feature grids (distributed.TableParallelLayout): all-gather of the top-K positions, this rank's level runs
    evaluated at every rank's samples straight into the all-to-all send buffer -- laid out [destination][level][sample][8], so
    what arrives is the level-major encoding the head's first layer reads -- the mirror-image all-to-all of its data gradient,
    and the fused backward + Adam (gradient scale 1/W) on the owned levels only;
  * everything replicated (field / proposal grids, MLPs, conv head): gradients land in the arenas and
    `Optimizers.exchange_and_step` -- reduce-scatter, Adam on the own shard, all-gather -- is called at its place in the list,
    on the stream of its task.
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib, ops
from . import distributed as D

_KERNEL, _PY = 0, 1
import os as _os

PLANAR_HEADS = True  # level-major hand-off between the feature grids and the head MLP (module constants: tests flip them)
# fixed-point reduce pass for the F = 8 feature tables as well: measured 12 % slower alone (1.10 vs 0.98 ms per step for the four
# launches) and no faster inside the concurrent step (3.73-3.77 vs 3.66-3.76 ms, r02o) -- at ~1 record per row the float
# reduce's in-bucket sort is cheap and its Adam stream already runs at 5 TB/s; off by default, kept for bit-reproducible runs
FX_F8 = False
FUSED_CHAIN_WGRAD = True  # weight gradients of the 64-wide nets inside the chain
# ... and their hidden activations formed again there from the inputs (bit-identical to the forward's) instead of being written
# by the forward and read back: needs the fused backward and the six-product forward arithmetic of gemm mode 1
CHAIN_RECOMPUTE = True
PROP_BWD_SIDE = None  # None: the proposal backward goes to the side stream only when there are no feature heads
if _os.environ.get("SNF_PROP_BWD_SIDE") in ("0", "1", "2"):  # (2: on the heads' shared weight-gradient stream)
    PROP_BWD_SIDE = int(_os.environ["SNF_PROP_BWD_SIDE"])
FEATURE_SORTS_ON_HEAD_STREAM = True
# The head's last layer is linear (no bias, no output activation) and the renderer after it is a weighted sum over the K samples
# of a ray (MeanRenderer, sam_model.py:126-137; the weights are detached, sam_model.py:260-277):
#     sum_k w_k (W h_k)  =  W (sum_k w_k h_k)
# so the schedule renders the HIDDEN activations [R*K, 256] -> [R, 256] first and runs the last layer -- forward, data gradient,
# weight gradient -- on R rows instead of R*K.  Same real-number result, fp32 rounding in a different order (1e-7).
MEAN_BEFORE_LAST_LAYER = True
ROWS_OPERAND = True  # ... and its gradient broadcast formed inside the GEMM loaders
# The reachable-row (coarse) levels of every hash table are reduced over compact row indices and stepped inside the table backward
# (k_hg_reduce_sparse, round 3): no gradient is written for them and no snf_adam_step_rows pass follows.  0: round-2 behaviour.
SPARSE_LEVELS = _os.environ.get("SNF_HG_SPARSE_LEVELS", "1") == "1"
# ... for the F = 2 tables too (measured: +0.07 ms on the field grid alone -- its coarse levels are bound by the 8-byte gathers of
# the staged gradient, which the compact reduce does not remove, and the quad-merged bucket-wide kernel already avoids the pile-up)
SPARSE_LEVELS_F2 = False
PAIR_GRID_BWD = True  # both feature grids of a head in one table-backward launch
# (both grids in one FORWARD launch as well was built and measured slower inside the step -- 2.56 vs 2.52 ms, same box, twice: one 24-level
#  launch sits 0.15 ms on the head's chain where the two 12-level launches take 0.04 each -- and removed, profiles/EXPERIMENTS.md r05)
# The heads' weight gradients feed nothing but the optimizer step of their layers, while their stream's next kernels wait behind them.
# ONE more stream (the fourth: one per hardware queue of the runtime's default four) takes the weight-gradient launches of BOTH heads and
# the Adam launches that need them; a head's stream waits for it once, at the end of its task.  (Round 1's per-task companion streams
# made seven streams, which the runtime multiplexes onto four queues -- the slow mode of DESIGN 5.)
# Same box, alternating, bench.py --steps 20 --warmup 5: 2.544 2.515 2.532 2.551 2.553 (3.364) 2.529 2.534 ms without it, 2.510 2.451 2.488
# 2.486 2.478 2.452 2.524 2.513 2.552 2.509 2.460 2.474 with it (median 2.54 -> 2.49, no slow run in twelve): on by default, one rank.
WGRAD_STREAM = _os.environ.get("SNF_WGRAD_STREAM", "1") == "1"
FUSED_MEAN_EPILOGUE = True  # ... and the mean itself in the hidden layer's GEMM epilogue
# Without feature heads the step is ONE dependency chain on the main stream.  Its head -- sampling, the proposal network, the
# resampling, the positions and the backward sorts -- needs last step's PROPOSAL update only, not the field's: it is recorded on
# the side stream (where the proposal backward + Adam of the previous step already ran) and so runs under the tail of the
# previous step's field backward.  What it produces and the main stream reads late is double-buffered by step parity.
XSTEP_PROLOGUE = True
# the F = 2 grids (field, proposal) sorted into x-pair records: one record and one staged-gradient gather per x-neighbour corner pair
# (snf_hashgrid_sort_xp / snf_hashgrid_bwd_presorted_adam_xp; same sums bit for bit)
XPAIR_RECORDS = True
# Many ranks: the gradient exchange + optimizer of the replicated groups recorded into the schedule (collectives as list entries, Adam
# as recorded launches) instead of `Optimizers.exchange_and_step` enqueueing its ~15 torch / C-ABI calls per group eagerly each step
RECORD_EXCHANGE = True
# The colour net's input row cat(SH16(d), geo) (nerfacto_field.py:336-343) formed inside its forward / recomputing backward
# (snf_mlp64_fwd_sh / snf_mlp64_bwd_fused_sh) instead of written by snf_head_input and read back twice; the backward writes the
# geo columns' gradient only ([N, 16] instead of [N, 32]).  Needs the recomputing fused backward (gemm mode 1).
FUSED_SH_INPUT = _os.environ.get("SNF_FUSED_SH_INPUT", "1") == "1"
FUSED_DENSITY = _os.environ.get("SNF_FUSED_DENSITY", "1") == "1"  # trunc_exp of the base net's output 0 in its epilogue
# steps without a proposal update: the proposal stage's forward as ONE launch (no encoding / hidden activations stored for a backward)
FUSED_PROP_FWD = _os.environ.get("SNF_FUSED_PROP_FWD", "1") == "1"
# Non-finite-gradient guard (trainer.py:419-437, optimizers.py:138-149: GradScaler.step skips an optimizer whose gradients hold an inf / NaN).
# Adam is fused into the table backward here, so a poisoned step cannot be vetoed afterwards: each group of losses owns a device record
# {veto, skipped} (include/samnerf_hip.h, snf_step_guard) -- written behind its loss kernels by snf_guard_update, read by every
# optimizer-side launch of the parameters those losses reach.  Domains: "nerf" (rgb + interlevel + distortion -> `fields`,
# `proposal_networks`), "sam" (the SAM head's loss -> its slice of `sam_field` and `conv`), "clipseg" (its slice of `sam_field`).  The
# reference has ONE optimizer for both heads; a guard per head keeps the two head streams independent (a shared verdict would make
# each head's table backward wait for the other head's loss).  The host never reads a record inside a step: `opt.step_count` counts
# optimistically and the kernels subtract `skipped` on the device; `StepProgram.guard_report()` reads them (synchronising).
# Many ranks: the verdict has to be the same on every rank (one rank's NaN reaches all of them through the gradient exchange), which costs
# three 4-byte MAX-all-reduces per step, each issued where its loss is ready -- on torch's one RCCL stream they queue in host order between
# the step's real collectives, and a verdict whose loss comes late (the SAM head's) would hold the collectives issued after it.  That has
# never been measured on real links, so the guard is ON for one rank and OPT-IN for many (SNF_STEP_GUARD=all); 0 records no guard at all.
_SG = _os.environ.get("SNF_STEP_GUARD", "1")
STEP_GUARD = _SG in ("1", "all")
STEP_GUARD_MULTI = _SG == "all"


# SNF_ABLATE_SKIP="key,key": launches whose key contains one of these are NOT issued (results are garbage; timing probe only)
_SKIP = tuple(k for k in _os.environ.get("SNF_ABLATE_SKIP", "").split(",") if k)
# SNF_CU_ROUTE="key=cus,key=cus": launches whose key contains `key` are issued on a companion stream confined to the first `cus` CUs
# (ops.masked_stream), fenced by events against the task stream they belong to -- CU partitioning per KERNEL instead of per task.
# (`key=p1`: a companion stream of LOWER priority instead of fewer CUs, `key=p-1` of higher priority)
CU_ROUTE = tuple((k, v) for k, v in (kv.split("=") for kv in _os.environ.get("SNF_CU_ROUTE", "").split(",") if "=" in kv))


class _Plan:
    __slots__ = ("entries", "dyn")

    def __init__(self) -> None:
        self.entries: list = []          # [kind, fn, args(list), key, units, stream]
        self.dyn: Dict[tuple, list] = {}  # dynamic value key -> [(args list, index)]


class StepProgram:
    """See the module docstring.  Built lazily by `Trainer.train_iteration`; `run(step)` enqueues one train step."""

    # ------------------------------------------------------------------------------------------------------------
    @staticmethod
    def unsupported_reason(trainer) -> Optional[str]:
        from .model import SAMModel
        from .pipeline import SyntheticSAMDataManager  # noqa: F401
        model = trainer.pipeline.model
        c = model.config
        if not torch.cuda.is_available():
            return "no GPU"
        if not isinstance(model, SAMModel):
            return "not a SAMModel"
        if c.num_proposal_iterations != 1 or c.use_same_proposal_network:
            return "more than one proposal iteration"
        if not model.field._fusable():
            return "nerfacto field is not the fused 32-64-16 / 31-64-64-3 shape"
        if ops.HASHGRID_BWD_MODE != "sorted" or not ops.PLANAR_FIELD_ENCODING:
            return "non-default hash-grid backward mode"
        prop = model.proposal_networks[0].mlp_base
        if not ops.mlp_tiny_supported(prop.network.n_input_dims, prop.network.weights(), prop.network.output_activation):
            return "proposal network is not the 10-16-1 shape"
        R = trainer.pipeline.datamanager.config.train_num_rays_per_batch
        P, S = c.num_proposal_samples_per_ray[0], c.num_nerf_samples_per_ray
        if R * max(P, S) > ops.HASHGRID_BWD_MAX_SAMPLES or 8 * 16 * R * S >= (1 << 32):
            return "batch beyond one sorted-backward launch"
        if (R * 3) % 64:
            return "ray count not a multiple of 64"
        if c.distill_sam:
            if c.patch_size > 1 and R % (c.patch_size ** 2):
                return "ray count not a multiple of the patch area"
            if c.use_dino_feature:
                return "dino head"
        if not model.arenas:
            return "parameters are not in arenas"
        if D.collectives_on() and c.distill_sam:
            sf = model.sam_field
            heads = [list(sf.clip_encs)] + ([list(sf.clipseg_encs)] if c.use_clipseg_feature else [])
            layouts = [ops.table_parallel_layout(tuple(e.spec for e in h)) for h in heads]
            if any(l is not None for l in layouts):
                total = [sum(e.n_output_dims for e in h) for h in heads]
                if not (PLANAR_HEADS and int(_lib.load().snf_get_gemm_mode()) >= 1 and all(l is not None for l in layouts)
                        and all(t % 16 == 0 and 64 <= t <= 256 for t in total)):
                    return "table-parallel heads need the level-major head path"
                if R * c.num_sam_samples * D.world_size() > ops.HASHGRID_BWD_MAX_SAMPLES:
                    return "gathered feature samples beyond one sorted-backward launch"
        return None

    def __init__(self, trainer) -> None:
        self.tr = trainer
        self.model = trainer.pipeline.model
        self.opt = trainer.optimizers
        self.cfg = self.model.config
        self.dev = self.model.device
        self.lib = _lib.load()
        c = self.cfg
        self.R = trainer.pipeline.datamanager.config.train_num_rays_per_batch
        self.P, self.S = c.num_proposal_samples_per_ray[0], c.num_nerf_samples_per_ray
        self.distill = bool(c.distill_sam)
        self.K = c.num_sam_samples if self.distill else 0
        self.heads = (["sam"] + (["clipseg"] if c.use_clipseg_feature else [])) if self.distill else []
        self.bufs: Dict[str, torch.Tensor] = {}
        self.plans: Dict[tuple, _Plan] = {}
        self.events: Dict[str, torch.cuda.Event] = {}
        self._head_busy: Dict[tuple, bool] = {}  # (parity, head) -> a task of that parity has been enqueued
        self.count = 0
        self.multi = bool(D.collectives_on())
        self.world = D.world_size() if self.multi else 1
        self.rank = torch.distributed.get_rank() if self.multi else 0
        self._tp_params: list = []  # tables whose levels are spread over the ranks (stale elsewhere after a step)
        self._head_domain: Optional[str] = None  # the head whose task is being recorded (guard domain of `sam_field` / `conv`)
        self._own_sort_stream = None
        self._feat_sorted = None
        self._feat_sort_stream_id = None
        # reference semantics of the optimizer on steps where the proposal network gets no gradient: the reference pins
        # torch < 2 (requirements.txt:32), whose zero_grad() zero-fills -- Adam still steps the group (moments decay)
        self.main = torch.cuda.current_stream()

    # ------------------------------------------------------------------------------------------------------------
    # memory
    def buf(self, name: str, shape, dtype=torch.float32, parity: Optional[int] = None, zero: bool = False) -> torch.Tensor:
        key = name if parity is None else f"{name}@{parity}"
        t = self.bufs.get(key)
        shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        if t is None:
            t = (torch.zeros if zero else torch.empty)(shape, device=self.dev, dtype=dtype)
            self.bufs[key] = t
        elif tuple(t.shape) != shape or t.dtype != dtype:
            raise RuntimeError(f"schedule buffer {key!r} exists as {tuple(t.shape)} {t.dtype}, requested {shape} {dtype}")
        return t

    def event(self, name: str) -> torch.cuda.Event:
        ev = self.events.get(name)
        if ev is None:
            ev = self.events[name] = torch.cuda.Event()
        return ev

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.bufs.values())

    # ------------------------------------------------------------------------------------------------------------
    # recording helpers (valid while a plan is being built)
    def _k(self, st, name: str, *args, tag: str = "", units: float = 0.0, dyn: Optional[dict] = None) -> None:
        """Record one C-ABI launch on stream `st`; the stream handle is appended as the last argument."""
        fn = getattr(self.lib, name)
        a = [x.data_ptr() if isinstance(x, torch.Tensor) else x for x in args]
        key_ = name + ("/" + tag if tag else "")
        task_st = st
        if CU_ROUTE and self._overlap:
            cus = next((c for k, c in CU_ROUTE if k in key_), None)
            if cus is not None:  # this launch runs on `cus` CUs, between two event edges against its task stream
                if cus.startswith("lane"):  # `key=lane<name>`: ONE shared stream for every routed launch of every task (a lane by kernel type)
                    st = ops.make_stream(cus)
                else:
                    st = (ops.priority_stream(int(cus[1:]), ops.stream_name(task_st)) if cus.startswith("p")
                          else ops.masked_stream(int(cus), ops.stream_name(task_st)))
                self._route_n = getattr(self, "_route_n", 0) + 1
                if st.stream_id == task_st.stream_id:
                    st = task_st
                else:
                    self._edge(task_st, st, f"cu_route_in_{self._route_n}")
        a.append(st.cuda_stream)
        self._plan.entries.append([_KERNEL, fn, a, key_, units, st])
        for key, idx in (dyn or {}).items():
            self._plan.dyn.setdefault(key, []).append((a, idx))
        if st is not task_st:
            self._edge(st, task_st, f"cu_route_out_{self._route_n}")

    def _py(self, fn, *args) -> None:
        self._plan.entries.append([_PY, fn, list(args), None, 0.0, None])

    # -- the non-finite-gradient guard (STEP_GUARD above) ----------------------------------------------------------------------
    def _guard(self, domain: str) -> Optional[torch.Tensor]:
        if not STEP_GUARD or (self.multi and not STEP_GUARD_MULTI):
            return None
        return self.buf(f"guard_{domain}", (2,), torch.int32, zero=True)

    def _guard_of_group(self, group: str) -> Optional[torch.Tensor]:
        """The record the optimizer-side launches of `group` recorded NOW obey: the head being recorded for `sam_field` / `conv`."""
        if group in ("sam_field", "conv"):
            return self._guard(self._head_domain) if self._head_domain else None
        return self._guard("nerf")

    def _dense_ranges(self, group: str, first: int = 0, last: Optional[int] = None):
        """Contiguous arena ranges [(offset, elements)] of the parameters first .. last-1 of `group` that are NOT hash tables."""
        a = self.opt.arenas[group]
        names = list(a.offsets)
        out: list = []
        for i in range(first, len(names) if last is None else last):
            if names[i] in a.tables:
                continue
            off = a.offsets[names[i]][0]
            end = a.offsets[names[i + 1]][0] if i + 1 < len(names) else a.numel
            if out and out[-1][0] + out[-1][1] == off:
                out[-1] = (out[-1][0], out[-1][1] + end - off)
            else:
                out.append((off, end - off))
        return out

    def _guard_update(self, st, domain: str, values: torch.Tensor, n: int = 1) -> None:
        """Behind the loss kernels of `domain` on their stream: commit last step's verdict, judge this step's -- from the loss values
        and from the dense weights of the domain (an inf weight can hide from the loss behind a ReLU and still make the gradients NaN)."""
        g = self._guard(domain)
        if g is None:
            return
        self._k(st, "snf_guard_update", values, n, g)
        if domain == "nerf":
            spans = [("fields", r) for r in self._dense_ranges("fields")] + \
                    [("proposal_networks", r) for r in self._dense_ranges("proposal_networks")]
        else:
            lo_i, hi_i = self.tr._head_param_ranges()[domain]
            spans = [("sam_field", r) for r in self._dense_ranges("sam_field", lo_i, hi_i)]
            if domain == "sam" and "conv" in self.opt.arenas and self.cfg.patch_size > 1:
                spans += [("conv", r) for r in self._dense_ranges("conv")]
        for group, (off, cnt) in spans:
            self._k(st, "snf_guard_scan", self.opt.arenas[group].param[off:off + cnt], cnt, g)
        if self.multi:  # every rank obeys the same verdict: one rank's NaN reaches all of them through the gradient exchange
            self._py(self._comm, st, D._all_reduce_max, g[0:1])

    def _kg(self, st, group: str, name: str, *args, **kw) -> None:
        """`_k` for an optimizer-side entry point: issued with the group's guard bound on this host thread (snf_step_guard)."""
        g = self._guard_of_group(group)
        if g is None:
            self._k(st, name, *args, **kw)
            return
        self._py(self.lib.snf_step_guard, g.data_ptr())
        self._k(st, name, *args, **kw)
        self._py(self.lib.snf_step_guard, None)

    def fold_guards(self) -> None:
        """Vetoed steps into the host's counters (call with the streams joined: checkpoints, Trainer.synchronize users): a group's
        step_count drops by the steps its guard vetoed and the records start again at zero, so that what is saved is what
        torch.optim.Adam's state['step'] would hold.  (`sam_field` has two guards, one per head: the larger count is taken.)"""
        rep = self.guard_report()
        if not rep:
            return
        n = {d: r["veto"] + r["skipped"] for d, r in rep.items()}
        heads = max([n.get(h, 0) for h in self.heads] or [0])
        for g, k in (("fields", n.get("nerf", 0)), ("proposal_networks", n.get("nerf", 0)), ("sam_field", heads), ("conv", n.get("sam", 0))):
            if g in self.opt.step_count and k:
                self.opt.step_count[g] = max(self.opt.step_count[g] - k, 0)
        for name, t in self.bufs.items():
            if name.startswith("guard_"):
                t.zero_()

    def guard_report(self) -> Dict[str, Dict[str, int]]:
        """{domain: {veto, skipped}} read from the device (synchronises): `skipped` = steps whose losses were inf / NaN so far (their
        optimizer step did not happen and does not count), `veto` = the last judged step is one of them."""
        out = {}
        for name, t in self.bufs.items():
            if name.startswith("guard_"):
                v = t.cpu().tolist()
                out[name[6:]] = {"veto": int(v[0]), "skipped": int(v[1])}
        return out

    def _edge(self, src, dst, name: str) -> None:
        """dst waits for everything enqueued on src so far (no-op when they are the same stream)."""
        if src.stream_id == dst.stream_id:
            return
        ev = self.event(name)
        self._py(ev.record, src)
        self._py(dst.wait_event, ev)

    @staticmethod
    def _off(t: torch.Tensor, nbytes: int) -> int:
        return t.data_ptr() + nbytes

    # ------------------------------------------------------------------------------------------------------------
    def _weights_of(self, net):
        return net.weights()

    def _table_adam(self, enc, group: str):
        """(param, grad, exp_avg, exp_avg_sq flat views of the table, first fused level) of a hash table in its arena."""
        a = self.opt.arenas[group]
        pname = next(n for n, e in a.tables.items() if e is enc)
        off, _ = a.offsets[pname]
        n = enc.params.numel()
        stride = (1 << enc.log2_hashmap_size) * enc.n_features_per_level
        n_sparse = next(((seg[2] - off) // stride for seg in self.opt._plan(group) if seg[0] == "rows" and seg[1] == off), 0)
        return (a.param[off:off + n], a.grad[off:off + n], a.exp_avg[off:off + n], a.exp_avg_sq[off:off + n], n_sparse,
                (off + n_sparse * stride, off + n))

    @staticmethod
    def _xpair(F: int, N: int, L: int, T: int) -> bool:
        """x-pair records for this grid?  F = 2 and what the fixed-point reduce takes (N even, L <= 64); tables of at least 2^18 rows:
        below that a bucket's accumulators are replicated per lane group and the pair reduce measured slower (proposal grid, T = 17:
        sort 52 -> 49 us but reduce 64 -> 70 us)."""
        return bool(XPAIR_RECORDS and F == 2 and N % 2 == 0 and L <= 64 and T >= 18)

    def _f2_sort(self, F: int, N: int, L: int, T: int) -> str:
        return "snf_hashgrid_sort_xp" if self._xpair(F, N, L, T) else "snf_hashgrid_sort"

    def _grid_bwd(self, st, g, N, enc, group, ld, col, sorted_ws, stage, with_opt: bool, done: list, run=None,
                  grad_scale: float = 1.0) -> None:
        """Table-gradient backward of one grid from the presorted records (+ Adam of its dense levels when with_opt).
        run = (first level, levels): only that level run of the table (table-parallel ownership); `g` / the sort are the run's."""
        L, F, T = enc.n_levels, enc.n_features_per_level, enc.log2_hashmap_size
        p, gbuf, m, v, n_sparse, fused_range = self._table_adam(enc, group)
        sc = enc.scalings
        tag = f"F{F}L{L}"
        if run is not None:
            l0, L = run
            a, e = (l0 << T) * F, ((l0 + L) << T) * F
            p, gbuf, m, v = p[a:e], gbuf[a:e], m[a:e], v[a:e]
            sc = ops._sc_run(enc.scalings, l0, L)
            first_fused = fused_range[0] - n_sparse * ((1 << T) * F)  # arena offset of the table
            n_sparse = min(max(n_sparse - l0, 0), L)
            fused_range = (first_fused + ((l0 + n_sparse) << T) * F, first_fused + ((l0 + L) << T) * F)
            tag = f"F{F}L{L}tp"
        nrun = ops.hashgrid_run_levels(sc) if F == 2 else 0
        fuse = with_opt and self.opt.fuse_table_adam and n_sparse < L
        if run is None and self._xpair(F, N, L, T):
            # the workspace holds x-pair records (self._f2_sort): their own backward entry, Adam of the levels >= n_sparse in its reduce
            oc = self.opt.config[group]["optimizer"]
            fused = ((L - n_sparse) << T) * F if fuse else 0
            self._kg(st, group, "snf_hashgrid_bwd_presorted_adam_xp", g, N, L, T, ld, col, nrun, gbuf, sorted_ws, stage, n_sparse if fuse else L,
                    p if fuse else None, m if fuse else None, v if fuse else None, 0.0, float(oc.betas[0]), float(oc.betas[1]),
                    float(oc.eps), 1, float(grad_scale), tag=tag,
                    units=float(N) * 8 * F * 4 * (L + (n_sparse if fuse else L)) + 24.0 * fused,
                    dyn={("lr", group): 14, ("t", group): 18} if fuse else None)
            if fuse:
                done.append(fused_range)
            return
        sp = self._sparse_lists(enc, N, n_sparse) if (run is None and not (F == 8 and FX_F8)) else None
        if sp is not None:
            # reachable-row levels through the compact fixed-point reduce; with the optimizer on this launch steps the WHOLE table
            step_it = bool(with_opt and self.opt.fuse_table_adam)
            oc = self.opt.config[group]["optimizer"]
            ns, rows, start, longest = sp  # (ns <= n_sparse: the levels in between keep the gradient write + row-Adam pass)
            scratch = self.buf(f"sp_scratch_{id(enc)}", (64,), torch.int32)
            n_tab, stride = (L << T) * F, (1 << T) * F
            fused = ((L - n_sparse) << T) * F if step_it else 0
            reach_params = int(rows.numel()) * F if step_it else 0
            left = (n_sparse - ns) if step_it else L  # levels whose gradient is written back
            self._kg(st, group, "snf_hashgrid_bwd_presorted_adam_sp", g, N, L, F, T, ld, col, nrun, gbuf, sorted_ws, stage,
                    n_sparse if step_it else L, p if step_it else None, m if step_it else None, v if step_it else None, 0.0,
                    float(oc.betas[0]), float(oc.betas[1]), float(oc.eps), 1, float(grad_scale), rows, start, ns, longest,
                    1 if step_it else 0, scratch, tag=tag,
                    units=float(N) * 8 * F * 4 * (L + left) + 24.0 * (fused + reach_params), dyn={("lr", group): 15, ("t", group): 19})
            if step_it:
                t0 = fused_range[1] - n_tab
                done.append((t0, t0 + ns * stride))  # the reachable rows of the leading ns levels
                if n_sparse < L:
                    done.append(fused_range)         # the dense levels
            return
        if F == 8 and FX_F8 and N % 2 == 0 and L <= 64:
            # fixed-point reduce (order-independent sums, no in-bucket sort); scratch private to this launch: the sorted
            # workspace is shared by the SAM and ClipSeg heads, whose backward passes run concurrently
            oc = self.opt.config[group]["optimizer"]
            scratch = self.buf(f"fx_scratch_{id(enc)}_{run}", (64,), torch.int32)
            from_level = n_sparse if fuse else L
            fused = ((L - from_level) << T) * F
            self._kg(st, group, "snf_hashgrid_bwd_presorted_adam_fx", g, N, L, F, T, ld, col, nrun, gbuf, sorted_ws, stage, from_level,
                    p, m, v, 0.0, float(oc.betas[0]), float(oc.betas[1]), float(oc.eps), 1, float(grad_scale), scratch,
                    tag=tag, units=float(N) * 8 * F * 4 * (L + from_level) + 24.0 * fused,
                    dyn={("lr", group): 15, ("t", group): 19})
            if fuse:
                done.append(fused_range)
            return
        if fuse:
            oc = self.opt.config[group]["optimizer"]
            fused = ((L - n_sparse) << T) * F
            self._kg(st, group, "snf_hashgrid_bwd_presorted_adam", g, N, L, F, T, ld, col, nrun, gbuf, sorted_ws, stage, n_sparse,
                    p, m, v, 0.0, float(oc.betas[0]), float(oc.betas[1]), float(oc.eps), 1, float(grad_scale), tag=tag,
                    units=float(N) * 8 * F * 4 * (L + n_sparse) + 24.0 * fused, dyn={("lr", group): 15, ("t", group): 19})
            done.append(fused_range)
        else:
            self._k(st, "snf_hashgrid_bwd_presorted", g, N, L, F, T, ld, col, nrun, gbuf, sorted_ws, stage, tag=tag)

    # -- multi-rank pieces, executed at their place in the replayed list (every rank issues them in the same order) -------
    def _collective(self, st, kind: str, out: torch.Tensor, inp: torch.Tensor) -> None:
        with torch.cuda.stream(st):
            (D._all_gather_into if kind == "all_gather" else D._all_to_all)(out, inp)

    def _copy(self, st, dst: torch.Tensor, src: torch.Tensor) -> None:
        with torch.cuda.stream(st):
            dst.copy_(src)

    def _exchange(self, st, group: str, first, last, count_step: bool, done) -> None:
        """Gradient mean over the ranks + Adam of one replicated group (or slice of it) on stream `st`."""
        with torch.cuda.stream(st):
            self.opt.exchange_and_step(group, first, last, count_step=count_step, done=done)

    def _opt_step(self, st, group: str, lo: int, hi: int, done, first=None, last=None, count_step: bool = True) -> None:
        """Optimizer step of arena elements [lo, hi) of `group`: recorded Adam launches on one rank; on many ranks the gradient
        exchange of `Optimizers.exchange_and_step` recorded piece by piece (RECORD_EXCHANGE) -- its collectives as entries of the
        list, its Adam passes as recorded C-ABI launches -- or, with the switch off, that method called at its place in the list."""
        if not self.multi:
            self._adam(st, group, lo, hi, done)
        elif RECORD_EXCHANGE:
            self._record_exchange(st, group, lo, hi, list(done))
        else:
            self._py(self._exchange, st, group, first, last, count_step, list(done))

    def _comm(self, st, fn, *args) -> None:
        with torch.cuda.stream(st):
            fn(*args)

    def _zero(self, st, t: torch.Tensor) -> None:
        with torch.cuda.stream(st):
            t.zero_()

    def _record_exchange(self, st, group: str, lo: int, hi: int, done) -> None:
        """`Optimizers.exchange_and_step` for arena elements [lo, hi) of `group`, written down once (engine.py:155-213 is the
        reference for every branch): table-parallel tables -> Adam on the owned levels, no exchange; dense segments -> reduce-scatter,
        Adam on this rank's 1/W shard, the rest of the gradient slice cleared, all-gather of the parameters (or all-reduce +
        replicated Adam); reachable-row segments -> only the listed rows travel, every rank steps them.  Gradients arrive SUMMED:
        1/W is folded into the Adam launches."""
        opt, a, W = self.opt, self.opt.arenas[group], self.world
        scale = 1.0 / W
        owned = opt._tp_tables(group)
        for seg in opt._plan(group):
            s0, s1 = max(lo, seg[1]), min(hi, seg[2])
            if s1 <= s0:
                continue
            tp = next((t for t in owned if t[0] <= seg[1] < t[1]), None)
            if tp is not None:
                x0, x1 = max(s0, tp[2]), min(s1, tp[3])
                if x1 > x0:
                    self._adam(st, group, x0, x1, done, scale)
                continue
            if seg[0] == "dense":
                g, p, n = a.grad[s0:s1], a.param[s0:s1], s1 - s0
                if opt.sharded:
                    chunk, bulk, l, h = D.shard_bounds(n, W, self.rank)
                    if chunk > 0:
                        self._py(self._comm, st, D.shard_reduce, g, l, h, bulk)
                        self._adam(st, group, s0 + l, s0 + h, (), scale)
                        if l > 0:
                            self._py(self._zero, st, g[:l])
                        if h < bulk:
                            self._py(self._zero, st, g[h:bulk])
                        self._py(self._comm, st, D.shard_gather, p, l, h, bulk)
                    if bulk < n:
                        self._py(self._comm, st, D._all_reduce, g[bulk:])
                        self._adam(st, group, s0 + bulk, s1, (), scale)
                else:
                    self._py(self._comm, st, D._all_reduce, g)
                    self._adam(st, group, s0, s1, (), scale)
            else:
                F = seg[4]
                i0, i1 = opt._cut(group, seg, s0, s1)
                if i1 > i0:
                    idx = torch.div(seg[3][i0:i1].long(), F, rounding_mode="floor")
                    self._keep.append(idx)
                    self._py(self._comm, st, D.exchange_rows, a.grad.view(-1, F), idx)
                    self._adam(st, group, s0, s1, (), scale)

    def _adam(self, st, group: str, lo: int, hi: int, done, scale: float = 1.0) -> None:
        a, oc = self.opt.arenas[group], self.opt.config[group]["optimizer"]
        b1, b2, eps = float(oc.betas[0]), float(oc.betas[1]), float(oc.eps)
        for piece in self.opt.adam_pieces(group, lo, hi, done):
            if piece[0] == "dense":
                x0, x1 = piece[1], piece[2]
                self._kg(st, group, "snf_adam_step", a.param[x0:x1], a.grad[x0:x1], a.exp_avg[x0:x1], a.exp_avg_sq[x0:x1], x1 - x0,
                        0.0, b1, b2, eps, 1, float(scale), 1, units=32.0 * (x1 - x0), dyn={("lr", group): 5, ("t", group): 9})
            else:
                rows, F = piece[1], piece[2]
                self._keep.append(rows)
                self._kg(st, group, "snf_adam_step_rows", a.param, a.grad, a.exp_avg, a.exp_avg_sq, rows, rows.numel(), int(F), 0.0,
                        b1, b2, eps, 1, float(scale), 1, units=32.0 * rows.numel() * F, dyn={("lr", group): 7, ("t", group): 11})

    def _sparse_lists(self, enc, N: int, n_sparse: int):
        """(levels, rows, list offsets, longest list) of a table's reachable-row levels for a backward over N samples, or None
        when the table has none / the switch is off / a bucket's list is too long for the compact reduce."""
        F, T = enc.n_features_per_level, enc.log2_hashmap_size
        if not (SPARSE_LEVELS and n_sparse > 0 and self.opt.skip_unreachable_rows and (F == 8 or SPARSE_LEVELS_F2)):
            return None
        log2B = int(self.lib.snf_hashgrid_bucket_bits(int(N), int(T)))
        ns, rows, start, longest = enc.reach_lists(log2B, int(self.lib.snf_hashgrid_sparse_max_rows(F)))
        if ns == 0:
            return None
        assert ns <= n_sparse
        self._keep.extend((rows, start))
        return ns, rows, start, longest

    def _sort_ws(self, name: str, N: int, L: int, T: int, parity: Optional[int] = None) -> Tuple[torch.Tensor, int]:
        nbytes = int(self.lib.snf_hashgrid_bwd_workspace_bytes(N, L, T))
        return self.buf(name, ((nbytes + 3) // 4,), torch.int32, parity), nbytes

    def _chain_recompute(self) -> bool:
        return CHAIN_RECOMPUTE and int(self.lib.snf_get_gemm_mode()) == 1

    def _mlp64_bwd(self, st, x, ldx, in_real, ws, out_act, y, h1, h2, dy, lddy, dy_off, dy0, dx, lddx, N, pre: str):
        """ops._mlp64_bwd_launch: data-gradient chain + the three (two) weight-gradient GEMMs into the gradient arena."""
        nh = len(ws) - 1
        out = ws[-1].shape[0]
        tag = f"{in_real}x{'x'.join(['64'] * nh)}x{out}"
        if FUSED_CHAIN_WGRAD and int(self.lib.snf_get_gemm_mode()) >= 1:
            # the chain forms its own weight gradients (bf16x3, per-wave LDS transposes): dH1 / dH2 / dZ are never written
            nb = int(self.lib.snf_mlp64_bwd_fused_workspace_bytes(nh))
            wsb = self.buf(pre + "wgrad_ws", (nb // 4,))
            self._k(st, "snf_mlp64_bwd_fused", dy, lddy, dy_off, dy0, y, out, x, ldx, ws[0], in_real, ws[1] if nh == 2 else None,
                    ws[-1], nh, out, out_act, N, h1, h2, dx, lddx, ws[0].main_grad, ws[1].main_grad if nh == 2 else None,
                    ws[-1].main_grad, wsb, nb, tag=tag)
            return
        ldz = (out + 3) // 4 * 4
        dh1 = self.buf(pre + "dh1", (N, 64))
        dh2 = self.buf(pre + "dh2", (N, 64)) if nh == 2 else None
        dz = self.buf(pre + "dz", (N, ldz))
        self._k(st, "snf_mlp64_bwd_data", dy, lddy, dy_off, dy0, y, out, ws[0], in_real, ws[1] if nh == 2 else None, ws[-1],
                nh, out, out_act, N, h1, h2, dh1, dh2, dz, ldz, dx, lddx, tag=tag)
        pairs = [(dh1, 64, x, ldx, in_real, ws[0])]
        if nh == 2:
            pairs.append((dh2, 64, h1, 64, 64, ws[1]))
        pairs.append((dz, ldz, h2 if nh == 2 else h1, 64, 64, ws[-1]))
        for g, ldg, a, lda, I, w in pairs:
            O = w.shape[0]
            self._k(st, "snf_linear_bwd_weight", g, None, a, N, I, O, ldg, 0, lda, ops.ACT_NONE, w.main_grad, None,
                    tag=f"{I}x{O}")

    # ------------------------------------------------------------------------------------------------------------
    def _side_streams(self, overlap: bool):
        """(head streams, stream of the forward-time sorts, stream of the feature sorts, stream of the step's prologue)."""
        main = self.main
        side = {h: (self.tr._side[h] if overlap else main) for h in self.heads}
        sort_st = feat_sort_st = main
        if overlap and ops.PRESORT_SIDE_STREAM:
            pref = self.tr.presort_host
            if pref == "auto":
                pref = "clipseg" if "clipseg" in side else ("sam" if "sam" in side else "own")
            if pref in side:
                sort_st = side[pref]
            elif pref == "main":
                sort_st = main
            elif pref == "wgrad":  # the heads' shared weight-gradient stream: idle while the forward runs
                sort_st = ops.make_stream("wgrad")
            else:
                if self._own_sort_stream is None:
                    self._own_sort_stream = ops.make_stream("presort")
                sort_st = self._own_sort_stream
            feat_sort_st = side.get("clipseg", main) if FEATURE_SORTS_ON_HEAD_STREAM else main
        xstep = XSTEP_PROLOGUE and not self.heads and not self.multi and sort_st.stream_id != main.stream_id
        return side, sort_st, feat_sort_st, (sort_st if xstep else main)

    def _build(self, parity: int, updated: bool, with_opt: bool, overlap: bool, prop_adam_when_idle: bool) -> _Plan:
        plan = self._plan = _Plan()
        self._overlap = bool(overlap)
        self._keep: list = getattr(self, "_keep", [])
        model, cfg, opt = self.model, self.cfg, self.opt
        R, P, S, K = self.R, self.P, self.S, self.K
        N0, N1, NK = R * P, R * S, R * K
        main = self.main
        # The backward sorts depend on positions only; they run beside the forward on the LEAST loaded stream.  With the steps
        # pipelined (no join at the end of a step) the SAM stream is the busiest of the three (two 256-wide layers, the conv
        # head, 0.4 GB of table Adam): event timeline of r02d: 3.8 / 3.6 / 2.4 ms busy per 4.06 ms step for sam / main /
        # clipseg with the sorts on the sam stream -- so they ride on the clipseg stream (Trainer.presort_host overrides); a run
        # without feature heads gets a stream of its own for them.
        side, sort_st, feat_sort_st, pre = self._side_streams(overlap)
        xstep = pre.stream_id != main.stream_id  # the step's prologue runs on the side stream, under the previous step's tail
        pp = parity if xstep else None
        b = self.buf

        # ---- inputs (filled by `_load_inputs` before the replay) and constants
        o, d = b("in_o", (R, 3), parity=pp), b("in_d", (R, 3), parity=pp)
        image = b("in_image", (R, 3), parity=pp)
        t_rand, u_rand = b("in_t_rand", (R,), parity=pp), b("in_u_rand", (R,), parity=pp)
        if "nears" not in self.bufs:
            b("nears", (R,)).fill_(float(model.collider.near_plane))
            b("fars", (R,)).fill_(float(model.collider.far_plane))
            b("one", (1,)).fill_(1.0)
        nears, fars, one = b("nears", (R,)), b("fars", (R,)), b("one", (1,))

        # ================= main stream: proposal sampler (ray_samplers.py:549-599) =================
        if xstep:  # last step's nerf losses (the last main-stream readers of what this block overwrites) are through
            self._py(pre.wait_event, self.event("losses_done"))
        prop = model.proposal_networks[0]
        penc, pnet = prop.mlp_base.encoding, prop.mlp_base.network
        pw0, pw1 = pnet.weights()
        PL, PF, PT = penc.n_levels, penc.n_features_per_level, penc.log2_hashmap_size
        sb0, eb0 = b("sb0", (R, P + 1)), b("eb0", (R, P + 1))
        self._k(pre, "snf_sample_spacing", nears, fars, t_rand, R, P, sb0, eb0)
        u0, sel0 = b("u0", (N0, 3)), b("sel0", (N0,), torch.uint8)
        self._k(pre, "snf_positions", o, d, eb0, None, R, P, P, ops.CONTRACT_LINF, 1, u0, sel0)
        if updated:
            ws_p, ws_p_bytes = self._sort_ws("ws_prop", N0, PL, PT)
            self._edge(pre, sort_st, "u0_ready")
            self._k(sort_st, self._f2_sort(PF, N0, PL, PT), u0, penc.scalings, N0, PL, PT, ws_p, ws_p_bytes, tag=f"L{PL}")
            if sort_st.stream_id != main.stream_id:
                self._py(self.event("prop_sorted").record, sort_st)
        I0, H0 = pnet.n_input_dims, pw0.shape[0]
        dens0 = b("dens0", (N0,))
        if (not updated) and FUSED_PROP_FWD and PL == 5 and PF == 2 and H0 == 16 and I0 == 10 and pw1.shape[0] == 1:
            # a step whose proposal network receives no gradient (ray_samplers.py:586-591: the `updated` gate) needs neither the
            # [N, 10] encoding nor the hidden activations: grid + density net + trunc_exp per sample in registers, one launch
            # (snf_prop_density_fwd, the eval render's proposal stage: bit-identical densities)
            self._k(pre, "snf_prop_density_fwd", u0, penc.params, penc.scalings, N0, PL, PF, PT, pw0, pw1, H0, sel0, dens0,
                    tag=f"F{PF}L{PL}")
        else:
            enc0 = b("enc0", (N0, PL * PF))
            self._k(pre, "snf_hashgrid_fwd", u0, penc.params, penc.scalings, N0, PL, PF, PT, enc0, PL * PF, 0, tag=f"F{PF}L{PL}")
            hid0 = b("hid0", (N0, H0)) if updated else None
            raw0 = b("raw0", (N0, 1))
            self._k(pre, "snf_mlp_tiny_fwd", enc0, I0, pw0, pw1, I0, H0, N0, hid0, raw0, tag=f"{I0}x{H0}x1")
            self._k(pre, "snf_trunc_exp_fwd", raw0, 1, sel0, N0, dens0)
        w0 = b("w0", (R, P))
        self._k(pre, "snf_weights_fwd", dens0, 1, 1, None, eb0, R, P, w0, None)
        sb1, eb1 = b("sb1", (R, S + 1), parity=pp), b("eb1", (R, S + 1), parity=pp)
        self._k(pre, "snf_pdf_resample", w0, sb0, u_rand, nears, fars, R, P, S, 1.0,
                float(model.proposal_sampler.pdf_sampler.histogram_padding), sb1, eb1, dyn={("anneal",): 8})

        # ================= main stream: nerfacto field (ops._NerfactoField) =================
        fenc, fbase, fhead = model.field.mlp_base.encoding, model.field.mlp_base.network, model.field.mlp_head
        bw0, bw1 = fbase.weights()
        hw0, hw1, hw2 = fhead.weights()
        FL, FF, FT = fenc.n_levels, fenc.n_features_per_level, fenc.log2_hashmap_size
        u1, sel1 = b("u1", (N1, 3), parity=pp), b("sel1", (N1,), torch.uint8, parity=pp)
        self._k(pre, "snf_positions", o, d, eb1, None, R, S, S, ops.CONTRACT_LINF, 1, u1, sel1)
        ws_f, ws_f_bytes = self._sort_ws("ws_field", N1, FL, FT, pp)
        self._edge(pre, sort_st, "u1_ready")
        self._edge(pre, main, "prologue_done")
        self._k(sort_st, self._f2_sort(FF, N1, FL, FT), u1, fenc.scalings, N1, FL, FT, ws_f, ws_f_bytes, tag=f"L{FL}")
        if sort_st.stream_id != main.stream_id:
            self._py(self.event("field_sorted").record, sort_st)
        enc1 = b("enc1", (FL * FF * N1,))
        self._k(main, "snf_hashgrid_fwd", u1, fenc.params, fenc.scalings, N1, FL, FF, FT, enc1, 0, 0, tag=f"F{FF}L{FL}")
        C = bw1.shape[0]  # 1 + geo
        rc = self._chain_recompute()  # the hidden activations of the two field nets are not stored
        hb1, h = (None if rc else b("hb1", (N1, 64))), b("h", (N1, C))
        density1 = b("density1", (N1,))
        # (trunc_exp of output 0 from the chain's epilogue where the base net stores nothing but its 16 outputs; else two kernels)
        if FUSED_DENSITY:
            self._k(main, "snf_mlp64_fwd_density", enc1, 0, bw0, FL * FF, None, bw1, 1, C, ops.ACT_NONE, N1, hb1, None, h, C, sel1,
                    density1, tag=f"{FL * FF}x64x{C}")
        else:
            self._k(main, "snf_mlp64_fwd", enc1, 0, bw0, FL * FF, None, bw1, 1, C, ops.ACT_NONE, N1, hb1, None, h, C,
                    tag=f"{FL * FF}x64x{C}")
            self._k(main, "snf_trunc_exp_fwd", h, C, sel1, N1, density1)
        n_geo = C - 1
        sh_in = bool(FUSED_SH_INPUT and rc and FUSED_CHAIN_WGRAD and n_geo == 15 and C % 4 == 0)
        hh1, hh2, rgb = (None if rc else b("hh1", (N1, 64))), (None if rc else b("hh2", (N1, 64))), b("rgb", (N1, 3))
        if sh_in:
            x2 = None
            self._k(main, "snf_mlp64_fwd_sh", d, R, S, h, C, n_geo, hw0, hw1, hw2, 2, 3, ops.ACT_SIGMOID, None, None, rgb, 3,
                    tag=f"{16 + n_geo}x64x64x3")
        else:
            x2 = b("x2", (N1, 32))
            self._k(main, "snf_head_input", d, self._off(h, 4), R, S, n_geo, C, x2, 32)
            self._k(main, "snf_mlp64_fwd", x2, 32, hw0, 16 + n_geo, hw1, hw2, 2, 3, ops.ACT_SIGMOID, N1, hh1, hh2, rgb, 3,
                    tag=f"{16 + n_geo}x64x64x3")
        w1 = b("w1", (R, S))
        self._k(main, "snf_weights_fwd", density1, 1, 1, None, eb1, R, S, w1, None)
        out_rgb, depth, acc, pdepth = b("out_rgb", (R, 3)), b("out_depth", (R, 1)), b("out_acc", (R, 1)), b("out_pdepth", (R, 1))
        self._k(main, "snf_composite_fwd", rgb, w1, None, R, S, 1, out_rgb, None, None)
        self._k(main, "snf_composite_fwd", None, w1, eb1, R, S, 1, None, acc, depth)
        self._k(main, "snf_composite_fwd", None, w0, eb0, R, P, 1, None, None, pdepth)

        # ================= feature-sample selection, shared by the heads (sam_model.py:243-255) =================
        if self.heads:
            sf = model.sam_field
            ids, wk = b("ids", (R, K), torch.int32), b("wk", (R, K), parity=parity)
            self._k(main, "snf_topk_sharpen", w1, R, S, K, float(cfg.sharpening_temperature), ids, wk)
            uk = b("uk", (NK, 3), parity=parity)
            self._k(main, "snf_positions", o, d, eb1, ids, R, S, K, ops.CONTRACT_L2, 0, uk, None)
            # table-parallel grids: every rank's top-K positions, gathered once for both heads (rank order)
            self._tp_layout = ops.table_parallel_layout(tuple(e.spec for e in sf.clip_encs)) if self.multi else None
            uk_all = None
            if self._tp_layout is not None:
                uk_all = b("uk_all", (self.world * NK, 3), parity=parity)
                self._py(self._collective, main, "all_gather", uk_all, uk)
            for hname in self.heads:  # the heads' forward needs the selected samples only
                self._edge(main, side[hname], f"selected_{hname}")
            # the SAM and ClipSeg grids share the two level geometries: one sort per geometry, needed by the heads' BACKWARD
            # only -- off the main stream's chain (it sits between the nerfacto forward and its backward there)
            geo_ws = {}
            if self._tp_layout is not None:
                # (the heads share the ownership layout as well: one sort per owned level run, over the gathered positions)
                for gi, l0, nl, _ in self._tp_layout.runs(self.rank):
                    enc = sf.clip_encs[gi]
                    T = enc.log2_hashmap_size
                    ws, nb = self._sort_ws(f"ws_feat_tp{gi}_{l0}_{nl}", self.world * NK, nl, T, parity)
                    self._k(feat_sort_st, "snf_hashgrid_sort", uk_all, ops._sc_run(enc.scalings, l0, nl), self.world * NK, nl, T,
                            ws, nb, tag=f"L{nl}tp")
                    geo_ws[(gi, l0, nl)] = ws
            else:
                for enc in sf.clip_encs:
                    key = ops._geometry_key(enc.scalings, enc.n_levels, enc.log2_hashmap_size)
                    if key not in geo_ws:
                        ws, nb = self._sort_ws(f"ws_feat{len(geo_ws)}", NK, enc.n_levels, enc.log2_hashmap_size, parity)
                        self._k(feat_sort_st, "snf_hashgrid_sort", uk, enc.scalings, NK, enc.n_levels, enc.log2_hashmap_size, ws,
                                nb, tag=f"L{enc.n_levels}")
                        geo_ws[key] = ws
            self._feat_sorted = None
            if any(side[h].stream_id != feat_sort_st.stream_id for h in self.heads):
                self._feat_sorted = self.event(f"feat_sorted_{parity}")
                self._py(self._feat_sorted.record, feat_sort_st)
            self._feat_sort_stream_id = feat_sort_st.stream_id

        # ================= nerf losses, forward (nerfacto.py:316-344) =================
        mse_acc = b("mse_acc_rgb", (516,), zero=True)
        mse_out = b("mse_out_rgb", (2,))
        RC = R * 3 // 64
        self._k(main, "snf_rowmse_loss_fwd", out_rgb, image, RC, 64, 1.0, 0, mse_acc, mse_out)
        rows_d, gw_d = b("rows_d", (R,)), b("gw_d", (R, S))
        self._k(main, "snf_distortion", sb1, w1, R, S, 1.0 / float(R), rows_d, gw_d)
        rows_i = b("rows_i", (R,))
        gwp = b("gwp", (R, P)) if updated else None
        # (the loss rows stay unscaled -- the summary applies the multiplier; the gradient carries it, as autograd's does)
        self._k(main, "snf_interlevel", sb1, w1, sb0, w0, R, S, P, float(cfg.interlevel_loss_mult) / float(R * S), rows_i, gwp)
        summary = b("loss_summary", (8,))
        self._k(main, "snf_nerf_loss_summary", mse_out, rows_i, float(cfg.interlevel_loss_mult) / float(R * S), rows_d,
                1.0 / float(R), float(cfg.distortion_loss_mult), R, summary)
        self._guard_update(main, "nerf", summary)
        if self._guard("nerf") is not None and pre.stream_id != main.stream_id:  # (an Adam launch of this step on the prologue's stream reads the verdict)
            self._py(self.event("guard_nerf_set").record, main)
        if xstep:
            self._py(self.event("losses_done").record, main)
        self._losses_mark = len(plan.entries)  # the proposal backward's inputs exist from here on

        # ================= feature heads: one task per head on its own stream =================
        for hname in self.heads:
            st = side[hname]
            self._head_domain = hname
            self._head_task(st, hname, parity, with_opt, geo_ws)
            self._head_domain = None
            if st.stream_id != main.stream_id:
                self._py(self._mark_head_busy, st, parity, hname)

        # ================= main stream: backward of the nerf losses =================
        d_rgb = b("d_out_rgb", (R, 3))
        self._k(main, "snf_rowmse_loss_bwd", out_rgb, image, RC, 64, 1.0, 0, one, mse_out, d_rgb)
        grgb, gw = b("grgb", (N1, 3)), b("gw", (R, S))
        self._k(main, "snf_composite_bwd", rgb, w1, d_rgb, R, S, grgb, gw)
        self._k(main, "snf_add_scaled", N1, float(cfg.distortion_loss_mult), gw_d, gw)
        gd1 = b("gd1", (N1,))
        self._k(main, "snf_weights_bwd", density1, 1, 1, None, eb1, gw, R, S, gd1)
        if sh_in:
            # the colour net's backward forms its input row again (harmonics + geo columns of h) and writes d(geo) only, [N, 16]:
            # the base net reads it as the gradient of its output columns 1 .. 15 (dy_col_off = -1; column 0 comes from graw)
            dgeo = b("dgeo", (N1, 16))
            nbw = int(self.lib.snf_mlp64_bwd_fused_workspace_bytes(2))
            wsb = self.buf("hd_wgrad_ws", (nbw // 4,))
            self._k(main, "snf_mlp64_bwd_fused_sh", grgb, 3, rgb, 3, d, R, S, h, C, n_geo, hw0, hw1, hw2, 2, 3, ops.ACT_SIGMOID, dgeo,
                    16, hw0.main_grad, hw1.main_grad, hw2.main_grad, wsb, nbw, tag=f"{16 + n_geo}x64x64x3")
            dyb, lddyb, offb = dgeo, 16, -1
        else:
            dx2 = b("dx2", (N1, 32))
            self._mlp64_bwd(main, x2, 32, 16 + n_geo, (hw0, hw1, hw2), ops.ACT_SIGMOID, rgb, hh1, hh2, grgb, 3, 0, None, dx2, 32,
                            N1, "hd_")
            dyb, lddyb, offb = dx2, 32, 15
        graw = b("graw", (N1,))
        self._k(main, "snf_trunc_exp_bwd", h, C, sel1, gd1, N1, graw, 1)
        denc1 = b("denc1", (FL * FF * N1,))
        self._mlp64_bwd(main, enc1, 0, FL * FF, (bw0, bw1), ops.ACT_NONE, None, hb1, None, dyb, lddyb, offb, graw, denc1, 0, N1,
                        "bs_")
        if sort_st.stream_id != main.stream_id:
            self._py(main.wait_event, self.event("field_sorted"))
        done_f: list = []
        fuse_local = with_opt and not self.multi  # across ranks the gradient mean comes first (exchange_and_step)
        self._grid_bwd(main, denc1, N1, fenc, "fields", 0, 0, ws_f, None, fuse_local, done_f)
        done_p: list = []
        # The proposal network's backward (interlevel loss -> weights -> tiny MLP -> its hash grid) shares nothing with the
        # field's backward but the forward results.  Without feature heads the GPU is otherwise on ONE stream during the
        # backward, so that chain goes to the (then idle) sort stream and runs beside the field backward; with heads the three
        # streams already saturate the chip and it stays on the main stream (step_program.PROP_BWD_SIDE overrides).
        # (with heads and the shared weight-gradient stream: there -- 2.553 2.549 2.569 2.541 -> 2.532 2.534 2.537 2.525 ms, same box)
        side_prop = True if xstep else (PROP_BWD_SIDE if PROP_BWD_SIDE is not None else (2 if (self.heads and WGRAD_STREAM) else (not self.heads)))
        use_wg = bool(side_prop == 2 and WGRAD_STREAM and self.heads and not self.multi and overlap)
        if side_prop == 2 and not use_wg:
            side_prop = False  # (no shared weight-gradient stream in this schedule: with heads the chain stays on the main stream)
        prop_st = sort_st if (updated and side_prop and sort_st.stream_id != main.stream_id) else main
        if updated and use_wg:
            prop_st = ops.make_stream("wgrad")
        prop_block = []
        if updated:
            mark = len(plan.entries)
            gd0 = b("gd0", (N0,))
            self._k(prop_st, "snf_weights_bwd", dens0, 1, 1, None, eb0, gwp, R, P, gd0)
            graw0 = b("graw0", (N0, 1))
            self._k(prop_st, "snf_trunc_exp_bwd", raw0, 1, sel0, gd0, N0, graw0, 1)
            denc0 = b("denc0", (N0, I0))
            self._k(prop_st, "snf_mlp_tiny_bwd", graw0, enc0, I0, hid0, pw0, pw1, I0, H0, N0, denc0, I0, pw0.main_grad,
                    pw1.main_grad, tag=f"{I0}x{H0}x1")
            if sort_st.stream_id != prop_st.stream_id:
                self._py(prop_st.wait_event, self.event("prop_sorted"))
            stage0 = b("stage_prop", (PL * N0 * PF,))
            self._grid_bwd(prop_st, denc0, N0, penc, "proposal_networks", PL * PF, 0, ws_p, stage0, fuse_local, done_p)
            if with_opt and prop_st.stream_id != main.stream_id:
                self._opt_step(prop_st, "proposal_networks", 0, opt.arenas["proposal_networks"].numel, done_p)
            if prop_st.stream_id != main.stream_id:
                # issue the side chain right after the nerf losses (its inputs), i.e. before the field backward on the host too
                prop_block = plan.entries[mark:]
                del plan.entries[mark:]
        if with_opt:
            self._opt_step(main, "fields", 0, opt.arenas["fields"].numel, done_f)
            if (updated and prop_st.stream_id == main.stream_id) or (not updated and prop_adam_when_idle):
                # (a step without a proposal backward: zero gradients, the moments decay.  With the prologue on the side stream
                # it goes there too -- the next prologue reads these parameters)
                if self._guard("nerf") is not None and pre.stream_id != main.stream_id:
                    self._py(pre.wait_event, self.event("guard_nerf_set"))
                self._opt_step(pre, "proposal_networks", 0, opt.arenas["proposal_networks"].numel, done_p)
        if prop_block:
            ev_in, ev_out = self.event("prop_bwd_inputs"), self.event("prop_bwd_done")
            at = self._losses_mark
            block = [[_PY, ev_in.record, [main], None, 0.0, None], [_PY, prop_st.wait_event, [ev_in], None, 0.0, None]] + prop_block \
                    + [[_PY, ev_out.record, [prop_st], None, 0.0, None]]
            plan.entries[at:at] = block
            if not xstep:
                self._py(main.wait_event, ev_out)  # the next step's proposal forward reads the stepped parameters
        self._plan = None
        return plan

    # ------------------------------------------------------------------------------------------------------------
    def _linear_fwd_ws(self, st, x, w, bias, N, I, O, act, y, tag, name):
        nbytes = int(self.lib.snf_linear_fwd_workspace_bytes(N, I, O))
        ws = self.buf(name, (max(nbytes, 16) // 4,))
        self._k(st, "snf_linear_fwd_ws", x, w, bias, N, I, O, I, O, act, y, ws, nbytes, tag=tag)

    def _head_task(self, st, hname: str, parity: int, with_opt: bool, geo_ws: dict) -> None:
        """One feature head, forward to optimizer (sam_model.py:243-277,316-328; sam_field.py:112-140)."""
        model, cfg, opt = self.model, self.cfg, self.opt
        R, K = self.R, self.K
        NK = R * K
        b = self.buf
        sf = model.sam_field
        # the stream of this head's weight-gradient launches and of the Adam launches behind them (WGRAD_STREAM; else the head's own)
        wst = st
        if WGRAD_STREAM and not self.multi and st.stream_id != self.main.stream_id:
            wst = ops.make_stream("wgrad")
        self._wg_n = getattr(self, "_wg_n", 0)

        def to_w(tag_: str) -> None:  # what the head's stream has enqueued so far is input of the next launch on the wgrad stream
            if wst is not st:
                self._wg_n += 1
                self._edge(st, wst, f"wg_in_{hname}_{parity}_{self._wg_n}_{tag_}")
        encs = list(sf.clip_encs if hname == "sam" else sf.clipseg_encs)
        net = sf.sam_net if hname == "sam" else sf.clipseg_net
        ws_ = net.weights()
        uk, wk = b("uk", (NK, 3), parity=parity), b("wk", (R, K), parity=parity)
        total = sum(e.n_output_dims for e in encs)
        # The encoding travels LEVEL-MAJOR ([total/8][NK][8], ld = -8) between the grids and the head's first layer: the
        # level-at-a-time grid kernels then store whole lines (a row-major [NK,192] gets 32 bytes per 768-byte row from each
        # level: 2x write amplification), the first layer's A fragments are 1 KB contiguous per half-wave, its data gradient
        # is written straight into the staged-gradient layout of the table backward (no staging pass) and its weight
        # gradient reads the same buffer.  Needs the bf16x3 GEMM kernels (gemm mode >= 1).
        planar = (PLANAR_HEADS and int(self.lib.snf_get_gemm_mode()) >= 1 and all(e.n_features_per_level == 8 for e in encs)
                  and total % 16 == 0 and 64 <= total <= 256)
        ld_enc = -8 if planar else total
        enc_out = b(f"{hname}_enc", (NK * total,) if planar else (NK, total))
        layout = ops.table_parallel_layout(tuple(e.spec for e in encs)) if self.multi else None
        W = self.world
        if layout is not None:
            # Table-parallel: this rank evaluates its level runs at the samples of EVERY rank, one launch per (run, destination)
            # writing level-major into the send buffer [destination][own level][sample][8]; after the all-to-all the blocks sit in
            # owner = level order, i.e. `enc_out` is the level-major encoding [total/8][NK][8] the first layer reads.
            assert planar and self._tp_layout is not None and layout.runs(self.rank) == self._tp_layout.runs(self.rank)
            per = layout.per
            uk_all = b("uk_all", (W * NK, 3), parity=parity)
            send = b(f"{hname}_tp_send", (W * per * NK * 8,))
            for gi, l0, nl, col in layout.runs(self.rank):
                e = encs[gi]
                F, T = e.n_features_per_level, e.log2_hashmap_size
                slab = self._off(e.params, ((l0 << T) * F) * 4)
                for w in range(W):
                    self._k(st, "snf_hashgrid_fwd", self._off(uk_all, w * NK * 12), slab, ops._sc_run(e.scalings, l0, nl), NK, nl, F, T,
                            self._off(send, ((w * per + col // F) * NK * 8) * 4), 0, 0, tag=f"F{F}L{nl}tp")
                if not any(e.params is q for q in self._tp_params):
                    self._tp_params.append(e.params)
            self._py(self._collective, st, "all_to_all", enc_out, send)
        else:
            col = 0
            for e in encs:
                L, F, T = e.n_levels, e.n_features_per_level, e.log2_hashmap_size
                if planar:
                    self._k(st, "snf_hashgrid_fwd", uk, e.params, e.scalings, NK, L, F, T, self._off(enc_out, col * NK * 4), 0, 0,
                            tag=f"F{F}L{L}")
                else:
                    self._k(st, "snf_hashgrid_fwd", uk, e.params, e.scalings, NK, L, F, T, enc_out, total, col, tag=f"F{F}L{L}")
                col += L * F
        # the head MLP (tcnn CutlassMLP role): ReLU between layers, no output activation
        acts, x = [enc_out], enc_out
        n_lay = len(ws_)
        commute = (MEAN_BEFORE_LAST_LAYER and n_lay >= 2 and net.output_activation == ops.ACT_NONE and ws_[-1].shape[1] % 4 == 0)
        # ... then the last hidden layer's output is only rendered and differentiated through its ReLU: the weight-stationary /
        # full-width bf16-split kernels form the broadcast gradient in their loaders (rows_ok) and, for K = 16, render in the
        # forward GEMM's epilogue and keep the ReLU mask as bits instead of the activations (fuse_mean)
        rows_ok = fuse_mean = False
        if commute:
            wp = ws_[n_lay - 2]
            rows_ok = (ROWS_OPERAND and int(self.lib.snf_get_gemm_mode()) >= 1 and NK >= 8192 and
                       int(self.lib.snf_linear_bwd_weight_workspace_bytes(NK, wp.shape[1], wp.shape[0])) > 0 and
                       64 <= wp.shape[0] <= 256 and wp.shape[0] % 16 == 0 and 64 <= wp.shape[1] <= 256 and wp.shape[1] % 16 == 0)
            fuse_mean = rows_ok and FUSED_MEAN_EPILOGUE and K == 16 and wp.shape[0] % 32 == 0
        for i, w in enumerate(ws_[:n_lay - 1] if commute else ws_):
            O, I = w.shape
            act = ops.ACT_RELU if i < n_lay - 1 else net.output_activation
            if fuse_mean and i == n_lay - 2:
                hbar = b(f"{hname}_hbar", (R, O))
                ymask = b(f"{hname}_mask{i}", (NK, O // 8), torch.uint8)
                self._k(st, "snf_linear_fwd_mean", x, w, NK, I, O, ld_enc if i == 0 else I, wk, K, hbar, ymask, None, O,
                        tag=f"{I}x{O}")
                acts.append(ymask)
                x = None
                continue
            y = b(f"{hname}_a{i}", (NK, O))
            self._k(st, "snf_linear_fwd", x, w, None, NK, I, O, ld_enc if i == 0 else I, O, act, y, tag=f"{I}x{O}")
            acts.append(y)
            x = y
        if commute:
            w_last = ws_[-1]
            Cf, Ih = w_last.shape
            if not fuse_mean:
                hbar = b(f"{hname}_hbar", (R, Ih))
                self._k(st, "snf_feature_mean_fwd", x, wk, R, K, Ih, hbar)
            fm = b(f"{hname}_fm", (R, Cf))
            self._k(st, "snf_linear_fwd", hbar, w_last, None, R, Ih, Cf, Ih, Cf, ops.ACT_NONE, fm, tag=f"{Ih}x{Cf}r")
        else:
            Cf = x.shape[1]
            fm = b(f"{hname}_fm", (R, Cf))
            self._k(st, "snf_feature_mean_fwd", x, wk, R, K, Cf, fm)
        conv = hname == "sam" and cfg.patch_size > 1
        if conv:
            c0, c1 = model.conv_head[0], model.conv_head[2]
            p, k = cfg.patch_size, c0.weight.shape[-1]
            kk = k * k
            O0, O1 = c0.weight.shape[0], c1.weight.shape[0]
            npatch = R // (p * p)
            colb = b("cv_col", (R, Cf * kk))
            self._k(st, "snf_patch_unfold", fm, R, p, Cf, k, colb)
            hc = b("cv_h", (R, O0))
            self._linear_fwd_ws(st, colb, c0.weight, c0.bias, R, Cf * kk, O0, ops.ACT_RELU, hc, f"{Cf * kk}x{O0}", "cv_ws0")
            cm = b("cv_cm", (npatch, O0 * kk))
            self._k(st, "snf_patch_unfold_mean", hc, R, p, O0, k, cm)
            pred = b("cv_y", (npatch, O1))
            self._linear_fwd_ws(st, cm, c1.weight, c1.bias, npatch, O0 * kk, O1, ops.ACT_NONE, pred, f"{O0 * kk}x{O1}pm",
                                "cv_ws1")
            rows, Cp = npatch, O1
        else:
            pred, rows, Cp = fm, R, Cf
        target = b(f"in_{hname}", (rows, Cp), parity=parity)
        weight = float(cfg.sam_loss_weight if hname == "sam" else cfg.clipseg_loss_weight)
        acc, out = b(f"mse_acc_{hname}", (516,), zero=True), b(f"mse_out_{hname}", (2,))
        self._k(st, "snf_rowmse_loss_fwd", pred, target, rows, Cp, weight, 1, acc, out)
        self._guard_update(st, hname, out)
        # ---- backward
        one = b("one", (1,))
        dpred = b(f"{hname}_dpred", (rows, Cp))
        self._k(st, "snf_rowmse_loss_bwd", pred, target, rows, Cp, weight, 1, one, out, dpred)
        if conv:
            w0, b0, w1, b1 = c0.weight, c0.bias, c1.weight, c1.bias
            to_w("pm")
            self._k(wst, "snf_linear_bwd_weight", dpred, None, cm, npatch, O0 * kk, O1, O1, O1, O0 * kk, ops.ACT_NONE,
                    w1.main_grad, None if b1 is None else b1.main_grad, tag=f"{O0 * kk}x{O1}pm")
            dcm = b("cv_dcm", (npatch, O0 * kk))
            self._k(st, "snf_linear_bwd_data", dpred, None, w1, npatch, O0 * kk, O1, O1, O1, O0 * kk, ops.ACT_NONE, dcm,
                    tag=f"{O0 * kk}x{O1}pm")
            dh = b("cv_dh", (R, O0))
            self._k(st, "snf_patch_fold_mean", dcm, R, p, O0, k, dh)
            to_w("c0")
            self._k(wst, "snf_linear_bwd_weight", dh, hc, colb, R, Cf * kk, O0, O0, O0, Cf * kk, ops.ACT_RELU, w0.main_grad,
                    None if b0 is None else b0.main_grad, tag=f"{Cf * kk}x{O0}")
            dcol = b("cv_dcol", (R, Cf * kk))
            self._k(st, "snf_linear_bwd_data", dh, hc, w0, R, Cf * kk, O0, O0, O0, Cf * kk, ops.ACT_RELU, dcol,
                    tag=f"{Cf * kk}x{O0}")
            dfm = b("cv_dfm", (R, Cf))
            self._k(st, "snf_patch_fold", dcol, R, p, Cf, k, dfm)
        else:
            dfm = dpred
        wgrad_bytes = max(int(self.lib.snf_linear_bwd_weight_workspace_bytes(NK, w.shape[1], w.shape[0])) for w in ws_)
        wgrad_ws = b(f"{hname}_wgrad_ws", (max(wgrad_bytes, 16) // 4,))
        if commute:
            # last layer on the R rendered rows: dW = dfm^T hbar, d(hbar) = dfm W; then every sample's share w_k d(hbar)
            to_w("r")
            self._k(wst, "snf_linear_bwd_weight", dfm, None, hbar, R, Ih, Cf, Cf, Cf, Ih, ops.ACT_NONE, w_last.main_grad, None,
                    tag=f"{Ih}x{Cf}r")
            dhbar = b(f"{hname}_dhbar", (R, Ih))
            self._k(st, "snf_linear_bwd_data", dfm, None, w_last, R, Ih, Cf, Cf, Cf, Ih, ops.ACT_NONE, dhbar, tag=f"{Ih}x{Cf}r")
            # the samples' shares w_k d(hbar): formed by the loaders of the layer below (rows_ok) or written out
            gy = dhbar
            if not rows_ok:
                gy = b(f"{hname}_dfeat", (NK, Ih))
                self._k(st, "snf_feature_mean_bwd", dhbar, wk, R, K, Ih, gy)
        else:
            gy = b(f"{hname}_dfeat", (NK, Cf))
            self._k(st, "snf_feature_mean_bwd", dfm, wk, R, K, Cf, gy)
        for i in range(n_lay - (2 if commute else 1), -1, -1):
            w = ws_[i]
            O, I = w.shape
            act = ops.ACT_RELU if i < n_lay - 1 else net.output_activation
            xin, yout = acts[i], acts[i + 1]
            ldx = ld_enc if i == 0 else I
            gx = b(f"{hname}_dx{i}", (NK * I,) if (i == 0 and planar) else (NK, I))
            nb = int(self.lib.snf_linear_bwd_weight_workspace_bytes(NK, I, O))
            if rows_ok and i == n_lay - 2:
                bits, ldy = (1, O // 8) if fuse_mean else (0, O)
                self._k(st, "snf_linear_bwd_data_rows", gy, wk, K, yout, bits, w, NK, I, O, O, ldy, ldx, act, gx, tag=f"{I}x{O}")
                to_w("rows")
                self._k(wst, "snf_linear_bwd_weight_rows", gy, wk, K, yout, bits, xin, NK, I, O, O, ldy, ldx, act, w.main_grad, wgrad_ws,
                        nb, tag=f"{I}x{O}")
                gy = gx
                continue
            self._k(st, "snf_linear_bwd_data", gy, yout, w, NK, I, O, O, O, ldx, act, gx, tag=f"{I}x{O}")
            to_w(f"l{i}")
            if nb > 0:  # full-width weight gradient: operands read once, partial sums in a scratch buffer
                self._k(wst, "snf_linear_bwd_weight_ws", gy, yout, xin, NK, I, O, O, O, ldx, act, w.main_grad, None, wgrad_ws, nb,
                        tag=f"{I}x{O}")
            else:
                self._k(wst, "snf_linear_bwd_weight", gy, yout, xin, NK, I, O, O, O, ldx, act, w.main_grad, None, tag=f"{I}x{O}")
            gy = gx
        done: list = []
        col = 0
        if self._feat_sorted is not None and st.stream_id != self._feat_sort_stream_id:
            self._py(st.wait_event, self._feat_sorted)
        if layout is not None:
            # the first layer's data gradient [owner][own level][NK][8] goes back to the owners; what arrives is
            # [source rank][own level][NK][8], regrouped to the staged-gradient layout [own level][W*NK][8] of the table backward
            # over the gathered samples (rank order, like uk_all)
            per = layout.per
            grecv = b(f"{hname}_tp_grecv", (W * per * NK * 8,))
            gstage = b(f"{hname}_tp_gstage", (per * W * NK * 8,))
            self._py(self._collective, st, "all_to_all", grecv, gy)
            self._py(self._copy, st, gstage.view(per, W, NK * 8), grecv.view(W, per, NK * 8).transpose(0, 1))
            for gi, l0, nl, col in layout.runs(self.rank):
                e = encs[gi]
                self._grid_bwd(st, self._off(gstage, ((col // 8) * W * NK * 8) * 4), W * NK, e, "sam_field", 0, 0,
                               geo_ws[(gi, l0, nl)], None, with_opt, done, run=(l0, nl), grad_scale=1.0 / W)
        else:
            fuse_local = with_opt and not self.multi
            pair = (PAIR_GRID_BWD and planar and len(encs) == 2 and not FX_F8 and all(e.n_features_per_level == 8 for e in encs)
                    and encs[0].log2_hashmap_size == encs[1].log2_hashmap_size)
            if pair:
                # both grids of the head in one reduce launch (snf_hashgrid_bwd_presorted_adam_pair)
                e0, e1 = encs
                T = e0.log2_hashmap_size
                oc = opt.config["sam_field"]["optimizer"]
                tabs = [self._table_adam(e, "sam_field") for e in encs]
                fuse = fuse_local and opt.fuse_table_adam
                frm = [t[4] if fuse else e.n_levels for t, e in zip(tabs, encs)]
                # reachable-row levels of either table: compact fixed-point reduce in front of the paired launch, stepped there
                # when the optimizer is on -- the launch then steps the whole table
                sps = [self._sparse_lists(e, NK, t[4]) for t, e in zip(tabs, encs)]
                nsp = [sp[0] if sp is not None else 0 for sp in sps]
                longest = max([sp[3] for sp in sps if sp is not None] or [0])
                scratch = self.buf(f"{hname}_sp_scratch", (64,), torch.int32)
                # algorithmic bytes: every corner contribution read once (+ written back as a gradient on the levels left to the
                # caller's optimizer); p / exp_avg / exp_avg_sq read + written for every stepped parameter (dense levels, reachable rows)
                units = 0.0
                for e, t, f, sp in zip(encs, tabs, frm, sps):
                    left = (f - (sp[0] if sp is not None else 0)) if fuse else e.n_levels  # levels whose gradient is written back
                    assert sp is None or sp[0] <= t[4]
                    units += float(NK) * 8 * 8 * 4 * (e.n_levels + left) + 24.0 * (((e.n_levels - f) << T) * 8)
                    if fuse and sp is not None:
                        units += 24.0 * 8 * int(sp[1].numel())
                self._kg(st, "sam_field", "snf_hashgrid_bwd_presorted_adam_pair", gy, self._off(gy, e0.n_levels * 8 * NK * 4), NK, e0.n_levels,
                        e1.n_levels, T, tabs[0][1], tabs[1][1], geo_ws[ops._geometry_key(e0.scalings, e0.n_levels, T)],
                        geo_ws[ops._geometry_key(e1.scalings, e1.n_levels, T)], frm[0], frm[1], tabs[0][0], tabs[0][2], tabs[0][3],
                        tabs[1][0], tabs[1][2], tabs[1][3],
                        sps[0][1] if sps[0] is not None else None, sps[0][2] if sps[0] is not None else None, nsp[0],
                        sps[1][1] if sps[1] is not None else None, sps[1][2] if sps[1] is not None else None, nsp[1], longest,
                        1 if fuse else 0, scratch, 0.0, float(oc.betas[0]), float(oc.betas[1]),
                        float(oc.eps), 1, 1.0, tag=f"F8L{e0.n_levels}+{e1.n_levels}", units=units,
                        dyn={("lr", "sam_field"): 27, ("t", "sam_field"): 31})
                if fuse:
                    for t, e, f, sp in zip(tabs, encs, frm, sps):
                        if sp is not None:  # the reachable rows of the leading sp[0] levels were stepped by the launch
                            t0 = t[5][1] - e.params.numel()
                            done.append((t0, t0 + ((sp[0] << T) * 8)))
                        if f < e.n_levels:
                            done.append(t[5])
            for e in ([] if pair else encs):
                L, F, T = e.n_levels, e.n_features_per_level, e.log2_hashmap_size
                ws_sorted = geo_ws[ops._geometry_key(e.scalings, L, T)]
                if planar:  # the first layer's data gradient IS the staged gradient gT[l][n][F] of this grid's levels
                    self._grid_bwd(st, self._off(gy, col * NK * 4), NK, e, "sam_field", 0, 0, ws_sorted, None, fuse_local, done)
                else:
                    stage = b(f"{hname}_stage", (L * NK * F,))
                    self._grid_bwd(st, gy, NK, e, "sam_field", total, col, ws_sorted, stage, fuse_local, done)
                col += L * F
        if with_opt:
            lo_i, hi_i = self.tr._head_param_ranges()[hname]
            arena = opt.arenas["sam_field"]
            names = list(arena.offsets)
            lo = arena.offsets[names[lo_i]][0]
            hi = arena.offsets[names[hi_i]][0] if hi_i < len(names) else arena.numel
            to_w("adam")  # (the table backward's leftover gradients of this head come from its own stream)
            self._opt_step(wst, "sam_field", lo, hi, done, first=lo_i, last=hi_i, count_step=(hname == self.heads[0]))
            if conv and "conv" in opt.arenas:
                self._opt_step(wst, "conv", 0, opt.arenas["conv"].numel, ())
        if wst is not st:  # the head's stream ends its task behind the wgrad stream: parameters stepped, gradient buffers free again
            self._wg_n += 1
            self._edge(wst, st, f"wg_out_{hname}_{parity}_{self._wg_n}")

    # ------------------------------------------------------------------------------------------------------------
    # run-time pieces referenced by the recorded schedule
    def _wait_head_free(self, main, parity: int, hname: str) -> None:
        ev = self.events.get(f"head_done_{hname}_{parity}")
        if ev is not None and self._head_busy.get((parity, hname)):
            main.wait_event(ev)

    def _mark_head_busy(self, st, parity: int, hname: str) -> None:
        self.event(f"head_done_{hname}_{parity}").record(st)
        self._head_busy[(parity, hname)] = True

    def _target_shape(self, hname: str) -> Tuple[int, int]:
        """(rows, channels) of a head's distillation target: what its prediction has (sam_model.py:259-277,316-328)."""
        sf = self.model.sam_field
        if hname == "sam" and self.cfg.patch_size > 1:
            return self.R // (self.cfg.patch_size ** 2), int(self.model.conv_head[2].weight.shape[0])
        net = sf.sam_net if hname == "sam" else sf.clipseg_net
        return self.R, int(net.n_output_dims)

    def _load_inputs(self, step: int, parity: int, overlap: bool) -> None:
        """next_train(step) into the schedule's input buffers; the samplers' per-ray jitter (ray_samplers.py:105,318)."""
        dm = self.tr.pipeline.datamanager
        R = self.R
        _, _, _, pre = self._side_streams(overlap)
        xstep = pre.stream_id != self.main.stream_id
        pp = parity if xstep else None
        if xstep:
            # the prologue's stream loads its own inputs (parity buffers: the previous step's backward still reads the other set);
            # what this parity held was last read two steps ago, which the previous step's losses (awaited here) came after
            pre.wait_event(self.event("losses_done"))
        b = lambda name, shape: self.buf(name, shape, parity=pp)  # noqa: E731
        with torch.cuda.stream(pre):
            into = getattr(dm, "next_train_into", None)
            targets = {h: self.bufs[f"in_{h}@{parity}"] for h in self.heads}
            if into is not None and "next_train" not in dm.__dict__:
                into(step, b("in_o", (R, 3)), b("in_d", (R, 3)), b("in_image", (R, 3)), targets)
            else:
                rb, batch = dm.next_train(step)
                b("in_o", (R, 3)).copy_(rb.origins.reshape(R, 3))
                b("in_d", (R, 3)).copy_(rb.directions.reshape(R, 3))
                b("in_image", (R, 3)).copy_(batch["image"].reshape(R, 3))
                for h, t in targets.items():
                    t.copy_(batch[h].reshape(t.shape))
            ps = self.model.proposal_sampler
            for smp, name in ((ps.initial_sampler, "in_t_rand"), (ps.pdf_sampler, "in_u_rand")):
                dst = b(name, (R,))
                if smp.jitter_override is not None:
                    dst.copy_(smp.jitter_override.reshape(-1))
                else:
                    torch.rand((R,), out=dst)

    # ------------------------------------------------------------------------------------------------------------
    def run(self, step: int):
        """One train step: same contract as the eager `Trainer.train_iteration` body (callbacks are the caller's)."""
        tr, opt, model = self.tr, self.opt, self.model
        if torch.cuda.current_stream().stream_id != self.main.stream_id:
            raise RuntimeError("StepProgram was built for another main stream")
        ps = model.proposal_sampler
        updated = bool(ps._steps_since_update > ps.update_sched(ps._step) or ps._step < 10)
        ps.last_updated = updated
        overlap = bool(tr.overlap)
        if overlap and self.heads and tr._side is None:
            tr._side = {"sam": ops.make_stream("sam"), "clipseg": ops.make_stream("clipseg")}
        with_opt = bool(opt.enabled)
        parity = self.count & 1
        key = (parity, updated, with_opt, overlap, tr.presort_host, ops.PRESORT_SIDE_STREAM, tr.zero_grad_adam, self.world)
        plan = self.plans.get(key)
        if plan is None:
            # build every variant of this configuration now (both buffer parities x proposal update or not): the first step pays
            # for all of them, no later step stalls the host in the middle of a run
            for par in (parity, parity ^ 1):
                for h in self.heads:  # the targets' buffers exist before the first load
                    self.buf(f"in_{h}", self._target_shape(h), parity=par)
                for upd in (updated, not updated):
                    k2 = (par, upd) + key[2:]
                    if k2 not in self.plans:
                        self.plans[k2] = self._build(par, upd, with_opt, overlap, tr.zero_grad_adam)
            plan = self.plans[key]
        if overlap:  # a head task of this parity (two steps ago) may still be reading the buffers this step overwrites
            for h in self.heads:
                self._wait_head_free(self.main, parity, h)
        self._load_inputs(step, parity, overlap)
        # ---- this step's dynamic values
        stepped = []
        if with_opt:
            stepped = [g for g in opt.arenas if g != "proposal_networks" or updated or tr.zero_grad_adam]
        vals = {("anneal",): float(ps._anneal)}
        for g in opt.arenas:
            vals[("lr", g)] = float(opt.lr(g))
            vals[("t", g)] = int(opt.step_count[g] + 1)
        for k, sites in plan.dyn.items():
            v = vals[k]
            for a, i in sites:
                a[i] = v
        # ---- replay
        sel = ops._TIMING["names"]
        if _SKIP:  # measurement only (tools/ablate_step.sh): what is a kernel's time worth inside the concurrent step?
            for kind, fn, args, key_, _, _ in plan.entries:
                if kind != _KERNEL:
                    fn(*args)
                elif not any(s_ in key_ for s_ in _SKIP):
                    fn(*args)
        elif sel is None:
            for kind, fn, args, key_, _, _ in plan.entries:
                if kind == _KERNEL:
                    rc = fn(*args)
                    if rc:
                        _lib.check(rc, key_)
                else:
                    fn(*args)
        else:
            self._replay_timed(plan, sel)
        if not self.multi or RECORD_EXCHANGE:  # (the eager exchange_and_step counts the steps of its group itself)
            for g in stepped:
                if ("t", g) in plan.dyn:  # only groups this schedule has an Adam launch for (not 'conv' without a conv head)
                    opt.step_count[g] += 1
        if self.multi and with_opt:
            for p_ in self._tp_params:  # the other ranks' copies of the owned levels are behind now (refreshed before eval / saving)
                p_._tp_stale = True
        if updated:
            ps._steps_since_update = 0
        self.count += 1
        # ---- results (views of the schedule's buffers: valid until the next step of the same parity overwrites them)
        s = self.bufs["loss_summary"]
        loss_dict = {"rgb_loss": s[1], "interlevel_loss": s[2], "distortion_loss": s[3]}
        for h in self.heads:
            loss_dict[tr.HEAD_LOSS[h]] = self.bufs[f"mse_out_{h}"][0]
        metrics_dict = {"psnr": s[5], "distortion": s[4]}
        loss = s[0]
        if overlap and not tr.pipeline_steps:
            for h in self.heads:
                self.main.wait_stream(tr._side[h])
            self.join_side_streams()
        if self.heads and not (overlap and tr.pipeline_steps):
            loss = loss + sum(loss_dict[tr.HEAD_LOSS[h]] for h in self.heads)
        return loss, loss_dict, metrics_dict

    def schedule_info(self) -> Dict[str, object]:
        """What ran, for the bench line: HIP streams the recorded launches of a step go to, whether the heads' weight gradients have
        the shared fourth stream (one rank only: many ranks leave that hardware queue to RCCL), launches per step, the guard."""
        plan = max(self.plans.values(), key=lambda p: len(p.entries)) if self.plans else None
        if plan is None:
            return {}
        kern = [e for e in plan.entries if e[0] == _KERNEL]
        return {"streams": len({e[5].stream_id for e in kern}),
                "wgrad_stream": bool(WGRAD_STREAM and self.heads and not self.multi and self.tr.overlap),
                "launches_per_step": len(kern), "launches_on_main_stream": sum(1 for e in kern if e[5].stream_id == self.main.stream_id),
                "pipeline_steps": bool(getattr(self.tr, "pipeline_steps", False)), "step_guard": self._guard("nerf") is not None,
                "ranks": self.world}

    def join_side_streams(self) -> None:
        """Order the main stream after everything the schedule has put on its own side stream (the forward-time sorts and,
        without feature heads, the proposal backward + Adam and the next step's prologue).  `Trainer.synchronize()` and every
        un-pipelined step call it: parameters and moments of `proposal_networks` are then safe to read on the main stream."""
        if self._own_sort_stream is not None:
            self.main.wait_stream(self._own_sort_stream)

    def _replay_timed(self, plan: _Plan, sel) -> None:
        """bench.py's per-launch HIP-event timing (ops.enable_kernel_timing): events go on the stream of the launch."""
        evs = ops._TIMING["events"]
        for kind, fn, args, key, units, st in plan.entries:
            if kind != _KERNEL:
                fn(*args)
                continue
            name = key.partition("/")[0]
            if sel == "all" or key in sel or name in sel:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(st)
                rc = fn(*args)
                b.record(st)
                evs.setdefault(key, []).append((a, b, st.stream_id, units))
            else:
                rc = fn(*args)
            if rc:
                _lib.check(rc, key)

    def outputs(self) -> Dict[str, torch.Tensor]:
        """The model outputs of the last step (sam_model.py:233-301 keys), as views of the schedule's buffers."""
        parity = (self.count - 1) & 1
        out = {"rgb": self.bufs["out_rgb"], "accumulation": self.bufs["out_acc"], "depth": self.bufs["out_depth"],
               "prop_depth_0": self.bufs["out_pdepth"]}
        if "sam" in self.heads:
            out["sam"] = self.bufs["cv_y"] if self.cfg.patch_size > 1 else self.bufs["sam_fm"]
        if "clipseg" in self.heads:
            out["clipseg"] = self.bufs["clipseg_fm"]
        del parity
        return out

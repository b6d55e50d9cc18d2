"""Python face of the C-ABI: thin `torch.autograd.Function`s around libsamnerf_hip.so.

PyTorch is plumbing here (device memory, streams, autograd bookkeeping); every numeric step runs in a
hand-written gfx950 kernel.  There is no CPU / eager fallback: tensors must live on a ROCm device.

This module holds the parts with process-wide state (stream factory, weight-gradient companion streams, the hash-grid
backward modes and their forward-time sorts) and the ops built on them (hash grids, dense layers, the fused 64-wide nets,
the nerfacto field); the stateless groups live in ops_sampling / ops_render / ops_losses / ops_data / ops_vit / ops_optim
and are re-exported here, so `samnerf_amd.ops.<name>` is the one public spelling.

Gradient-arena convention: a parameter tensor may carry a `main_grad` attribute (an fp32 view into the
model's flat gradient arena, see `arena.py`).  When present, the backward kernels accumulate straight into
it and autograd receives `None` for that parameter -- no per-step zero-filled temporaries, and the arena is
the RCCL all-reduce buffer.  Without it the functions behave like ordinary autograd ops.
"""
from __future__ import annotations

import ctypes
import os as _os
from typing import List, Optional, Sequence, Tuple

import threading

import torch

from . import _lib
from ._opcore import *  # noqa: F401,F403
from ._opcore import (_FN, _L, _TIMING, _chk, _has_gpu, _launch, _linear_fwd_ws, _p, _stream)  # noqa: F401
from .ops_data import *  # noqa: F401,F403
from .ops_losses import *  # noqa: F401,F403
from .ops_losses import _Distortion, _Interlevel, _RowMSELoss  # noqa: F401
from .ops_optim import *  # noqa: F401,F403
from .ops_render import *  # noqa: F401,F403
from .ops_render import _CompositeRGB, _FeatureMean, _HeadInput, _TruncExpSel, _Weights  # noqa: F401
from .ops_sampling import *  # noqa: F401,F403
from .ops_vit import *  # noqa: F401,F403

HASHGRID_BWD_MODE = "sorted"  # "sorted": bucketed, atomic-free (default) | "atomic": global fp32 atomics
PRESORT_FIELD_GRID = True      # the field / proposal grids' backward sorts run at forward time (module constants: tests flip them)
PLANAR_FIELD_ENCODING = True   # level-major hand-off between the field grid and the base MLP
PRESORT_SIDE_STREAM = True     # False: forward-time sorts stay on the caller's stream (bench.py's serial replay)
HASHGRID_RUN_MAX_RES = 64.0    # levels up to this resolution merge equal-row contributions before their LDS atomics

# ---------------------------------------------------------------------------------------------
# stream factory
# ---------------------------------------------------------------------------------------------
# The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES (4) hardware queues, and which streams share a queue changes the
# step time by up to 15 %.  The step therefore runs on exactly 4 streams by default (main, two heads, the forward-time sort).
# All streams are made here, by name; STREAM_SLOTS[name] = position modulo 4 in the creation sequence (unused filler streams
# are created to reach it) is a tuning knob for runs with more streams (ops.WGRAD_SIDE_STREAM).
STREAM_SLOTS = {"sam": 0, "clipseg": 1, "presort": 2, "wgrad": 2, "wgrad:sam": 3, "wgrad:clipseg": 0, "wgrad:main": 1}
_STREAMS_MADE = {"n": 0, "filler": [], "names": {}}


STREAM_CACHE = _os.environ.get("SNF_STREAM_CACHE", "1") == "1"


def make_stream(name: str) -> "torch.cuda.Stream":
    """The task stream called `name` on the current device.  One stream per (device, name) and process: a second trainer in the
    same process (bench.py's other workloads, the second exchange mode of a multi-rank run) takes the FIRST trainer's streams
    instead of fresh ones -- fresh streams land on other hardware queues, and a trainer built second ran its steps 15 % slower
    (2.54 -> 2.92 ms, profiles/r04_experiments.txt)."""
    key = (torch.cuda.current_device(), name)
    if STREAM_CACHE and key in _STREAMS_MADE.setdefault("by_name", {}):
        return _STREAMS_MADE["by_name"][key]
    st = _make_stream(name)
    _STREAMS_MADE.setdefault("by_name", {})[key] = st
    return st


# SNF_STREAM_CUS="sam=192,clipseg=192": the named task streams are created on that many CUs only (snf_stream_create_cu_mask) --
# a probe of CU partitioning between the tasks of the step (profiles/r05_cu_mask.txt); unset: every stream sees the whole chip
# (`name=p1` / `name=p-1`: a stream of lower / higher priority instead)
STREAM_CUS = {k: v for k, v in (kv.split("=") for kv in _os.environ.get("SNF_STREAM_CUS", "").split(",") if "=" in kv)}
_MASKED: dict = {}


def masked_stream(n_cus: int, tag: str = "") -> "torch.cuda.Stream":
    """A HIP stream whose kernels run on the first `n_cus` CUs (spread evenly over the XCDs by the driver), as a torch stream object.
    One per (device, n_cus, tag) and process; the handle lives as long as the process."""
    import ctypes
    key = (torch.cuda.current_device(), int(n_cus), tag)
    st = _MASKED.get(key)
    if st is None:
        h = ctypes.c_void_p()
        _lib.check(_L().snf_stream_create_cu_mask(int(n_cus), ctypes.byref(h)), "snf_stream_create_cu_mask")
        st = _MASKED[key] = torch.cuda.ExternalStream(h.value)
    return st


def priority_stream(priority: int, tag: str = "") -> "torch.cuda.Stream":
    """A HIP stream of the given priority (snf_stream_create_priority: > 0 lower than default) as a torch stream object."""
    import ctypes
    key = (torch.cuda.current_device(), "prio", int(priority), tag)
    st = _MASKED.get(key)
    if st is None:
        h = ctypes.c_void_p()
        _lib.check(_L().snf_stream_create_priority(int(priority), ctypes.byref(h)), "snf_stream_create_priority")
        st = _MASKED[key] = torch.cuda.ExternalStream(h.value)
    return st


def _make_stream(name: str) -> "torch.cuda.Stream":
    if name in STREAM_CUS:
        spec = STREAM_CUS[name]
        st = priority_stream(int(spec[1:]), name) if spec.startswith("p") else masked_stream(int(spec), name)
        _STREAMS_MADE["names"][st.stream_id] = name
        return st
    slot = STREAM_SLOTS.get(name)
    if slot is not None:
        while _STREAMS_MADE["n"] % 4 != slot % 4:
            _STREAMS_MADE["filler"].append(torch.cuda.Stream())
            _STREAMS_MADE["n"] += 1
    st = torch.cuda.Stream()  # (stream priorities were measured: no effect on the step)
    _STREAMS_MADE["n"] += 1
    _STREAMS_MADE["names"][st.stream_id] = name
    return st


def stream_name(st) -> str:
    return _STREAMS_MADE["names"].get(st.stream_id, "main")


# ---------------------------------------------------------------------------------------------
# weight gradients on a companion stream
# ---------------------------------------------------------------------------------------------
# A layer's weight gradient feeds nothing but the optimizer, while its data gradient is on the backward's dependency chain.
# With WGRAD_SIDE_STREAM on, every snf_linear_bwd_weight launch of a task goes to a companion HIP stream of the task's
# stream and runs beside the following data-gradient kernels; the optimizer joins the companion before it steps
# (join_wgrad_stream, called by engine.Optimizers.exchange_and_step).
# Off by default since r01p: with the companions the step runs on 7 streams, the runtime multiplexes them onto its 4 hardware
# queues differently from run to run, and about one run in three lands in a 5-15 % slower mode (8 + 8 interleaved runs on one
# box: 3.84-3.94 ms without companions, 3.87-4.58 ms with); on exactly 4 streams every run is in the fast mode, and the
# companions' own gain (-1.5 % when measured in r01m) is gone since the table Adam moved into the backward.
WGRAD_SIDE_STREAM = False
_WGRAD_STREAMS: dict = {}


_WGRAD_PENDING: list = []  # (task stream, companion) pairs of the running backward pass
_GRAD_STREAMS: list = []   # streams on which gradient-writing nodes of the running backward pass were executed


def _join_pending_wgrads() -> None:
    """Final callback of the backward pass (runs on the thread, hence the stream, that called backward()): every task stream
    waits for its companion, and the caller's stream waits for every stream a gradient-writing node ran on -- what autograd
    does for AccumulateGrad leaves and cannot know for gradients written in place into the arenas.  So, as autograd
    promises, all gradients are ready on the stream that called backward() when it returns."""
    while _WGRAD_PENDING:
        cur, side = _WGRAD_PENDING.pop()
        cur.wait_stream(side)
    caller = torch.cuda.current_stream()
    while _GRAD_STREAMS:
        st = _GRAD_STREAMS.pop()
        if st.stream_id != caller.stream_id:
            caller.wait_stream(st)


def _queue_final_callback() -> bool:
    if _WGRAD_PENDING or _GRAD_STREAMS:
        return True  # already queued for this pass
    try:
        torch.autograd.Variable._execution_engine.queue_callback(_join_pending_wgrads)
        return True
    except RuntimeError:  # not inside a backward pass
        return False


def _note_grad_stream() -> None:
    """Called by every backward node that writes parameter gradients in place."""
    if not _has_gpu():
        return
    cur = torch.cuda.current_stream()
    if any(s.stream_id == cur.stream_id for s in _GRAD_STREAMS):
        return
    if _queue_final_callback():
        _GRAD_STREAMS.append(cur)


class _wgrad_stream:
    """Context (used inside autograd backward functions): the current stream becomes the companion of the caller's stream,
    ordered after the work issued so far; the listed tensors are kept alive for the companion."""

    def __init__(self, *tensors):
        self.tensors = tensors
        self.ctx = None

    def __enter__(self):
        if not (WGRAD_SIDE_STREAM and _has_gpu()):
            return self
        cur = torch.cuda.current_stream()
        if not any(c.stream_id == cur.stream_id for c, _ in _WGRAD_PENDING):
            if not _queue_final_callback():  # not inside a backward pass: stay on the caller's stream
                return self
            side = _WGRAD_STREAMS.get(cur.stream_id)
            if side is None:
                side = _WGRAD_STREAMS[cur.stream_id] = make_stream("wgrad:" + stream_name(cur))
            _WGRAD_PENDING.append((cur, side))
        side = _WGRAD_STREAMS[cur.stream_id]
        side.wait_stream(cur)
        for t in self.tensors:
            if t is not None and t.is_cuda:
                t.record_stream(side)
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


def join_wgrad_stream() -> None:
    """Order the current stream after the weight-gradient launches its companion stream has received (normally done by
    the backward pass's final callback; harmless to repeat)."""
    if torch.cuda.is_available():
        cur = torch.cuda.current_stream()
        side = _WGRAD_STREAMS.get(cur.stream_id)
        if side is not None:
            cur.wait_stream(side)


# Inside `Function.forward` grad mode is always off and `ctx.needs_input_grad` only says which inputs require grad -- also under
# torch.no_grad(), where no backward will ever come.  The wrappers below note the caller's grad mode here so that the forward
# passes do not prepare one in eval: no backward sorts, no saved hidden activations, no row-major detour for the sorted backward
# (the render path spent 11 of 58 ms per 512 x 512 image in sorts it never used, profiles/r03_render_before.txt).
# (per thread, like torch's grad mode itself: a no_grad render on another thread must not flip it under a training forward)
class _Track(threading.local):
    on = True

    def __getitem__(self, _i):  # `_TRACK[0]`: the caller's grad mode on this thread
        return self.on


_TRACK = _Track()


def _apply(fn, *args):
    prev = _TRACK.on
    _TRACK.on = torch.is_grad_enabled()
    try:
        return fn.apply(*args)
    finally:
        _TRACK.on = prev


def _grad_target(param: torch.Tensor) -> Tuple[torch.Tensor, bool]:
    """(buffer to accumulate into, whether autograd should get None)."""
    _note_grad_stream()
    mg = getattr(param, "main_grad", None)
    if mg is not None:
        return mg, True
    return torch.zeros_like(param), False


# ---------------------------------------------------------------------------------------------
# hash grid
# ---------------------------------------------------------------------------------------------
def _sc_values(sc: torch.Tensor) -> tuple:
    """The level resolutions as a host tuple: one device read per tensor OBJECT, cached on the object itself (a cache keyed
    by data_ptr would hand a dead tensor's values to whatever is allocated at its address next)."""
    v = sc.__dict__.get("_snf_values")
    if v is None:
        v = tuple(sc.detach().cpu().tolist())
        sc.__dict__["_snf_values"] = v
    return v


def hashgrid_run_levels(sc: torch.Tensor) -> int:
    """Leading levels whose backward sums runs of equal rows before sorting (resolution <= HASHGRID_RUN_MAX_RES: cells wide
    enough that consecutive samples of a ray share them)."""
    return sum(1 for s in _sc_values(sc) if s <= HASHGRID_RUN_MAX_RES)


def _geometry_key(sc: torch.Tensor, L: int, T: int):
    """(levels, log2_T, resolutions) -- what a backward sort depends on besides the positions."""
    return (L, T, _sc_values(sc))


_PRESORT_STREAM = {}
# A task stream that is idle while the nerfacto forward runs (the trainer hands over a head's stream): the forward-time sorts
# then need no stream of their own -- the step stays on three streams, one hardware queue each.
PRESORT_HOST_STREAM = None


@torch.no_grad()
def hashgrid_presort(u: torch.Tensor, sc: torch.Tensor, L: int, T: int, side_stream: bool = False) -> None:
    """Sort the (sample, level, corner) records of positions `u` for one level geometry NOW (forward pass) and attach the
    result to `u`: every hash-grid backward at these positions with this geometry then skips its own sort.
    side_stream: run the sort on a dedicated HIP stream beside the caller's forward kernels (the backward waits on its event)."""
    u = _chk(u, "u")
    N = u.shape[0]
    if HASHGRID_BWD_MODE != "sorted" or N > HASHGRID_BWD_MAX_SAMPLES or 8 * L * N >= (1 << 32) or T > 23:
        return
    cache = u.__dict__.setdefault("_snf_sorted", {})
    key = _geometry_key(sc, L, T)
    if key in cache:
        return
    nbytes = int(_L().snf_hashgrid_bwd_workspace_bytes(N, L, T))
    if not (side_stream and PRESORT_SIDE_STREAM):
        ws = torch.empty(((nbytes + 3) // 4,), device=u.device, dtype=torch.int32)
        _launch("snf_hashgrid_sort", _p(u), _p(sc), N, L, T, _p(ws), nbytes, _stream(), tag=f"L{L}")
        cache[key] = (ws, None)
        return
    cur = torch.cuda.current_stream()
    st = PRESORT_HOST_STREAM
    if st is None:
        st = _PRESORT_STREAM.setdefault(u.device.index, None) or make_stream("presort")
        _PRESORT_STREAM[u.device.index] = st
    st.wait_stream(cur)  # the positions are ready
    with torch.cuda.stream(st):
        ws = torch.empty(((nbytes + 3) // 4,), device=u.device, dtype=torch.int32)
        _launch("snf_hashgrid_sort", _p(u), _p(sc), N, L, T, _p(ws), nbytes, _stream(), tag=f"L{L}")
        ev = torch.cuda.Event()
        ev.record(st)
    u.record_stream(st)
    sc.record_stream(st)
    cache[key] = (ws, ev)


HASHGRID_BWD_MAX_SAMPLES = 1 << 21  # per launch of the sorted backward (21 sample bits in a record)


class FusedAdam:
    """What the hash-grid backward needs to BE the optimizer step of a table (snf_hashgrid_bwd_presorted_adam): the flat
    param / exp_avg / exp_avg_sq views of the table, Adam's hyper-parameters for this step and the first level to fuse (the
    coarse levels before it keep the reachable-row Adam launch).  engine.Optimizers arms a table with one of these per step;
    `done` reports back which levels the backward has stepped."""
    __slots__ = ("p", "m", "v", "lr", "b1", "b2", "eps", "step", "scale", "from_level", "done")

    def __init__(self, p, m, v, lr, b1, b2, eps, step, scale, from_level):
        self.p, self.m, self.v = p, m, v
        self.lr, self.b1, self.b2, self.eps, self.step, self.scale = lr, b1, b2, eps, step, scale
        self.from_level = from_level
        self.done = None  # (first level, end level) stepped by the backward

    def levels(self, l0: int, nl: int, T: int, F: int):
        """The same for the level sub-range [l0, l0+nl) of the table (table-parallel runs)."""
        a, b = (l0 << T) * F, ((l0 + nl) << T) * F
        sub = FusedAdam(self.p[a:b], self.m[a:b], self.v[a:b], self.lr, self.b1, self.b2, self.eps, self.step, self.scale,
                        min(max(self.from_level - l0, 0), nl))
        return sub


def _hashgrid_bwd_launch(u, g, sc, N, L, F, T, ld, col, buf, adam: Optional[FusedAdam] = None) -> bool:
    """Table-gradient backward of one grid (or level run).  Returns True when `adam` was given and the launch stepped the
    levels >= adam.from_level itself (only the presorted single-launch path can)."""
    if HASHGRID_BWD_MODE == "atomic":
        _launch("snf_hashgrid_bwd", _p(u), _p(g), _p(sc), N, L, F, T, ld, col, _p(buf), _stream(), tag=f"F{F}L{L}")
        return False
    # run aggregation pays for the ray-ordered F = 2 grids (field grid -8 %); the top-K-ordered F = 8 feature grids have
    # shorter runs and 4x the shuffle work per record (+20 % measured), so they keep the plain reduce
    nrun = hashgrid_run_levels(sc) if F == 2 else 0
    if adam is not None and adam.done is not None:
        raise RuntimeError("a hash table armed for the fused backward + Adam received a second backward launch in one step "
                           "(shared table or gradient accumulation): the fused step needs the table's whole gradient")
    presorted = getattr(u, "_snf_sorted", None)
    if presorted:
        hit = presorted.get(_geometry_key(sc, L, T))
        if hit is not None:
            ws, ev = hit
            if ev is not None:  # sorted on the side stream: order this stream after it, keep the buffer alive for it
                torch.cuda.current_stream().wait_event(ev)
                ws.record_stream(torch.cuda.current_stream())
            stage = None if ld == 0 else torch.empty((L * N * F,), device=g.device, dtype=torch.float32)
            if adam is not None and adam.from_level < L:
                fused = ((L - adam.from_level) << T) * F
                _launch("snf_hashgrid_bwd_presorted_adam", _p(g), N, L, F, T, ld, col, nrun, _p(buf), _p(ws), _p(stage),
                        adam.from_level, _p(adam.p), _p(adam.m), _p(adam.v), float(adam.lr), float(adam.b1), float(adam.b2),
                        float(adam.eps), int(adam.step), float(adam.scale), _stream(), tag=f"F{F}L{L}",
                        # algorithmic bytes: every corner contribution (F floats) read once, written back as a gradient
                        # only on the levels left to the caller; p, exp_avg, exp_avg_sq read + written on the fused ones
                        units=float(N) * 8 * F * 4 * (L + adam.from_level) + 24.0 * fused)
                return True
            _launch("snf_hashgrid_bwd_presorted", _p(g), N, L, F, T, ld, col, nrun, _p(buf), _p(ws), _p(stage), _stream(),
                    tag=f"F{F}L{L}")
            return False
    # batches beyond 2^21 samples (or 2^32 records) go through in slices: the gradient table accumulates
    per = min(HASHGRID_BWD_MAX_SAMPLES, ((1 << 32) - 1) // (8 * L))
    ws = None
    for n0 in range(0, N, per):
        n = min(per, N - n0)
        nbytes = int(_L().snf_hashgrid_bwd_workspace_bytes(n, L, T))
        if ws is None:
            ws = torch.empty(((nbytes + 3) // 4,), device=g.device, dtype=torch.int32)
        _launch("snf_hashgrid_bwd_sorted_ex", ctypes.c_void_p(u.data_ptr() + n0 * 3 * 4), ctypes.c_void_p(g.data_ptr() + n0 * ld * 4),
                _p(sc), n, L, F, T, ld, col, nrun, _p(buf), _p(ws), nbytes, _stream(), tag=f"F{F}L{L}")
    return False



@torch.no_grad()
def hashgrid_fwd_raw(u, table, scalings, L: int, F: int, T: int, planar: bool = False):
    """snf_hashgrid_fwd without autograd: [N, L F] (planar: level-major [L, N, F]) -- tools/bench_hashgrid_fwd.py."""
    u, table = _chk(u, "u"), _chk(table, "table")
    N = u.shape[0]
    out = torch.empty((L, N, F) if planar else (N, L * F), device=u.device, dtype=torch.float32)
    _launch("snf_hashgrid_fwd", _p(u), _p(table), _p(scalings), N, L, F, T, _p(out), 0 if planar else L * F, 0, _stream(),
            tag=f"F{F}L{L}")
    return out

class _HashGridMulti(torch.autograd.Function):
    """One or more hash grids evaluated at the same points, outputs concatenated along the feature axis."""

    @staticmethod
    def forward(ctx, u, specs, *tables):
        # specs: tuple of (scalings tensor, L, F, log2_T) per grid
        u = _chk(u, "u")
        N = u.shape[0]
        total = sum(L * F for (_, L, F, _) in specs)
        out = torch.empty((N, total), device=u.device, dtype=torch.float32)
        col = 0
        if _TRACK[0] and PRESORT_FIELD_GRID and N >= (1 << 16):
            # backward sorts that are not attached to `u` yet (the proposal grid) start now, on the side stream
            for (sc, L, F, T), need in zip(specs, ctx.needs_input_grad[2:]):
                if need:
                    hashgrid_presort(u, sc, L, T, side_stream=True)
        for (sc, L, F, T), tab in zip(specs, tables):
            tab = _chk(tab, "table")
            assert tab.numel() == (L << T) * F, "table size does not match (levels, log2_T, features)"
            _launch("snf_hashgrid_fwd", _p(u), _p(tab), _p(sc), N, L, F, T, _p(out), total, col, _stream(), tag=f"F{F}L{L}")
            col += L * F
        ctx.specs = specs
        ctx.tables = tables
        ctx.save_for_backward(u)
        return out

    @staticmethod
    def backward(ctx, g):
        (u,) = ctx.saved_tensors
        g = _chk(g, "grad_out")
        N, total = g.shape
        grads: List[Optional[torch.Tensor]] = []
        col = 0
        for (sc, L, F, T), tab in zip(ctx.specs, ctx.tables):
            if not tab.requires_grad:
                grads.append(None)
            else:
                buf, fused = _grad_target(tab)
                adam = getattr(tab, "_fused_adam", None) if fused else None
                if _hashgrid_bwd_launch(u, g, sc, N, L, F, T, total, col, buf, adam):
                    adam.done = (adam.from_level, L)
                grads.append(None if fused else buf)
            col += L * F
        return (None, None, *grads)


# -- table parallelism (distributed.py: TableParallelLayout) ------------------------------------------------------------
TABLE_PARALLEL = _os.environ.get("SNF_TABLE_PARALLEL", "1") == "1"


def table_parallel_layout(specs):
    """The slab-ownership layout of a head whose grids are `specs`, or None when the head stays replicated (one rank, the
    switch is off, F != 8, or the slab count does not divide by the world size)."""
    from . import distributed as D
    if not (TABLE_PARALLEL and D.collectives_on()):
        return None
    grids = [(L, F, T) for (_, L, F, T) in specs]
    world = D.world_size()
    if grids[0][1] != 8 or not D.TableParallelLayout.supported(grids, world):
        return None
    return D.TableParallelLayout(grids, world)


def _sc_run(sc: torch.Tensor, l0: int, nl: int) -> torch.Tensor:
    """scalings[l0 : l0+nl] as ONE tensor object per run (kept on `sc`), so its cached host values are read once."""
    runs = sc.__dict__.setdefault("_snf_runs", {})
    r = runs.get((l0, nl))
    if r is None:
        r = runs[(l0, nl)] = sc[l0:l0 + nl]
    return r


def tp_gathered_positions(u: torch.Tensor) -> torch.Tensor:
    """Positions of every rank's samples, gathered once per `u` (the heads share it, and the backward sorts attach to it)."""
    from . import distributed as D
    U = u.__dict__.get("_snf_tp_gathered")
    if U is None:
        U = D.tp_gather_positions(_chk(u, "u"))
        u.__dict__["_snf_tp_gathered"] = U
    return U


def tp_presort(u: torch.Tensor, specs, layout) -> None:
    """Forward-time sorts of the gathered positions for the level runs this rank owns."""
    from . import distributed as D
    import torch.distributed as dist
    U = tp_gathered_positions(u)
    for gi, l0, nl, _ in layout.runs(dist.get_rank()):
        sc, _, _, T = specs[gi]
        hashgrid_presort(U, _sc_run(sc, l0, nl), nl, T)


def tp_eval_run(U: torch.Tensor, specs, tables):
    """eval_run(grid, first_level, n_levels, out, ld, col) for distributed.tp_forward: the levels of one owned run at every
    row of U, written to out[:, col : col + n_levels*F]."""
    M = U.shape[0]

    def eval_run(gi, l0, nl, out, ld, col):
        sc, L, F, T = specs[gi]
        tab = _chk(tables[gi], "table")
        assert tab.numel() == (L << T) * F, "table size does not match (levels, log2_T, features)"
        _launch("snf_hashgrid_fwd", _p(U), ctypes.c_void_p(tab.data_ptr() + ((l0 << T) * F) * 4), _p(_sc_run(sc, l0, nl)),
                M, nl, F, T, _p(out), ld, col, _stream(), tag=f"F{F}L{nl}tp")

    return eval_run


def tp_accumulate(U: torch.Tensor, G: torch.Tensor, specs, tables, layout, rank: int) -> List[Optional[torch.Tensor]]:
    """Scatter-add the gradients G [rows of U, per*F] of the columns `rank` owns into the owned levels of the tables'
    gradient buffers.  Returns, per table, None (arena gradient written in place) or the full-size gradient tensor."""
    grads: List[Optional[torch.Tensor]] = [None] * len(tables)
    for gi, l0, nl, col in layout.runs(rank):
        sc, L, F, T = specs[gi]
        tab = tables[gi]
        if not tab.requires_grad:
            continue
        buf, fused = _grad_target(tab)
        adam = getattr(tab, "_fused_adam", None) if fused else None
        sub = adam.levels(l0, nl, T, F) if adam is not None else None
        if _hashgrid_bwd_launch(U, G, _sc_run(sc, l0, nl), U.shape[0], nl, F, T, layout.width, col,
                                buf.view(-1)[(l0 << T) * F:((l0 + nl) << T) * F], sub):
            adam.done = (l0 + sub.from_level, l0 + nl)
        tab._tp_stale = True  # the other ranks' copies of these levels are behind once Adam has run
        if not fused:
            grads[gi] = buf  # no arena: autograd gets the full-size gradient (zero outside the owned levels)
    return grads


class _HashGridTableParallel(torch.autograd.Function):
    """_HashGridMulti with the tables sharded by level over the ranks: this rank evaluates (and accumulates gradients for)
    its own levels at the samples of all ranks; features and their gradients cross the links, the tables never do."""

    @staticmethod
    def forward(ctx, u, specs, layout, *tables):
        from . import distributed as D
        u = _chk(u, "u")
        U = tp_gathered_positions(u)
        out = D.tp_forward(U, u.shape[0], layout, tp_eval_run(U, specs, tables))
        ctx.specs, ctx.layout, ctx.tables, ctx.U = specs, layout, tables, U
        return out

    @staticmethod
    def backward(ctx, g):
        from . import distributed as D
        import torch.distributed as dist
        g = _chk(g, "grad_out")
        G = D.tp_backward(g, ctx.layout)  # [W*N, per*F]
        grads = tp_accumulate(ctx.U, G, ctx.specs, ctx.tables, ctx.layout, dist.get_rank())
        return (None, None, None, *grads)


def _tp_refresh(tables) -> None:
    """Replicated evaluation of tables that were trained table-parallel: make them whole first (collective)."""
    for tab in tables:
        if getattr(tab, "_tp_stale", False):
            refresh = getattr(tab, "_tp_refresh", None)
            if refresh is None:
                raise RuntimeError("hash table was trained table-parallel but has no owner map to consolidate it "
                                   "(engine.Optimizers attaches one)")
            refresh()
            tab._tp_stale = False


def hashgrid(u, tables: Sequence[torch.Tensor], specs) -> torch.Tensor:
    specs = tuple(specs)
    if torch.is_grad_enabled() and any(t.requires_grad for t in tables):
        layout = table_parallel_layout(specs)
        if layout is not None:
            return _apply(_HashGridTableParallel, u, specs, layout, *tables)
    _tp_refresh(tables)
    return _apply(_HashGridMulti, u, specs, *tables)


# ---------------------------------------------------------------------------------------------
# dense layer
# ---------------------------------------------------------------------------------------------
class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, act: int):
        x, w = _chk(x, "x"), _chk(w, "w")
        N, I = x.shape
        O = w.shape[0]
        assert w.shape[1] == I
        y = torch.empty((N, O), device=x.device, dtype=torch.float32)
        _launch("snf_linear_fwd", _p(x), _p(w), _p(b), N, I, O, I, O, act, _p(y), _stream(), tag=f"{I}x{O}")
        ctx.act = act
        ctx.wref, ctx.bref = w, b
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y = ctx.saved_tensors
        w, b, act = ctx.wref, ctx.bref, ctx.act
        gy = _chk(gy, "grad_y")
        N, I = x.shape
        O = w.shape[0]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty((N, I), device=x.device, dtype=torch.float32)
            _launch("snf_linear_bwd_data", _p(gy), _p(y), _p(w), N, I, O, O, O, I, act, _p(gx), _stream(), tag=f"{I}x{O}")
        if w.requires_grad or (b is not None and b.requires_grad):
            wbuf, wfused = _grad_target(w)
            bbuf, bfused = (None, True) if b is None else _grad_target(b)
            with _wgrad_stream(gy, y, x, wbuf, bbuf):
                _launch("snf_linear_bwd_weight", _p(gy), _p(y), _p(x), N, I, O, O, O, I, act, _p(wbuf), _p(bbuf), _stream(),
                        tag=f"{I}x{O}")
            gw = None if wfused else wbuf
            gb = None if bfused else bbuf
        return gx, gw, gb, None


def linear(x, w, b=None, act: int = ACT_NONE) -> torch.Tensor:
    return _apply(_Linear, x, w, b, act)


def mlp(x, weights: Sequence[torch.Tensor], biases=None, out_act: int = ACT_NONE) -> torch.Tensor:
    n = len(weights)
    for i, w in enumerate(weights):
        b = None if biases is None else biases[i]
        x = linear(x, w, b, ACT_RELU if i < n - 1 else out_act)
    return x


# ---------------------------------------------------------------------------------------------
# the proposal networks' tiny density MLP (csrc/mlp_tiny.hip)
# ---------------------------------------------------------------------------------------------
def mlp_tiny_supported(in_dim: int, weights: Sequence[torch.Tensor], out_act: int) -> bool:
    return (len(weights) == 2 and out_act == ACT_NONE
            and bool(_L().snf_mlp_tiny_supported(in_dim, weights[0].shape[0], weights[1].shape[0])))


class _MLPTiny(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w0, w1):
        x = _chk(x, "x")
        N, I = x.shape
        H = w0.shape[0]
        need = _TRACK[0] and any(ctx.needs_input_grad)
        hid = torch.empty((N, H), device=x.device, dtype=torch.float32) if need else None
        y = torch.empty((N, 1), device=x.device, dtype=torch.float32)
        _launch("snf_mlp_tiny_fwd", _p(x), I, _p(w0), _p(w1), I, H, N, _p(hid), _p(y), _stream(), tag=f"{I}x{H}x1")
        ctx.refs = (w0, w1)
        if need:
            ctx.save_for_backward(x, hid)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, hid = ctx.saved_tensors
        w0, w1 = ctx.refs
        gy = _chk(gy, "grad_y")
        N, I = x.shape
        H = w0.shape[0]
        gx = torch.empty((N, I), device=x.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        w0buf, f0 = _grad_target(w0)
        w1buf, f1 = _grad_target(w1)
        _launch("snf_mlp_tiny_bwd", _p(gy), _p(x), I, _p(hid), _p(w0), _p(w1), I, H, N, _p(gx), I, _p(w0buf), _p(w1buf),
                _stream(), tag=f"{I}x{H}x1")
        return gx, (None if f0 else w0buf), (None if f1 else w1buf)


def mlp_tiny(x, w0, w1) -> torch.Tensor:
    return _apply(_MLPTiny, x, w0, w1)


# ---------------------------------------------------------------------------------------------
# SAM conv head as GEMMs (csrc/patchconv.hip)
# ---------------------------------------------------------------------------------------------
class _ConvHead(torch.autograd.Function):
    """Conv2d(C,C,k,pad) -> ReLU -> Conv2d(C,C,k,pad) -> mean over the p x p patch (samnerf/sam_model.py:196-200,259-264)
    on channel-last rows x [R, C] -> [R/p^2, C]: unfold + GEMM + ReLU, then patch-mean of the unfolded rows + GEMM."""

    @staticmethod
    def forward(ctx, x, w0, b0, w1, b1, p: int):
        x = _chk(x, "x")
        R, C = x.shape
        O0, O1, k = w0.shape[0], w1.shape[0], w0.shape[-1]
        kk = k * k
        assert w0.is_contiguous() and w1.is_contiguous() and w0.shape[1] == C and w1.shape[1] == O0 and w1.shape[-1] == k
        dev, st = x.device, _stream()
        col = torch.empty((R, C * kk), device=dev, dtype=torch.float32)
        _launch("snf_patch_unfold", _p(x), R, p, C, k, _p(col), st)
        h = torch.empty((R, O0), device=dev, dtype=torch.float32)
        _linear_fwd_ws(col, w0, b0, R, C * kk, O0, ACT_RELU, h, st, f"{C * kk}x{O0}")
        npatch = R // (p * p)
        cm = torch.empty((npatch, O0 * kk), device=dev, dtype=torch.float32)
        _launch("snf_patch_unfold_mean", _p(h), R, p, O0, k, _p(cm), st)
        y = torch.empty((npatch, O1), device=dev, dtype=torch.float32)
        _linear_fwd_ws(cm, w1, b1, npatch, O0 * kk, O1, ACT_NONE, y, st, f"{O0 * kk}x{O1}pm")
        ctx.p, ctx.k = p, k
        ctx.refs = (w0, b0, w1, b1)
        if _TRACK[0] and any(ctx.needs_input_grad):
            ctx.save_for_backward(col, h, cm)
        return y

    @staticmethod
    def backward(ctx, gy):
        col, h, cm = ctx.saved_tensors
        w0, b0, w1, b1 = ctx.refs
        p, k = ctx.p, ctx.k
        kk = k * k
        gy = _chk(gy, "grad_y")
        R, npatch = h.shape[0], cm.shape[0]
        C, O0, O1 = col.shape[1] // kk, w0.shape[0], w1.shape[0]
        dev, st = h.device, _stream()
        grads = [None] * 6
        # second convolution (+ mean): dW1, db1, d(cm)
        w1buf, w1f = _grad_target(w1)
        b1buf, b1f = (None, True) if b1 is None else _grad_target(b1)
        with _wgrad_stream(gy, cm, w1buf, b1buf):
            _launch("snf_linear_bwd_weight", _p(gy), None, _p(cm), npatch, O0 * kk, O1, O1, O1, O0 * kk, ACT_NONE, _p(w1buf),
                    _p(b1buf), _stream(), tag=f"{O0 * kk}x{O1}pm")
        dcm = torch.empty((npatch, O0 * kk), device=dev, dtype=torch.float32)
        _launch("snf_linear_bwd_data", _p(gy), None, _p(w1), npatch, O0 * kk, O1, O1, O1, O0 * kk, ACT_NONE, _p(dcm), st,
                tag=f"{O0 * kk}x{O1}pm")
        dh = torch.empty((R, O0), device=dev, dtype=torch.float32)
        _launch("snf_patch_fold_mean", _p(dcm), R, p, O0, k, _p(dh), st)
        # first convolution (ReLU derivative taken from h inside the GEMM loaders): dW0, db0, d(col) -> dx
        w0buf, w0f = _grad_target(w0)
        b0buf, b0f = (None, True) if b0 is None else _grad_target(b0)
        with _wgrad_stream(dh, h, col, w0buf, b0buf):
            _launch("snf_linear_bwd_weight", _p(dh), _p(h), _p(col), R, C * kk, O0, O0, O0, C * kk, ACT_RELU, _p(w0buf),
                    _p(b0buf), _stream(), tag=f"{C * kk}x{O0}")
        if ctx.needs_input_grad[0]:
            dcol = torch.empty((R, C * kk), device=dev, dtype=torch.float32)
            _launch("snf_linear_bwd_data", _p(dh), _p(h), _p(w0), R, C * kk, O0, O0, O0, C * kk, ACT_RELU, _p(dcol), st,
                    tag=f"{C * kk}x{O0}")
            dx = torch.empty((R, C), device=dev, dtype=torch.float32)
            _launch("snf_patch_fold", _p(dcol), R, p, C, k, _p(dx), st)
            grads[0] = dx
        grads[1] = None if w0f else w0buf
        grads[2] = None if b0f else b0buf
        grads[3] = None if w1f else w1buf
        grads[4] = None if b1f else b1buf
        return tuple(grads)


def conv_head(x, w0, b0, w1, b1, patch: int) -> torch.Tensor:
    """x [R, C] channel-last patch rows -> [R/patch^2, O]."""
    return _apply(_ConvHead, x, w0, b0, w1, b1, patch)


# ---------------------------------------------------------------------------------------------
# fused 64-wide MLP (activations in registers; see csrc/mlp_chain.hip)
# ---------------------------------------------------------------------------------------------
def mlp64_supported(in_dim: int, weights: Sequence[torch.Tensor]) -> bool:
    n = len(weights)
    return (n in (2, 3) and in_dim <= 32 and all(w.shape[1] == 64 for w in weights[1:])
            and all(w.shape[0] == 64 for w in weights[:-1]) and weights[-1].shape[0] <= 32)


def _mlp64_fwd_launch(x, in_real, ws, out_act, save: bool, planar_rows: int = 0):
    """planar_rows = N: `x` is the flat level-major encoding [in_real/2][N][2] (ldx = 0 in the C-ABI)."""
    N, ldx = (planar_rows, 0) if planar_rows else x.shape
    nh = len(ws) - 1
    out = ws[-1].shape[0]
    dev = x.device
    h1 = torch.empty((N, 64), device=dev, dtype=torch.float32) if save else None
    h2 = torch.empty((N, 64), device=dev, dtype=torch.float32) if (save and nh == 2) else None
    y = torch.empty((N, out), device=dev, dtype=torch.float32)
    _launch("snf_mlp64_fwd", _p(x), ldx, _p(ws[0]), in_real, _p(ws[1] if nh == 2 else None), _p(ws[-1]), nh, out,
            out_act, N, _p(h1), _p(h2), _p(y), out, _stream(), tag=f"{in_real}x{'x'.join(['64'] * nh)}x{out}")
    return y, h1, h2


def _mlp64_bwd_launch(x, in_real, ws, out_act, y, h1, h2, dy, lddy, dy_col_off, dy0, need_dx: bool, planar_rows: int = 0):
    """-> dX [N,32] (or None).  Weight gradients are accumulated into the weights' grad targets; returns them too.
    planar_rows = N: `x` is level-major [16][N][2] and dX is produced in the same layout (flat, 32*N)."""
    N, ldx = (planar_rows, 0) if planar_rows else x.shape
    nh = len(ws) - 1
    out = ws[-1].shape[0]
    dev = x.device
    ldz = (out + 3) // 4 * 4
    dh1 = torch.empty((N, 64), device=dev, dtype=torch.float32)
    dh2 = torch.empty((N, 64), device=dev, dtype=torch.float32) if nh == 2 else None
    dz = torch.empty((N, ldz), device=dev, dtype=torch.float32)
    dx = (torch.empty((32 * N,) if planar_rows else (N, 32), device=dev, dtype=torch.float32)) if need_dx else None
    tag = f"{in_real}x{'x'.join(['64'] * nh)}x{out}"
    _launch("snf_mlp64_bwd_data", _p(dy), lddy, dy_col_off, _p(dy0), _p(y), out, _p(ws[0]), in_real,
            _p(ws[1] if nh == 2 else None), _p(ws[-1]), nh, out, out_act, N, _p(h1), _p(h2), _p(dh1), _p(dh2), _p(dz),
            ldz, _p(dx), 0 if planar_rows else 32, _stream(), tag=tag)
    grads = []
    # (dZ, Hlast) -> dWout ; (dH2, H1) -> dW1 ; (dH1, X) -> dW0      [all pre-masked: act = NONE]
    pairs = [(dh1, 64, x, ldx, in_real, ws[0])]
    if nh == 2:
        pairs.append((dh2, 64, h1, 64, 64, ws[1]))
    pairs.append((dz, ldz, h2 if nh == 2 else h1, 64, 64, ws[-1]))
    for (g, ldg, a, lda, I, w) in pairs:
        if not w.requires_grad:
            grads.append(None)
            continue
        O = w.shape[0]
        buf, fused = _grad_target(w)
        with _wgrad_stream(g, a, buf):
            _launch("snf_linear_bwd_weight", _p(g), _p(None), _p(a), N, I, O, ldg, 0, lda, ACT_NONE, _p(buf), _p(None),
                    _stream(), tag=f"{I}x{O}")
        grads.append(None if fused else buf)
    return dx, grads


class _MLP64(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, in_real: int, out_act: int, *ws):
        x = _chk(x, "x")
        assert x.shape[1] >= 32 and x.shape[1] % 4 == 0, "mlp64 input must be padded to >= 32 columns (multiple of 4)"
        need = _TRACK[0] and any(ctx.needs_input_grad)
        y, h1, h2 = _mlp64_fwd_launch(x, in_real, ws, out_act, need)
        if need:
            ctx.save_for_backward(x, y, h1, *((h2,) if h2 is not None else ()))
            ctx.ws, ctx.in_real, ctx.out_act = ws, in_real, out_act
        return y

    @staticmethod
    def backward(ctx, gy):
        saved = ctx.saved_tensors
        x, y, h1 = saved[0], saved[1], saved[2]
        h2 = saved[3] if len(saved) > 3 else None
        gy = _chk(gy, "grad_y")
        dx, grads = _mlp64_bwd_launch(x, ctx.in_real, ctx.ws, ctx.out_act, y, h1, h2, gy, gy.shape[1], 0, None,
                                      ctx.needs_input_grad[0])
        if dx is not None and x.shape[1] != 32:
            full = torch.zeros_like(x)
            full[:, :32] = dx
            dx = full
        return (dx, None, None, *grads)


def mlp64(x, weights: Sequence[torch.Tensor], in_real: int, out_act: int = ACT_NONE) -> torch.Tensor:
    """Fused 64-wide MLP: x [N, >=32 (padded)] -> [N, out].  weights: [64,in_real], ([64,64]), [out,64]."""
    return _apply(_MLP64, x, in_real, out_act, *weights)


# ---------------------------------------------------------------------------------------------
# the whole nerfacto field as ONE autograd node: hash grid -> base MLP -> {trunc_exp density, SH ++ geo -> colour MLP}
# (TCNNNerfactoField.get_density + get_outputs, nerfstudio/fields/nerfacto_field.py:242-351), fused MLP kernels,
# no glue tensors (cat / split / zero-filled slice gradients) in between.
# ---------------------------------------------------------------------------------------------
class _NerfactoField(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, sel, dirs, R: int, S: int, spec, table, bw0, bw1, hw0, hw1, hw2):
        u, dirs, table = _chk(u, "u"), _chk(dirs, "dirs"), _chk(table, "table")
        sc, L, F, T = spec
        N = R * S
        dev = u.device
        need = _TRACK[0] and any(ctx.needs_input_grad)  # (a backward will come: the caller's grad mode and what requires grad)
        # the encoding travels level-major ([L][N][F]) between the grid and the base MLP when the sorted backward can take it
        # (or when there is no backward at all: eval chunks of 4 M samples): whole-line stores in the level-at-a-time grid kernels,
        # and the MLP's d(encoding) IS the backward's staged gradient
        planar = (PLANAR_FIELD_ENCODING and L * F == 32 and F == 2
                  and (not need or (HASHGRID_BWD_MODE == "sorted" and N <= HASHGRID_BWD_MAX_SAMPLES and 8 * L * N < (1 << 32)
                                    and T <= 23)))
        enc = torch.empty((L * F * N,) if planar else (N, L * F), device=dev, dtype=torch.float32)
        if need and table.requires_grad and PRESORT_FIELD_GRID:
            # the field grid's backward sort needs only the positions: it runs NOW on a side stream, beside the forward
            # kernels (the forward phase leaves most of the GPU idle), instead of on the backward's critical chain
            hashgrid_presort(u, sc, L, T, side_stream=True)
        _launch("snf_hashgrid_fwd", _p(u), _p(table), _p(sc), N, L, F, T, _p(enc), 0 if planar else L * F, 0, _stream(),
                tag=f"F{F}L{L}")
        h, hb1, _ = _mlp64_fwd_launch(enc, L * F, (bw0, bw1), ACT_NONE, need, N if planar else 0)
        C = h.shape[1]
        density = torch.empty((N,), device=dev, dtype=torch.float32)
        _launch("snf_trunc_exp_fwd", _p(h), C, _p(sel), N, _p(density), _stream())
        n_geo = C - 1
        x2 = torch.empty((N, 32), device=dev, dtype=torch.float32)
        geo = ctypes.c_void_p(h.data_ptr() + 4)
        _launch("snf_head_input", _p(dirs), geo, R, S, n_geo, C, _p(x2), 32, _stream())
        rgb, hh1, hh2 = _mlp64_fwd_launch(x2, 16 + n_geo, (hw0, hw1, hw2), ACT_SIGMOID, need)
        if need:
            ctx.save_for_backward(u, enc, h, hb1, x2, hh1, hh2, rgb)
            ctx.sel, ctx.spec, ctx.planar = sel, spec, planar
            ctx.params = (table, bw0, bw1, hw0, hw1, hw2)
        return density, rgb

    @staticmethod
    def backward(ctx, g_density, g_rgb):
        u, enc, h, hb1, x2, hh1, hh2, rgb = ctx.saved_tensors
        table, bw0, bw1, hw0, hw1, hw2 = ctx.params
        sc, L, F, T = ctx.spec
        N, C = h.shape
        dev = h.device
        # colour MLP: d rgb -> d(SH ++ geo)
        g_rgb = _chk(g_rgb, "grad_rgb") if g_rgb is not None else torch.zeros_like(rgb)
        dx2, gh = _mlp64_bwd_launch(x2, x2.shape[1] - 1, (hw0, hw1, hw2), ACT_SIGMOID, rgb, hh1, hh2, g_rgb, 3, 0, None, True)
        # density column: through trunc_exp * selector
        graw = torch.empty((N,), device=dev, dtype=torch.float32)
        gd = _chk(g_density, "grad_density") if g_density is not None else torch.zeros((N,), device=dev)
        _launch("snf_trunc_exp_bwd", _p(h), C, _p(ctx.sel), _p(gd), N, _p(graw), 1, _stream())
        # base MLP: dZ[:,0] = graw, dZ[:,1:] = dx2[:, 16:31]  (read in place: column offset 15 of the [N,32] buffer)
        denc, gb = _mlp64_bwd_launch(enc, L * F, (bw0, bw1), ACT_NONE, None, hb1, None, dx2, 32, 15, graw,
                                     table.requires_grad, N if ctx.planar else 0)
        gt = None
        if table.requires_grad:
            buf, fused = _grad_target(table)
            adam = getattr(table, "_fused_adam", None) if fused else None
            if _hashgrid_bwd_launch(u, denc, sc, N, L, F, T, 0 if ctx.planar else 32, 0, buf, adam):
                adam.done = (adam.from_level, L)
            gt = None if fused else buf
        return (None, None, None, None, None, None, gt, gb[0], gb[1], gh[0], gh[1], gh[2])


def nerfacto_field(u, sel, dirs, R: int, S: int, spec, table, base_ws, head_ws):
    """-> (density [R*S], rgb [R*S,3]); base_ws = (W0 [64,32], W1 [16,64]), head_ws = (W0 [64,31], W1 [64,64], W2 [3,64])."""
    return _apply(_NerfactoField, u, sel, dirs, R, S, spec, table, *base_ws, *head_ws)


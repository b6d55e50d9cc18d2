"""Ray data-parallelism over the GPUs of one node: one process per GPU, every rank a full replica drawing its
own rays (seed + rank, samnerf/train.py:87), ONE exchange per step -- the gradient mean.

The reference wraps the model in DDP but calls the unwrapped module, so its all-reduce never runs
(SURVEY.md fact 9); here the mean is real: `dist.all_reduce(SUM)` over each flat gradient arena (backend
"nccl" == RCCL over xGMI on ROCm; "gloo" on CPU for tests) and the 1/world factor folded into the fused Adam
pass.  By default the exchange is sharded (`sharded_step`: reduce-scatter -> Adam on 1/world of the arena -> all-gather
of the parameters), which moves the same bytes over xGMI but divides the HBM-bound optimizer pass by the world size;
`allreduce_gradients` + a replicated Adam remains available (SNF_SHARDED_OPTIMIZER=0).
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, Optional

import torch
import torch.distributed as dist


def _force() -> bool:
    """SNF_FORCE_COLLECTIVES=1: take the collective code paths even at world size 1 (exercises RCCL on a 1-GPU box)."""
    return os.environ.get("SNF_FORCE_COLLECTIVES", "0") == "1"


def _collectives_on() -> bool:
    return dist.is_initialized() and (dist.get_world_size() > 1 or _force())


def env_world() -> tuple:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_distributed(backend: Optional[str] = None) -> tuple:
    """Initialise torch.distributed from the torchrun environment; returns (rank, local_rank, world_size)."""
    rank, local_rank, world = env_world()
    if (world > 1 or (_force() and "RANK" in os.environ)) and not dist.is_initialized():
        # dmabuf IPC only on these hosts (RCCL's `hipIpcGetMemHandle: invalid argument` otherwise).  The HIP runtime reads the variable
        # when it starts (first device call): this is THE place the default is set -- before `set_device` below -- and a process whose
        # runtime is already up without it is told so (the variable then only reaches RCCL's own start-up and child processes).
        if "HSA_ENABLE_IPC_MODE_LEGACY" not in os.environ:
            if torch.cuda.is_available() and torch.cuda.is_initialized():
                import warnings
                warnings.warn("HSA_ENABLE_IPC_MODE_LEGACY was not set when the HIP runtime started: multi-process GPU work on this host "
                              "needs HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment BEFORE the first device call (RCCL otherwise fails "
                              "with `hipIpcGetMemHandle: invalid argument`)", RuntimeWarning, stacklevel=2)
            os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:  # SNF_DIST_BACKEND=gloo: CPU collectives with device tensors staged through the host (tests)
            backend = os.environ.get("SNF_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        import datetime
        # The group timeout is the watchdog of EVERY collective of the job (RCCL) / the per-operation timeout (gloo): it has to
        # outlast a rank-0 checkpoint write, a rank-0-only eval or a first-use build on one rank, so it keeps torch's order of
        # magnitude (30 min).  Only the START-UP is bounded tightly: the first collective runs on a temporary group with its own
        # short timeout, so a rank that never arrives surfaces as an error naming the rendezvous instead of a silent hang.
        timeout = datetime.timedelta(seconds=float(os.environ.get("SNF_DIST_TIMEOUT", "1800")))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timeout)
        first_collective_check(rank, world, backend)
    return rank, local_rank, world


def first_collective_check(rank: int, world: int, backend: str) -> None:
    """One tiny all-reduce right after the rendezvous: the first collective is where a broken fabric / IPC setup shows
    (HSA_ENABLE_IPC_MODE_LEGACY, a rank on the wrong device, a rank that died during start-up).  Fails with the facts a
    user needs instead of hanging in the middle of the first train step."""
    dev = torch.device("cuda", torch.cuda.current_device()) if (backend == "nccl" and torch.cuda.is_available()) else "cpu"
    t = torch.ones((1,), device=dev)
    import datetime
    startup = datetime.timedelta(seconds=float(os.environ.get("SNF_DIST_STARTUP_TIMEOUT", "180")))
    try:
        probe = dist.new_group(ranks=list(range(world)), timeout=startup, backend=backend)  # (collective: every rank makes it)
        dist.all_reduce(t, group=probe)
        if dev != "cpu":
            torch.cuda.synchronize()
        got = float(t.item())
    except Exception as e:  # noqa: BLE001
        raise RuntimeError(
            f"rank {rank}/{world}: the first {backend} collective failed ({type(e).__name__}: {e}).  Check that every rank "
            f"started (torchrun --nproc-per-node {world}), MASTER_ADDR={os.environ.get('MASTER_ADDR')} "
            f"MASTER_PORT={os.environ.get('MASTER_PORT')} are reachable, one rank per GPU, and "
            f"HSA_ENABLE_IPC_MODE_LEGACY=0 (current: {os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')})") from e
    if got != float(world):
        raise RuntimeError(f"rank {rank}/{world}: first all-reduce returned {got}, expected {world} (ranks missing or doubled)")
    # The DEFAULT group's communicator is created lazily by its first collective, under the 30-minute job timeout: create it now,
    # right behind the probe (every rank is known to be alive at this point -- the probe all-reduce just completed), so a failure
    # specific to that initialisation surfaces at start-up with the facts below instead of inside the first train step.
    try:
        t.fill_(1.0)
        dist.all_reduce(t)
        if dev != "cpu":
            torch.cuda.synchronize()
        got = float(t.item())
    except Exception as e:  # noqa: BLE001
        raise RuntimeError(f"rank {rank}/{world}: the first collective on the default {backend} group failed "
                           f"({type(e).__name__}: {e}) after the start-up probe on a temporary group had succeeded") from e
    if got != float(world):
        raise RuntimeError(f"rank {rank}/{world}: first all-reduce on the default group returned {got}, expected {world}")
    # (the probe group stays: it is idle from here on -- one extra communicator's buffers -- because tearing a communicator down is
    #  itself a collective on RCCL, and a rank that stalls in it would hang the job this check exists to protect)


# -- collectives.  backend "nccl" (= RCCL): straight through.  backend "gloo" with device tensors (tests: two ranks sharing
# one GPU, which RCCL refuses) is staged through host memory, so the same code paths run with the real kernels.
def _staged(t: torch.Tensor) -> bool:
    return t.is_cuda and dist.get_backend() == "gloo"


def _all_reduce(t: torch.Tensor) -> None:
    if _staged(t):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)


def _all_reduce_max(t: torch.Tensor) -> None:
    """MAX over the ranks, in place (the veto word of a step guard: one rank's non-finite loss vetoes the step everywhere)."""
    if _staged(t):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MAX)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)


def _broadcast(t: torch.Tensor, src: int) -> None:
    if _staged(t):
        h = t.cpu()
        dist.broadcast(h, src=src)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src)


def _all_gather_into(out: torch.Tensor, inp: torch.Tensor) -> None:
    if _staged(out):
        h = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(h, inp.cpu())
        out.copy_(h)
    else:
        dist.all_gather_into_tensor(out, inp)


def _all_to_all(out: torch.Tensor, inp: torch.Tensor) -> None:
    if _staged(out):
        h = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(h, inp.cpu())
        out.copy_(h)
    else:
        dist.all_to_all_single(out, inp)


def allreduce_gradients(grad_buffers: Iterable[torch.Tensor], async_op: bool = False):
    """SUM-all-reduce every flat gradient buffer in place; the caller divides by world size in the Adam pass."""
    if not _collectives_on():
        return []
    handles = []
    for g in grad_buffers:
        if async_op:
            handles.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True))
        else:
            _all_reduce(g)
    return handles


SHARD_ALIGN = 64  # floats: shard boundaries stay on 256-B lines


def shard_bounds(n: int, world: int, rank: int) -> tuple:
    """Split [0, n) into `world` equal 256-B-aligned shards plus a remainder: -> (chunk, bulk, lo, hi) with
    bulk = world * chunk <= n, this rank's shard [lo, hi) = [rank*chunk, (rank+1)*chunk); [bulk, n) is the remainder
    every rank keeps replicated (fewer than world * 64 elements)."""
    chunk = (n // world) // SHARD_ALIGN * SHARD_ALIGN
    return chunk, chunk * world, rank * chunk, (rank + 1) * chunk


def shard_reduce(grad: torch.Tensor, lo: int, hi: int, bulk: int) -> None:
    """First half of `sharded_step`: the SUM over the ranks of grad[:bulk] lands in this rank's shard grad[lo:hi]."""
    if dist.get_backend() == "gloo":  # no reduce_scatter in gloo (CPU tests / ranks sharing a GPU): all-reduce, use the own shard
        _all_reduce(grad[:bulk])
    else:
        dist.reduce_scatter_tensor(grad[lo:hi], grad[:bulk], op=dist.ReduceOp.SUM)


def shard_gather(param: torch.Tensor, lo: int, hi: int, bulk: int) -> None:
    """Second half: every rank's updated shard param[lo:hi] into param[:bulk] on all ranks (in place on RCCL: the input is this
    rank's slot of the output; gloo needs a separate input buffer)."""
    _all_gather_into(param[:bulk], param[lo:hi].clone() if dist.get_backend() == "gloo" else param[lo:hi])


def sharded_step(param: torch.Tensor, grad: torch.Tensor, step_fn) -> None:
    """The exchange step of ray data-parallel training, ZeRO-1 style, on one flat (param, grad) slice pair:

        reduce-scatter(SUM) the gradient -> each rank runs the optimizer on its 1/world shard only
        (`step_fn(lo, hi)`, which must also re-zero grad[lo:hi]) -> all-gather the updated parameters.

    Same bytes on the xGMI links as the all-reduce it replaces (an all-reduce IS reduce-scatter + all-gather), but the
    HBM-bound Adam pass and its state traffic shrink by the world size.  The small unaligned remainder is all-reduced
    and stepped redundantly on every rank.  Gradients outside the own shard are zeroed here.  With world size 1 this is
    just step_fn(0, n).  The caller folds 1/world into step_fn (gradients arrive SUMMED)."""
    n = param.numel()
    if not _collectives_on():
        step_fn(0, n)
        return
    world, rank = dist.get_world_size(), dist.get_rank()
    chunk, bulk, lo, hi = shard_bounds(n, world, rank)
    if chunk > 0:
        shard_reduce(grad, lo, hi, bulk)
        step_fn(lo, hi)
        if lo > 0:
            grad[:lo].zero_()
        if hi < bulk:
            grad[hi:bulk].zero_()
        shard_gather(param, lo, hi, bulk)
    if bulk < n:
        _all_reduce(grad[bulk:])
        step_fn(bulk, n)


def collectives_on() -> bool:
    return _collectives_on()


def exchange_rows(grad_rows: torch.Tensor, row_index: torch.Tensor) -> None:
    """SUM-all-reduce the listed rows of a [n_rows, F] gradient view only (the reachable rows of a coarse hash level:
    every other row of that level is zero on every rank and stays zero): pack -> all_reduce -> unpack."""
    if not _collectives_on() or row_index.numel() == 0:
        return
    packed = grad_rows.index_select(0, row_index)
    _all_reduce(packed)
    grad_rows.index_copy_(0, row_index, packed)


def split_range(n: int, granule: int = 1) -> tuple:
    """This rank's contiguous share [lo, hi) of n items, in whole granules (eval: rays of one image, patches kept whole)."""
    if not _collectives_on():
        return 0, n
    world, rank = dist.get_world_size(), dist.get_rank()
    units = (n + granule - 1) // granule
    lo_u, hi_u = units * rank // world, units * (rank + 1) // world
    return min(lo_u * granule, n), min(hi_u * granule, n)


def all_gather_rows(local: torch.Tensor) -> torch.Tensor:
    """Concatenate every rank's [n_r, ...] rows in rank order (n_r may differ): sizes are exchanged, rows padded to the
    longest share for the collective and trimmed afterwards."""
    if not _collectives_on():
        return local
    world = dist.get_world_size()
    sizes = torch.zeros((world,), dtype=torch.int64, device=local.device)
    mine = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    _all_gather_into(sizes, mine)
    sizes = sizes.tolist()
    m = max(max(sizes), 1)
    padded = torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    out = torch.empty((world * m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    _all_gather_into(out, padded)
    return torch.cat([out[r * m:r * m + sizes[r]] for r in range(world)], dim=0)


def gather_sharded_state(buf: torch.Tensor) -> None:
    """Make a sharded optimizer-state buffer whole on every rank (before a checkpoint)."""
    if not _collectives_on():
        return
    world, rank = dist.get_world_size(), dist.get_rank()
    chunk, bulk, lo, hi = shard_bounds(buf.numel(), world, rank)
    if chunk > 0:
        _all_gather_into(buf[:bulk], buf[lo:hi].clone())


def broadcast_parameters(param_buffers: Iterable[torch.Tensor], src: int = 0) -> None:
    """Make every replica start from rank 0's parameters (DDP's constructor-time broadcast)."""
    if not _collectives_on():
        return
    for p in param_buffers:
        _broadcast(p, src)


def world_size() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


# ---------------------------------------------------------------------------------------------------------------------
# Table parallelism for the feature hash grids
# ---------------------------------------------------------------------------------------------------------------------
# The four F=8 feature tables are 805 MB -- 92 % of all parameters -- while a step only reads/writes N*L*8 rows of them.
# Replicating them (plain ray data-parallelism) puts 2 x 7/8 x 805 MB per GPU on the xGMI links every step; owning them
# costs N x 768 B per head and direction.  So each rank OWNS a contiguous run of (grid, level) slabs of every head, evaluates
# those levels for the samples of ALL ranks, and the activations travel instead of the tables:
#
#   forward   all-gather positions [N,3] -> own levels at W*N points -> all-to-all -> [N, L_total*F] on the sample's rank
#   backward  all-to-all of d(features)  -> sorted scatter-add of W*N samples into the OWN slabs only
#   step      Adam on the own slabs, no gradient exchange and no parameter all-gather for the tables
#
# The sums are the ones the all-reduce would have formed (every sample's contribution to every row, then 1/W in Adam), in a
# different fp32 order.  The dense layers, the field / proposal grids (69 MB) and the conv head stay replicated
# (`sharded_step`).  Tables are made whole again (`broadcast` of each owner's slabs) before evaluation and checkpoints.


class TableParallelLayout:
    """Slab ownership for one head = a tuple of hash grids evaluated at the same points (each (levels, F, log2_T)).

    Slabs are numbered grid-major / level-minor -- the order of the head's output columns -- and rank r owns slabs
    [r*per, (r+1)*per).  `runs(r)`: the owned slabs as (grid, first_level, n_levels, local_column) runs, one kernel launch
    each; a rank's levels inside one grid are always contiguous."""

    def __init__(self, grids, world: int):
        self.grids = [(int(L), int(F), int(T)) for (L, F, T) in grids]
        self.world = int(world)
        self.F = self.grids[0][1]
        self.n_slabs = sum(L for L, _, _ in self.grids)
        self.per = self.n_slabs // self.world
        self.width = self.per * self.F            # columns a rank produces
        self.total = self.n_slabs * self.F        # columns of the head

    @staticmethod
    def supported(grids, world: int) -> bool:
        grids = list(grids)
        return (world >= 1 and len(grids) > 0 and len({F for _, F, _ in grids}) == 1
                and sum(L for L, _, _ in grids) % world == 0)

    def runs(self, rank: int):
        lo, hi = rank * self.per, (rank + 1) * self.per
        out, base = [], 0
        for gi, (L, _, _) in enumerate(self.grids):
            a, b = max(lo, base), min(hi, base + L)
            if b > a:
                out.append((gi, a - base, b - a, (a - lo) * self.F))
            base += L
        return out

    def owned_levels(self, rank: int, grid: int):
        """(first_level, end_level) of `grid` owned by `rank` ((0, 0) if none)."""
        for gi, l0, nl, _ in self.runs(rank):
            if gi == grid:
                return l0, l0 + nl
        return 0, 0

    def owned_elements(self, rank: int, grid: int):
        """The same as a flat element range of that grid's [L, 2^T, F] table."""
        l0, l1 = self.owned_levels(rank, grid)
        _, F, T = self.grids[grid]
        return (l0 << T) * F, (l1 << T) * F


def tp_gather_positions(u: torch.Tensor) -> torch.Tensor:
    """[N, 3] on every rank -> [W*N, 3] in rank order (N must be the same on every rank: rays x top-K)."""
    world = dist.get_world_size()
    out = torch.empty((world * u.shape[0], u.shape[1]), dtype=u.dtype, device=u.device)
    _all_gather_into(out, u.contiguous())
    return out


def tp_exchange(block: torch.Tensor) -> torch.Tensor:
    """All-to-all of a [W, N, C] block: slice w goes to rank w; the result's slice w came from rank w."""
    out = torch.empty_like(block)
    _all_to_all(out.view(-1), block.contiguous().view(-1))
    return out


def tp_forward(u_all: torch.Tensor, n_local: int, layout: TableParallelLayout, eval_run) -> torch.Tensor:
    """Features of this rank's n_local samples from all owners.  eval_run(grid, first_level, n_levels, out, ld, col) writes
    the levels of one run for every row of u_all into out[:, col : col + n_levels*F]."""
    world, rank = layout.world, dist.get_rank()
    mine = torch.empty((world * n_local, layout.width), dtype=torch.float32, device=u_all.device)
    for gi, l0, nl, col in layout.runs(rank):
        eval_run(gi, l0, nl, mine, layout.width, col)
    got = tp_exchange(mine.view(world, n_local, layout.width))          # [owner, n, per*F]
    return got.permute(1, 0, 2).reshape(n_local, layout.total)          # owner-major == slab order == column order


def tp_backward(grad_out: torch.Tensor, layout: TableParallelLayout) -> torch.Tensor:
    """d(features) [n_local, total] of this rank's samples -> [W*n_local, per*F]: the gradients of the columns this rank owns,
    for the samples of every rank (rank order, matching tp_gather_positions)."""
    world, n_local = layout.world, grad_out.shape[0]
    send = grad_out.view(n_local, world, layout.width).permute(1, 0, 2).contiguous()
    return tp_exchange(send).view(world * n_local, layout.width)


def tp_refresh_table(param_flat: torch.Tensor, layout: TableParallelLayout, grid: int) -> None:
    """Make one table whole on every rank: each owner broadcasts its levels."""
    for w in range(layout.world):
        lo, hi = layout.owned_elements(w, grid)
        if hi > lo:
            _broadcast(param_flat[lo:hi], w)

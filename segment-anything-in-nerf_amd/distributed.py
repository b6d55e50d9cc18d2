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
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def allreduce_gradients(grad_buffers: Iterable[torch.Tensor], async_op: bool = False):
    """SUM-all-reduce every flat gradient buffer in place; the caller divides by world size in the Adam pass."""
    if not _collectives_on():
        return []
    handles = []
    for g in grad_buffers:
        h = dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=async_op)
        if async_op:
            handles.append(h)
    return handles


SHARD_ALIGN = 64  # floats: shard boundaries stay on 256-B lines


def shard_bounds(n: int, world: int, rank: int) -> tuple:
    """Split [0, n) into `world` equal 256-B-aligned shards plus a remainder: -> (chunk, bulk, lo, hi) with
    bulk = world * chunk <= n, this rank's shard [lo, hi) = [rank*chunk, (rank+1)*chunk); [bulk, n) is the remainder
    every rank keeps replicated (fewer than world * 64 elements)."""
    chunk = (n // world) // SHARD_ALIGN * SHARD_ALIGN
    return chunk, chunk * world, rank * chunk, (rank + 1) * chunk


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
        if dist.get_backend() == "gloo":  # no reduce_scatter in gloo (CPU tests): all-reduce, then use the own shard
            dist.all_reduce(grad[:bulk], op=dist.ReduceOp.SUM)
        else:
            dist.reduce_scatter_tensor(grad[lo:hi], grad[:bulk], op=dist.ReduceOp.SUM)
        step_fn(lo, hi)
        if lo > 0:
            grad[:lo].zero_()
        if hi < bulk:
            grad[hi:bulk].zero_()
        # in place (input = this rank's slot of the output) on RCCL; gloo needs a separate input buffer
        dist.all_gather_into_tensor(param[:bulk], param[lo:hi].clone() if dist.get_backend() == "gloo" else param[lo:hi])
    if bulk < n:
        dist.all_reduce(grad[bulk:], op=dist.ReduceOp.SUM)
        step_fn(bulk, n)


def collectives_on() -> bool:
    return _collectives_on()


def exchange_rows(grad_rows: torch.Tensor, row_index: torch.Tensor) -> None:
    """SUM-all-reduce the listed rows of a [n_rows, F] gradient view only (the reachable rows of a coarse hash level:
    every other row of that level is zero on every rank and stays zero): pack -> all_reduce -> unpack."""
    if not _collectives_on() or row_index.numel() == 0:
        return
    packed = grad_rows.index_select(0, row_index)
    dist.all_reduce(packed, op=dist.ReduceOp.SUM)
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
    dist.all_gather_into_tensor(sizes, mine)
    sizes = sizes.tolist()
    m = max(max(sizes), 1)
    padded = torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    out = torch.empty((world * m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded)
    return torch.cat([out[r * m:r * m + sizes[r]] for r in range(world)], dim=0)


def gather_sharded_state(buf: torch.Tensor) -> None:
    """Make a sharded optimizer-state buffer whole on every rank (before a checkpoint)."""
    if not _collectives_on():
        return
    world, rank = dist.get_world_size(), dist.get_rank()
    chunk, bulk, lo, hi = shard_bounds(buf.numel(), world, rank)
    if chunk > 0:
        dist.all_gather_into_tensor(buf[:bulk], buf[lo:hi].clone())


def broadcast_parameters(param_buffers: Iterable[torch.Tensor], src: int = 0) -> None:
    """Make every replica start from rank 0's parameters (DDP's constructor-time broadcast)."""
    if not _collectives_on():
        return
    for p in param_buffers:
        dist.broadcast(p, src=src)


def world_size() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1

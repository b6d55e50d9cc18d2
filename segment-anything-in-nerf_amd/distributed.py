"""Ray data-parallelism over the GPUs of one node: one process per GPU, every rank a full replica drawing its
own rays (seed + rank, samnerf/train.py:87), ONE exchange per step -- the gradient mean.

The reference wraps the model in DDP but calls the unwrapped module, so its all-reduce never runs
(SURVEY.md fact 9); here the mean is real: `dist.all_reduce(SUM)` over each flat gradient arena (backend
"nccl" == RCCL over xGMI on ROCm; "gloo" on CPU for tests) and the 1/world factor folded into the fused Adam
pass.  Four collectives per step (one per parameter group), the largest being the ~0.8 GB `sam_field` arena.
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, Optional

import torch
import torch.distributed as dist


def env_world() -> tuple:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_distributed(backend: Optional[str] = None) -> tuple:
    """Initialise torch.distributed from the torchrun environment; returns (rank, local_rank, world_size)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
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
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return []
    handles = []
    for g in grad_buffers:
        h = dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=async_op)
        if async_op:
            handles.append(h)
    return handles


def broadcast_parameters(param_buffers: Iterable[torch.Tensor], src: int = 0) -> None:
    """Make every replica start from rank 0's parameters (DDP's constructor-time broadcast)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for p in param_buffers:
        dist.broadcast(p, src=src)


def world_size() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1

"""Batch builder kernels (csrc/batch.hip): no gradients flow through these (split out of ops.py; `samnerf_amd.ops` re-exports everything here)."""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

import torch

from ._opcore import (ACT_BY_NAME, ACT_GELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, CONTRACT_L2, CONTRACT_LINF, CONTRACT_NONE, _L, _chk,
                      _launch, _linear_fwd_ws, _p, _stream)

# ---------------------------------------------------------------------------------------------
# batch builder (csrc/batch.hip) -- no gradients flow through these
# ---------------------------------------------------------------------------------------------
@torch.no_grad()
def pixel_indices(u, batch_size: int, patch: int, num_images: int, H: int, W: int) -> torch.Tensor:
    """u [batch_size / patch^2, 3] ~ U[0,1) -> (camera, row, col) int64 [batch_size, 3]."""
    u = _chk(u, "u")
    assert u.shape == (batch_size // (patch * patch), 3)
    out = torch.empty((batch_size, 3), device=u.device, dtype=torch.int64)
    _launch("snf_pixel_indices", _p(u), batch_size, patch, num_images, H, W, _p(out), _stream())
    return out


@torch.no_grad()
def generate_rays(indices, c2w, intrinsics):
    """indices [R,3] int64, c2w [N,3,4], intrinsics [N,4] -> origins [R,3], directions [R,3], pixel_area [R,1],
    camera_indices [R,1] int64."""
    assert indices.is_cuda and indices.dtype == torch.int64 and indices.is_contiguous()
    c2w, intrinsics = _chk(c2w, "c2w"), _chk(intrinsics, "intrinsics")
    R, N, dev = indices.shape[0], c2w.shape[0], indices.device
    o = torch.empty((R, 3), device=dev, dtype=torch.float32)
    d = torch.empty((R, 3), device=dev, dtype=torch.float32)
    pa = torch.empty((R, 1), device=dev, dtype=torch.float32)
    ci = torch.empty((R, 1), device=dev, dtype=torch.int64)
    _launch("snf_generate_rays", _p(indices), R, _p(c2w), _p(intrinsics), N, _p(o), _p(d), _p(pa), _p(ci), _stream())
    return o, d, pa, ci


@torch.no_grad()
def gather_nearest(points, features, image_shape, point_stride: int = 1, point_offset: int = 0) -> torch.Tensor:
    """features[cam, long(row * fh/H), long(col * fw/W)] for every point_stride-th point of `points` [B,3] int64."""
    assert points.is_cuda and points.dtype == torch.int64 and points.is_contiguous()
    features = _chk(features, "features")
    N, fh, fw, C = features.shape
    B = points.shape[0] // point_stride
    out = torch.empty((B, C), device=points.device, dtype=torch.float32)
    _launch("snf_gather_nearest", _p(points), B, point_stride, point_offset, _p(features), N, fh, fw, C, int(image_shape[0]),
            int(image_shape[1]), _p(out), _stream())
    return out

"""SceneContraction with the reference's constructor (field_components/spatial_distortions.py:42-88).

In the product path the contraction is fused into the positions kernel (`ops.positions`); the object mostly
carries the norm order.  Calling it on a tensor of world positions runs the same kernel."""
from __future__ import annotations

from typing import Optional, Union

import torch

from . import ops


class SpatialDistortion(torch.nn.Module):
    def forward(self, positions):  # pragma: no cover - interface
        raise NotImplementedError


class SceneContraction(SpatialDistortion):
    def __init__(self, order: Optional[Union[float, int]] = None) -> None:
        super().__init__()
        if order not in (None, 2, float("inf")):
            raise ValueError("SceneContraction: the MI355X kernels implement order None/2 (L2) and inf")
        self.order = order

    @property
    def kernel_mode(self) -> int:
        return ops.CONTRACT_LINF if self.order == float("inf") else ops.CONTRACT_L2

    def forward(self, positions: torch.Tensor) -> torch.Tensor:
        flat = positions.reshape(-1, 3).contiguous()
        n = flat.shape[0]
        zeros = torch.zeros_like(flat)
        eb = torch.zeros((n, 2), device=flat.device, dtype=torch.float32)
        u, _ = ops.positions(flat, zeros, eb, None, self.kernel_mode, False)
        return (u * 4.0 - 2.0).view(positions.shape)


def contraction_mode(distortion: Optional[SpatialDistortion]) -> int:
    if distortion is None:
        return ops.CONTRACT_NONE
    return distortion.kernel_mode

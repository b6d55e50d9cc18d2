"""Losses with the reference's names (nerfstudio/model_components/losses.py:46-143)."""
from __future__ import annotations

import torch

from . import ops


def ray_samples_to_sdist(ray_samples) -> torch.Tensor:
    return ray_samples.spacing_bins


def interlevel_loss(weights_list, ray_samples_list) -> torch.Tensor:
    c = ray_samples_to_sdist(ray_samples_list[-1]).detach()
    w = weights_list[-1][..., 0].detach()
    loss = 0.0
    for ray_samples, weights in zip(ray_samples_list[:-1], weights_list[:-1]):
        loss = loss + ops.interlevel_loss(weights[..., 0], ray_samples_to_sdist(ray_samples), c, w)
    return loss


def distortion_loss(weights_list, ray_samples_list) -> torch.Tensor:
    return ops.distortion_loss(weights_list[-1][..., 0], ray_samples_to_sdist(ray_samples_list[-1]))


class MSELoss(torch.nn.Module):
    """nn.MSELoss() (nerfacto.py:226: `self.rgb_loss = MSELoss()`), one HIP launch per direction on the GPU."""

    def forward(self, target: torch.Tensor, pred: torch.Tensor) -> torch.Tensor:
        # the reference calls rgb_loss(image, outputs["rgb"]); MSE is symmetric, the gradient goes to whichever needs it
        if pred.requires_grad or not target.requires_grad:
            return ops.mse_loss(pred, target)
        return ops.mse_loss(target, pred)

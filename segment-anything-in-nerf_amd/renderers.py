"""Renderers with the reference's names (nerfstudio/model_components/renderers.py:58-140,197-270;
samnerf/sam_model.py:126-137)."""
from __future__ import annotations

import torch
from torch import nn

from . import ops
from .rays import RaySamples


class RGBRenderer(nn.Module):
    def __init__(self, background_color="last_sample") -> None:
        super().__init__()
        if background_color != "last_sample":
            raise NotImplementedError("nerfacto renders with background_color='last_sample' (nerfacto.py:77,220)")
        self.background_color = background_color

    def forward(self, rgb: torch.Tensor, weights: torch.Tensor, ray_indices=None, num_rays=None) -> torch.Tensor:
        w = weights[..., 0] if weights.dim() == 3 else weights
        return ops.composite_rgb(rgb, w, self.training)


class _Accumulation(torch.autograd.Function):
    """sum_s w[r, s] from the compositing kernel; d(accumulation)/d(w) = 1, so the backward broadcasts the ray's gradient."""

    @staticmethod
    def forward(ctx, w):
        ctx.shape = tuple(w.shape)
        return ops.accumulation(w)

    @staticmethod
    def backward(ctx, g):
        return g.expand(ctx.shape).contiguous()


class AccumulationRenderer(nn.Module):
    """renderers.py:200-223: sum of the weights along a ray = the accumulation output of the compositing kernel
    (snf_composite_fwd).  Nothing on this path differentiates through it (it feeds the viewer / metrics only), so by default the
    result is detached; `differentiable=True` keeps it in the graph (the kernel forward, a broadcast backward)."""

    @classmethod
    def forward(cls, weights: torch.Tensor, ray_indices=None, num_rays=None, differentiable: bool = False) -> torch.Tensor:
        w = weights[..., 0] if weights.dim() == 3 else weights
        if differentiable:
            return _Accumulation.apply(w)
        return ops.accumulation(w.detach())


class DepthRenderer(nn.Module):
    def __init__(self, method: str = "median") -> None:
        super().__init__()
        if method != "median":
            raise NotImplementedError("nerfacto uses the median depth (renderers.py:241)")
        self.method = method

    def forward(self, weights: torch.Tensor, ray_samples: RaySamples, ray_indices=None, num_rays=None) -> torch.Tensor:
        w = weights[..., 0] if weights.dim() == 3 else weights
        depth, _ = ops.render_depth_acc(w.detach(), ray_samples.euclid_bins, want_acc=False)
        return depth


class MeanRenderer(nn.Module):
    """samnerf/sam_model.py:126-137: sum_k w_k * embeds_k (weights enter as constants at every call site)."""

    @classmethod
    def forward(cls, embeds: torch.Tensor, weights: torch.Tensor) -> torch.Tensor:
        R, K, C = embeds.shape
        return ops.feature_mean(embeds.reshape(R * K, C), weights.reshape(R, K), R, K)

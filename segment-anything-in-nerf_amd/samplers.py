"""Ray samplers with the reference's class names and call signatures
(nerfstudio/model_components/ray_samplers.py:54-126,223-369,509-599), running on the sampling kernels.

Randomness: the reference draws one `torch.rand((R,1))` per ray in the initial sampler and one in the PDF sampler
(ray_samplers.py:105-106,318-319).  Here the jitter tensors are drawn by torch on the device and PASSED to the
kernels, so a test (or a parity run) can inject the exact values through `jitter_override`.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
from torch import nn

from . import ops
from .rays import RayBundle, RaySamples


class Sampler(nn.Module):
    def __init__(self, num_samples: Optional[int] = None) -> None:
        super().__init__()
        self.num_samples = num_samples
        self.jitter_override: Optional[torch.Tensor] = None

    def generate_ray_samples(self, *args, **kwargs) -> RaySamples:  # pragma: no cover - interface
        raise NotImplementedError

    def forward(self, *args, **kwargs):
        return self.generate_ray_samples(*args, **kwargs)

    def _jitter(self, num_rays: int, device) -> torch.Tensor:
        if self.jitter_override is not None:
            return self.jitter_override.to(device=device, dtype=torch.float32).reshape(-1, 1)
        return torch.rand((num_rays, 1), dtype=torch.float32, device=device)


class UniformLinDispPiecewiseSampler(Sampler):
    """ray_samplers.py:223-246 (SpacedSampler.generate_ray_samples :79-126 with the piecewise spacing fn)."""

    def __init__(self, num_samples: Optional[int] = None, train_stratified=True, single_jitter=False) -> None:
        super().__init__(num_samples=num_samples)
        if not single_jitter:
            raise NotImplementedError("the hot path uses single_jitter=True (nerfacto.py:128,211)")
        self.train_stratified = train_stratified
        self.single_jitter = single_jitter

    def generate_ray_samples(self, ray_bundle: Optional[RayBundle] = None, num_samples: Optional[int] = None) -> RaySamples:
        assert ray_bundle is not None and ray_bundle.nears is not None and ray_bundle.fars is not None
        num_samples = num_samples or self.num_samples
        assert num_samples is not None
        t = self._jitter(len(ray_bundle), ray_bundle.origins.device) if (self.train_stratified and self.training) else None
        sb, eb = ops.sample_spacing(ray_bundle.nears, ray_bundle.fars, num_samples, t)
        return RaySamples(ray_bundle, eb, sb)


class PDFSampler(Sampler):
    """ray_samplers.py:249-369, include_original=False path."""

    def __init__(self, num_samples: Optional[int] = None, train_stratified: bool = True, single_jitter: bool = False,
                 include_original: bool = True, histogram_padding: float = 0.01) -> None:
        super().__init__(num_samples=num_samples)
        if include_original or not single_jitter:
            raise NotImplementedError("the hot path uses include_original=False, single_jitter=True (ray_samplers.py:543)")
        self.train_stratified = train_stratified
        self.include_original = include_original
        self.histogram_padding = histogram_padding
        self.single_jitter = single_jitter

    def generate_ray_samples(self, ray_bundle: Optional[RayBundle] = None, ray_samples: Optional[RaySamples] = None,
                             weights: torch.Tensor = None, num_samples: Optional[int] = None, eps: float = 1e-5,
                             anneal: float = 1.0) -> RaySamples:
        if ray_samples is None or ray_bundle is None:
            raise ValueError("ray_samples and ray_bundle must be provided")
        num_samples = num_samples or self.num_samples
        assert num_samples is not None and ray_samples.spacing_bins is not None
        w = weights[..., 0] if weights.dim() == 3 else weights
        u = self._jitter(len(ray_bundle), w.device) if (self.train_stratified and self.training) else None
        sb, eb = ops.pdf_resample(w.detach(), ray_samples.spacing_bins, ray_bundle.nears, ray_bundle.fars, num_samples, u,
                                  anneal=anneal, histogram_padding=self.histogram_padding)
        return RaySamples(ray_bundle, eb, sb)


class ProposalNetworkSampler(Sampler):
    """ray_samplers.py:509-599: initial sampler -> density_fn -> weights -> anneal -> PDF, `num_proposal_network_iterations`
    times; `updated` gates whether the proposal density runs with gradients."""

    def __init__(self, num_proposal_samples_per_ray: Tuple[int, ...] = (64,), num_nerf_samples_per_ray: int = 32,
                 num_proposal_network_iterations: int = 2, single_jitter: bool = False,
                 update_sched: Callable = lambda x: 1, initial_sampler: Optional[Sampler] = None) -> None:
        super().__init__()
        self.num_proposal_samples_per_ray = num_proposal_samples_per_ray
        self.num_nerf_samples_per_ray = num_nerf_samples_per_ray
        self.num_proposal_network_iterations = num_proposal_network_iterations
        self.update_sched = update_sched
        if self.num_proposal_network_iterations < 1:
            raise ValueError("num_proposal_network_iterations must be >= 1")
        self.initial_sampler = initial_sampler or UniformLinDispPiecewiseSampler(single_jitter=single_jitter)
        self.pdf_sampler = PDFSampler(include_original=False, single_jitter=single_jitter)
        self._anneal = 1.0
        self._steps_since_update = 0
        self._step = 0
        self.last_updated = True

    def set_anneal(self, anneal: float) -> None:
        self._anneal = anneal

    def step_cb(self, step):
        self._step = step
        self._steps_since_update += 1

    def generate_ray_samples(self, ray_bundle: Optional[RayBundle] = None,
                             density_fns: Optional[List[Callable]] = None) -> Tuple[RaySamples, List, List]:
        assert ray_bundle is not None and density_fns is not None
        weights_list, ray_samples_list = [], []
        n = self.num_proposal_network_iterations
        weights, ray_samples = None, None
        updated = self._steps_since_update > self.update_sched(self._step) or self._step < 10
        if torch.is_grad_enabled():
            # remembered for the trainer: on a non-update step the proposal parameters have NO gradient in the reference
            # (grad is None), so torch.optim.Adam skips them entirely -- no moment decay, no step count
            self.last_updated = bool(updated)
        for i_level in range(n + 1):
            is_prop = i_level < n
            num_samples = self.num_proposal_samples_per_ray[i_level] if is_prop else self.num_nerf_samples_per_ray
            if i_level == 0:
                ray_samples = self.initial_sampler(ray_bundle, num_samples=num_samples)
            else:
                assert weights is not None
                # anneal pow (ray_samplers.py:583) is folded into the PDF kernel
                ray_samples = self.pdf_sampler(ray_bundle, ray_samples, weights, num_samples=num_samples,
                                               anneal=self._anneal)
            if is_prop:
                if updated:
                    density = density_fns[i_level](ray_samples)
                else:
                    with torch.no_grad():
                        density = density_fns[i_level](ray_samples)
                weights = ray_samples.get_weights(density)
                weights_list.append(weights)
                ray_samples_list.append(ray_samples)
        if updated:
            self._steps_since_update = 0
        assert ray_samples is not None
        return ray_samples, weights_list, ray_samples_list

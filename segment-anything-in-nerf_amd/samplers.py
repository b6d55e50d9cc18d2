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

    def _proposal_round_gets_gradients(self) -> bool:
        """The `updated` gate of ray_samplers.py:566: proposal densities are differentiated in the first ten steps and then
        whenever more steps than `update_sched(step)` have passed since the last differentiated round."""
        return self._step < 10 or self._steps_since_update > self.update_sched(self._step)

    def generate_ray_samples(self, ray_bundle: Optional[RayBundle] = None,
                             density_fns: Optional[List[Callable]] = None) -> Tuple[RaySamples, List, List]:
        """-> (fine samples, [weights of every proposal round], [samples of every proposal round]).  Round 0 draws from the
        piecewise-uniform initial sampler; every later round -- and the final, fine draw -- resamples the previous round's weight
        histogram (the anneal exponent of ray_samplers.py:583 is applied inside the PDF kernel)."""
        if ray_bundle is None or density_fns is None:
            raise ValueError("ray_bundle and density_fns must be provided")
        rounds = self.num_proposal_network_iterations
        with_grad = self._proposal_round_gets_gradients()
        if torch.is_grad_enabled():
            # remembered for the trainer: on a round without gradients the proposal parameters have NO gradient in the
            # reference (grad is None / zero-filled, see Trainer.zero_grad_adam)
            self.last_updated = bool(with_grad)
        history_w: List[torch.Tensor] = []
        history_s: List[RaySamples] = []
        samples = self.initial_sampler(ray_bundle, num_samples=self.num_proposal_samples_per_ray[0])
        for r in range(rounds):
            with torch.enable_grad() if (with_grad and torch.is_grad_enabled()) else torch.no_grad():
                density = density_fns[r](samples)
            w = samples.get_weights(density)
            history_w.append(w)
            history_s.append(samples)
            n_next = self.num_proposal_samples_per_ray[r + 1] if r + 1 < rounds else self.num_nerf_samples_per_ray
            samples = self.pdf_sampler(ray_bundle, samples, w, num_samples=n_next, anneal=self._anneal)
        if with_grad:
            self._steps_since_update = 0
        return samples, history_w, history_s

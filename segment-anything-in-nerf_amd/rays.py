"""RayBundle / Frustums / RaySamples with the reference's field names (nerfstudio/cameras/rays.py:31-270).

Storage differs from the reference on purpose: a RaySamples keeps the compact per-ray bin edges
(`euclid_bins`, `spacing_bins`, both [R, n+1]) that the HIP kernels consume, and materialises the
reference-shaped views (`frustums.starts` [R,n,1], `deltas`, `spacing_starts` ...) only when asked.
A top-K gathered RaySamples (samnerf/sam_model.py:250-255) is the same object plus `ids` [R,K].
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, Optional

import torch

from . import ops


@dataclass
class RayBundle:
    origins: torch.Tensor  # [R,3]
    directions: torch.Tensor  # [R,3] unit
    pixel_area: torch.Tensor  # [R,1]
    camera_indices: Optional[torch.Tensor] = None  # [R,1] long
    nears: Optional[torch.Tensor] = None  # [R,1]
    fars: Optional[torch.Tensor] = None  # [R,1]
    metadata: Optional[Dict[str, torch.Tensor]] = None
    times: Optional[torch.Tensor] = None

    def __len__(self) -> int:
        return self.origins.numel() // self.origins.shape[-1]

    @property
    def shape(self):
        return self.origins.shape[:-1]

    def set_camera_indices(self, camera_index: int) -> None:
        self.camera_indices = torch.ones_like(self.origins[..., 0:1]).long() * camera_index

    def _map(self, fn) -> "RayBundle":
        f = lambda t: None if t is None else fn(t)  # noqa: E731
        return RayBundle(f(self.origins), f(self.directions), f(self.pixel_area), f(self.camera_indices),
                         f(self.nears), f(self.fars),
                         None if self.metadata is None else {k: fn(v) for k, v in self.metadata.items()}, f(self.times))

    def flatten(self) -> "RayBundle":
        return self._map(lambda t: t.reshape(-1, t.shape[-1]))

    def __getitem__(self, idx) -> "RayBundle":
        return self._map(lambda t: t[idx])

    def reshape(self, new_shape) -> "RayBundle":
        return self._map(lambda t: t.reshape(*new_shape, t.shape[-1]))

    def _apply_fn_to_fields(self, fn, dataclass_fn=None) -> "RayBundle":
        """tensor_dataclass._apply_fn_to_fields (nerfstudio/utils/tensor_dataclass.py:259-320) for the tensor fields."""
        return self._map(fn)

    def get_row_major_sliced_ray_bundle(self, start_idx: int, end_idx: int) -> "RayBundle":
        return self.flatten()[start_idx:end_idx]

    def to(self, device) -> "RayBundle":
        return self._map(lambda t: t.to(device))


class Frustums:
    """Reference-shaped view of a RaySamples (origins/directions/starts/ends/pixel_area, get_positions)."""

    def __init__(self, rs: "RaySamples"):
        self._rs = rs

    @property
    def shape(self):
        return (self._rs.num_rays, self._rs.num_samples)

    @property
    def origins(self):
        return self._rs.ray_bundle.origins[:, None, :].expand(*self.shape, 3)

    @property
    def directions(self):
        return self._rs.ray_bundle.directions[:, None, :].expand(*self.shape, 3)

    @property
    def pixel_area(self):
        return self._rs.ray_bundle.pixel_area[:, None, :].expand(*self.shape, 1)

    @property
    def starts(self):
        return self._rs._gather(self._rs.euclid_bins[:, :-1])[..., None]

    @property
    def ends(self):
        return self._rs._gather(self._rs.euclid_bins[:, 1:])[..., None]

    def get_positions(self) -> torch.Tensor:
        """World-space sample centres o + d (start+end)/2 as [R,n,3] (rays.py:48-57), computed on device."""
        u, _ = ops.positions(self._rs.ray_bundle.origins, self._rs.ray_bundle.directions, self._rs.euclid_bins,
                             self._rs.ids, ops.CONTRACT_NONE, False)
        return (u * 4.0 - 2.0).view(*self.shape, 3)


@dataclass
class RaySamples:
    ray_bundle: RayBundle
    euclid_bins: torch.Tensor  # [R, n+1]
    spacing_bins: Optional[torch.Tensor] = None  # [R, n+1]
    ids: Optional[torch.Tensor] = None  # [R, K] int32: a gathered subset of the n samples
    spacing_to_euclidean_fn: Optional[Callable] = None
    metadata: Optional[Dict[str, torch.Tensor]] = field(default=None)

    @property
    def num_rays(self) -> int:
        return self.euclid_bins.shape[0]

    @property
    def num_bins_samples(self) -> int:
        return self.euclid_bins.shape[1] - 1

    @property
    def num_samples(self) -> int:
        return self.ids.shape[1] if self.ids is not None else self.num_bins_samples

    @property
    def shape(self):
        return (self.num_rays, self.num_samples)

    def _gather(self, t: torch.Tensor) -> torch.Tensor:
        return t if self.ids is None else torch.gather(t, -1, self.ids.long())

    @property
    def frustums(self) -> Frustums:
        return Frustums(self)

    @property
    def camera_indices(self):
        ci = self.ray_bundle.camera_indices
        return None if ci is None else ci[:, None, :].expand(*self.shape, 1)

    @property
    def deltas(self):
        return self._gather(self.euclid_bins[:, 1:] - self.euclid_bins[:, :-1])[..., None]

    @property
    def spacing_starts(self):
        return None if self.spacing_bins is None else self._gather(self.spacing_bins[:, :-1])[..., None]

    @property
    def spacing_ends(self):
        return None if self.spacing_bins is None else self._gather(self.spacing_bins[:, 1:])[..., None]

    def get_weights(self, densities: torch.Tensor) -> torch.Tensor:
        """alpha-compositing weights [R,n,1] from densities [R,n,1] (rays.py:141-163)."""
        if self.ids is not None:
            raise NotImplementedError("get_weights is defined on un-gathered ray samples")
        return ops.weights_from_density(densities, self.euclid_bins)[..., None]

    def gather(self, ids: torch.Tensor) -> "RaySamples":
        """The `_apply_fn_to_fields(gather_fn)` of samnerf/sam_model.py:250-255 with best_ids [R,K]."""
        return RaySamples(self.ray_bundle, self.euclid_bins, self.spacing_bins, ids.to(torch.int32),
                          self.spacing_to_euclidean_fn, self.metadata)

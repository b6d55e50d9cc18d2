"""Fields with the reference's names and method signatures:
  Field                (nerfstudio/fields/base_field.py:30-118)
  HashMLPDensityField  (nerfstudio/fields/density_fields.py:39-128)      proposal network
  TCNNNerfactoField    (nerfstudio/fields/nerfacto_field.py:67-351)      density + colour
  SAMField             (samnerf/sam_field.py:25-140)                     SAM / ClipSeg feature heads
All sample math runs in the HIP kernels via `ops`; `density_fn` accepts a RaySamples (fast path used by the
proposal sampler) or, like the reference, a tensor of world positions.
"""
from __future__ import annotations

from enum import Enum
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from torch import nn

from . import ops
from .rays import RayBundle, RaySamples
from .spatial_distortions import SceneContraction, SpatialDistortion, contraction_mode
from .tcnn_compat import Encoding, Network, NetworkWithInputEncoding


class FieldHeadNames(Enum):
    """nerfstudio/field_components/field_heads.py:26-38 (the members this path produces)."""
    RGB = "rgb"
    DENSITY = "density"


def _positions_of(x, distortion, use_selector: bool):
    """normalised positions [N,3] (+ selector) from a RaySamples or a tensor of world positions."""
    mode = contraction_mode(distortion)
    if isinstance(x, RaySamples):
        cache = x.__dict__.setdefault("_positions_cache", {})  # the two feature heads ask for the same top-K positions
        key = (mode, use_selector)
        if key not in cache:
            cache[key] = ops.positions(x.ray_bundle.origins, x.ray_bundle.directions, x.euclid_bins, x.ids, mode, use_selector)
        u, sel = cache[key]
        return u, sel, x.shape
    flat = x.detach().reshape(-1, 3).contiguous()
    zeros = torch.zeros_like(flat)
    eb = torch.zeros((flat.shape[0], 2), device=flat.device, dtype=torch.float32)
    u, sel = ops.positions(flat, zeros, eb, None, mode, use_selector)
    return u, sel, tuple(x.shape[:-1])


class Field(nn.Module):
    def __init__(self) -> None:
        super().__init__()
        self._sample_locations = None
        self._density_before_activation = None

    def density_fn(self, positions) -> torch.Tensor:
        """base_field.py:38-55.  `positions`: RaySamples (fast path) or world positions [..., 3]."""
        density, _ = self.get_density(positions)
        return density

    def get_density(self, ray_samples):  # pragma: no cover - interface
        raise NotImplementedError

    def get_outputs(self, ray_samples, density_embedding=None) -> Dict:  # pragma: no cover - interface
        raise NotImplementedError

    def forward(self, ray_samples: RaySamples, compute_normals: bool = False) -> Dict:
        """base_field.py:99-118 (normals are off on this path: nerfacto.py:130)."""
        if compute_normals:
            raise NotImplementedError("predict_normals is disabled in the samnerf configs")
        density, density_embedding = self.get_density(ray_samples)
        field_outputs = self.get_outputs(ray_samples, density_embedding=density_embedding)
        field_outputs[FieldHeadNames.DENSITY] = density
        return field_outputs


def _hash_config(num_levels, features_per_level, log2_hashmap_size, base_res, max_res) -> Dict:
    growth = np.exp((np.log(max_res) - np.log(base_res)) / (num_levels - 1))
    return {"otype": "HashGrid", "n_levels": num_levels, "n_features_per_level": features_per_level,
            "log2_hashmap_size": log2_hashmap_size, "base_resolution": base_res, "per_level_scale": growth}


class HashMLPDensityField(Field):
    def __init__(self, aabb, num_layers: int = 2, hidden_dim: int = 64,
                 spatial_distortion: Optional[SpatialDistortion] = None, use_linear: bool = False, num_levels: int = 8,
                 max_res: int = 1024, base_res: int = 16, log2_hashmap_size: int = 18, features_per_level: int = 2,
                 device=None) -> None:
        super().__init__()
        if use_linear:
            raise NotImplementedError("use_linear=False in the samnerf configs (nerfacto.py:103-108)")
        if spatial_distortion is None:
            raise NotImplementedError("the samnerf configs always contract the scene (nerfacto.py:153-156)")
        self.register_buffer("aabb", torch.as_tensor(aabb, dtype=torch.float32), persistent=False)
        self.spatial_distortion = spatial_distortion
        self.use_linear = use_linear
        self.mlp_base = NetworkWithInputEncoding(
            n_input_dims=3, n_output_dims=1,
            encoding_config=_hash_config(num_levels, features_per_level, log2_hashmap_size, base_res, max_res),
            network_config={"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None",
                            "n_neurons": hidden_dim, "n_hidden_layers": num_layers - 1}, device=device)

    def get_density(self, ray_samples) -> Tuple[torch.Tensor, None]:
        u, sel, shape = _positions_of(ray_samples, self.spatial_distortion, True)
        raw = self.mlp_base(u)  # [N,1]
        density = ops.trunc_exp_sel(raw, sel).view(*shape, 1)
        return density, None

    def get_outputs(self, ray_samples, density_embedding=None) -> dict:
        return {}


class TCNNNerfactoField(Field):
    def __init__(self, aabb, num_images: int, num_layers: int = 2, hidden_dim: int = 64, geo_feat_dim: int = 15,
                 num_levels: int = 16, max_res: int = 2048, log2_hashmap_size: int = 19, num_layers_color: int = 3,
                 num_layers_transient: int = 2, hidden_dim_color: int = 64, hidden_dim_transient: int = 64,
                 appearance_embedding_dim: int = 32, transient_embedding_dim: int = 16,
                 use_transient_embedding: bool = False, use_semantics: bool = False, num_semantic_classes: int = 100,
                 pass_semantic_gradients: bool = False, use_pred_normals: bool = False,
                 use_average_appearance_embedding: bool = False, spatial_distortion: SpatialDistortion = None,
                 use_appearance_embedding: bool = False, device=None) -> None:
        super().__init__()
        if use_appearance_embedding or use_transient_embedding or use_semantics or use_pred_normals:
            raise NotImplementedError("appearance/transient/semantic/normal heads are disabled in the samnerf configs "
                                      "(samconfigs.py:80,134; nerfacto.py:130)")
        if spatial_distortion is None:
            raise NotImplementedError("the samnerf configs always contract the scene")
        self.register_buffer("aabb", torch.as_tensor(aabb, dtype=torch.float32), persistent=False)
        self.geo_feat_dim = geo_feat_dim
        self.spatial_distortion = spatial_distortion
        self.num_images = num_images
        self.use_appearance_embedding = False
        self.mlp_base = NetworkWithInputEncoding(
            n_input_dims=3, n_output_dims=1 + geo_feat_dim,
            encoding_config=_hash_config(num_levels, 2, log2_hashmap_size, 16, max_res),
            network_config={"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None",
                            "n_neurons": hidden_dim, "n_hidden_layers": num_layers - 1}, device=device)
        self.mlp_head = Network(
            n_input_dims=16 + geo_feat_dim, n_output_dims=3,
            network_config={"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "Sigmoid",
                            "n_neurons": hidden_dim_color, "n_hidden_layers": num_layers_color - 1}, device=device)
        self._h_full = None

    def _fusable(self) -> bool:
        enc, base, head = self.mlp_base.encoding, self.mlp_base.network, self.mlp_head
        return (enc.n_output_dims == 32 and base.n_neurons == 64 and base.n_hidden_layers == 1
                and head.n_neurons == 64 and head.n_hidden_layers == 2 and head.n_input_dims <= 32
                and base.n_output_dims <= 32)

    def forward(self, ray_samples: RaySamples, compute_normals: bool = False) -> Dict:
        """Field.forward (base_field.py:99-118) as one fused autograd node when the net has the nerfacto shape."""
        if compute_normals:
            raise NotImplementedError("predict_normals is disabled in the samnerf configs")
        if not (isinstance(ray_samples, RaySamples) and self._fusable()):
            return super().forward(ray_samples, compute_normals)
        if ray_samples.camera_indices is None:
            raise AttributeError("Camera indices are not provided.")
        u, sel, shape = _positions_of(ray_samples, self.spatial_distortion, True)
        R, S = shape
        enc = self.mlp_base.encoding
        density, rgb = ops.nerfacto_field(u, sel, ray_samples.ray_bundle.directions, R, S, enc.spec, enc.params,
                                          self.mlp_base.network.weights(), self.mlp_head.weights())
        return {FieldHeadNames.RGB: rgb.view(R, S, 3), FieldHeadNames.DENSITY: density.view(R, S, 1)}

    def get_density(self, ray_samples):
        u, sel, shape = _positions_of(ray_samples, self.spatial_distortion, True)
        h = self.mlp_base(u)  # [N, 1+geo]: column 0 = pre-activation density
        self._density_before_activation = h
        density = ops.trunc_exp_sel(h, sel).view(*shape, 1)
        self._h_full = h
        return density, h.view(*shape, -1)[..., 1:]

    def get_outputs(self, ray_samples: RaySamples, density_embedding: Optional[torch.Tensor] = None) -> Dict:
        assert density_embedding is not None
        if ray_samples.camera_indices is None:
            raise AttributeError("Camera indices are not provided.")
        R, S = ray_samples.shape
        h = self._h_full
        if h is None or density_embedding._base is not h:
            # embedding supplied from elsewhere: rebuild a [N, 1+geo] buffer around it
            h = torch.cat([torch.zeros_like(density_embedding[..., :1]), density_embedding], dim=-1).reshape(R * S, -1)
        x = ops.head_input(ray_samples.ray_bundle.directions, h, R, S)  # SH16(dir) ++ geo
        rgb = self.mlp_head(x).view(R, S, 3)
        self._h_full = None
        return {FieldHeadNames.RGB: rgb}


class SAMField(Field):
    def __init__(self, grid_layers, grid_sizes, grid_resolutions, hidden_layers=2,
                 spatial_distortion: SpatialDistortion = None, use_dino_features: bool = False,
                 use_clipseg_features: bool = False, device=None):
        super().__init__()
        assert len(grid_layers) == len(grid_sizes) and len(grid_resolutions) == len(grid_layers)
        if use_dino_features:
            raise NotImplementedError("the dino head is disabled in both samnerf configs (samconfigs.py:66,119)")
        self.spatial_distortion = spatial_distortion if spatial_distortion is not None else SceneContraction()
        self.use_dino_features = use_dino_features
        self.use_clipseg_features = use_clipseg_features
        mk = lambda: nn.ModuleList([  # noqa: E731
            SAMField._get_encoding(grid_resolutions[i][0], grid_resolutions[i][1], grid_layers[i], indim=3,
                                   hash_size=grid_sizes[i], device=device) for i in range(len(grid_layers))])
        cut = lambda n_in, n_out, nh: Network(  # noqa: E731
            n_input_dims=n_in, n_output_dims=n_out,
            network_config={"otype": "CutlassMLP", "activation": "ReLU", "output_activation": "None",
                            "n_neurons": 256, "n_hidden_layers": nh}, device=device)
        self.clip_encs = mk()
        tot = sum(e.n_output_dims for e in self.clip_encs)
        self.sam_net = cut(tot, 256, hidden_layers)
        if self.use_clipseg_features:
            self.clipseg_encs = mk()
            self.clipseg_net = cut(sum(e.n_output_dims for e in self.clipseg_encs), 192, 1)

        # the grids of one head are evaluated at the same points: under multi-GPU training their levels are sharded over
        # the ranks as one group (distributed.TableParallelLayout)
        for encs in (self.clip_encs, getattr(self, "clipseg_encs", ())):
            for i, e in enumerate(encs):
                e.tp_head = (tuple(encs), i)

    @staticmethod
    def _get_encoding(start_res, end_res, levels, indim=3, hash_size=19, device=None):
        growth = np.exp((np.log(end_res) - np.log(start_res)) / (levels - 1))
        return Encoding(n_input_dims=indim, encoding_config={
            "otype": "HashGrid", "n_levels": levels, "n_features_per_level": 8, "log2_hashmap_size": hash_size,
            "base_resolution": start_res, "per_level_scale": growth}, device=device)

    def get_outputs(self, ray_samples: RaySamples, get_feautre=["sam", "dino", "clipseg"]):  # noqa: B006 (sic)
        outputs = {}
        # positions.detach() -> L2 contraction -> (x+2)/4, no selector (sam_field.py:116-118)
        u, _, shape = _positions_of(ray_samples, self.spatial_distortion, False)
        if "sam" in get_feautre or "dino" in get_feautre:
            x = ops.hashgrid(u, [e.params for e in self.clip_encs], tuple(e.spec for e in self.clip_encs))
            outputs["hashgrid"] = x.view(*shape, -1)
        if "sam" in get_feautre:
            outputs["sam"] = self.sam_net(x).view(*shape, -1)
        if self.use_clipseg_features and "clipseg" in get_feautre:
            xc = ops.hashgrid(u, [e.params for e in self.clipseg_encs], tuple(e.spec for e in self.clipseg_encs))
            outputs["clipseg"] = self.clipseg_net(xc).view(*shape, -1)
        return outputs

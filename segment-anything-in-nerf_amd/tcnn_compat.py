"""Drop-in stand-ins for the tiny-cuda-nn torch modules the reference instantiates
(`tcnn.Encoding`, `tcnn.Network`, `tcnn.NetworkWithInputEncoding`; call sites samnerf/sam_field.py:51-109,
nerfstudio/fields/nerfacto_field.py:144-240, nerfstudio/fields/density_fields.py:73-100).

Same constructor arguments, `n_input_dims` / `n_output_dims` attributes and one flat fp32 `params` tensor per
module; the arithmetic is the reference's torch semantics (encodings.py:289-349, mlp.py:80-99) on the gfx950
kernels.  Networks are bias-free like tcnn's.  Parameters are created directly in HBM and initialised by the
counter-based fill kernel (tables U(-1,1)*1e-3 as encodings.py:257-258; layers U(+-1/sqrt(fan_in))).
"""
from __future__ import annotations

import itertools
import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from torch import nn

from . import ops
from .arena import fill_uniform_reference

_seed_counter = itertools.count(1)
_base_seed = 0


def manual_seed(seed: int) -> None:
    """Reset the deterministic parameter-initialisation stream."""
    global _seed_counter, _base_seed
    _base_seed = int(seed)
    _seed_counter = itertools.count(1)


def _next_seed() -> int:
    return (_base_seed << 20) + next(_seed_counter)


def _fill(t: torch.Tensor, lo: float, hi: float) -> None:
    seed = _next_seed()
    if t.is_cuda:
        ops.fill_uniform_(t, seed, lo, hi)
    else:  # host-side construction (shape / config tests): the identical numpy stream
        t.copy_(torch.from_numpy(fill_uniform_reference(t.numel(), seed, lo, hi)).view(t.shape))


def hash_scalings(n_levels: int, base_resolution: int, per_level_scale: float) -> torch.Tensor:
    """floor(base * g^l) evaluated as the reference's torch path does (encodings.py:252-254): fp32."""
    levels = torch.arange(n_levels)
    return torch.floor(base_resolution * np.float64(per_level_scale) ** levels)


def default_device() -> torch.device:
    return torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")


class Encoding(nn.Module):
    """tcnn.Encoding(n_input_dims=3, encoding_config={"otype": "HashGrid", ...})."""

    def __init__(self, n_input_dims: int, encoding_config: Dict, device=None, hash_init_scale: float = 1e-3):
        super().__init__()
        if encoding_config.get("otype") != "HashGrid" or n_input_dims != 3:
            raise ValueError("only the 3-D HashGrid encoding is on the hot path")
        self.n_input_dims = 3
        self.n_levels = int(encoding_config["n_levels"])
        self.n_features_per_level = int(encoding_config["n_features_per_level"])
        self.log2_hashmap_size = int(encoding_config["log2_hashmap_size"])
        self.base_resolution = int(encoding_config["base_resolution"])
        self.per_level_scale = float(encoding_config["per_level_scale"])
        if self.n_features_per_level not in (2, 8):
            raise ValueError("n_features_per_level must be 2 or 8")
        self.n_output_dims = self.n_levels * self.n_features_per_level
        device = device or default_device()
        rows = self.n_levels << self.log2_hashmap_size
        p = torch.empty((rows * self.n_features_per_level,), device=device, dtype=torch.float32)
        _fill(p, -hash_init_scale, hash_init_scale)
        self.params = nn.Parameter(p)
        self.params.hash_table_of = self  # lets the optimizer ask for the reachable rows of the coarse levels
        self.register_buffer("scalings", hash_scalings(self.n_levels, self.base_resolution,
                                                       self.per_level_scale).to(device), persistent=False)

    PRIME_Y, PRIME_Z = 2654435761, 805459861

    SPARSE_MAX_FRACTION = 0.4  # a level counts as reachable-row ("sparse") while fewer than this fraction of its rows can be addressed

    @torch.no_grad()
    def active_rows(self, max_fraction: Optional[float] = None) -> Tuple[int, torch.Tensor]:
        """Rows of the coarse levels that can EVER be addressed.

        A level of resolution s hashes lattice points (x, y, z) in [0, s]^3 (inputs in [0, 1]; one cell of slack on either
        side is added here for rounding at the borders), so at most (s + 3)^3 of its 2^T rows are reachable; the others
        never receive a gradient and Adam leaves them untouched forever.  Returns (n, rows): the first n levels are
        sparse (fewer than max_fraction * 2^T reachable rows -- the reachable count grows with the level, so they form
        a prefix) and `rows` is the sorted int64 list of their reachable row indices (l * 2^T + hash); levels >= n are
        treated as dense."""
        T, dev = self.log2_hashmap_size, self.params.device
        max_fraction = self.SPARSE_MAX_FRACTION if max_fraction is None else max_fraction
        mask32, maskT = 0xFFFFFFFF, (1 << T) - 1
        rows, n_sparse = [], 0
        for l in range(self.n_levels):
            s = int(self.scalings[l].item())
            if (s + 3) ** 3 >= max_fraction * (1 << T) * 4:  # cannot be sparse enough even before de-duplication
                break
            c = torch.arange(-1, s + 2, device=dev, dtype=torch.int64)
            x = (c & mask32)[:, None, None]
            y = ((c * self.PRIME_Y) & mask32)[None, :, None]
            z = ((c * self.PRIME_Z) & mask32)[None, None, :]
            idx = torch.unique(((x ^ y ^ z) & maskT).reshape(-1))
            if idx.numel() >= max_fraction * (1 << T):
                break
            rows.append(idx + (l << T))
            n_sparse = l + 1
        if n_sparse == 0:
            return 0, torch.empty((0,), device=dev, dtype=torch.int64)
        return n_sparse, torch.cat(rows)

    def reach_lists(self, log2B: int, max_rows: int):
        """`active_rows` grouped by the (level, bucket) workgroups of the table backward's reduce pass (B = 2^log2B buckets of
        2^(T - log2B) consecutive rows per level): (k, rows int32 [(level << T) + row, ascending], start int32 [k * B + 1], longest
        list) for the leading k <= n_sparse levels none of whose buckets lists more than `max_rows` rows (the hash does not
        spread a small lattice evenly: at T = 19 the level of resolution 60 reaches 767 rows per bucket on average, 1188 at most).
        What snf_hashgrid_bwd_presorted_adam_sp / _pair take to reduce AND step those levels over compact row indices."""
        cache = self.__dict__.setdefault("_reach_lists", {})
        key = (int(log2B), int(max_rows))
        if key not in cache:
            n_sparse, rows = self.active_rows()
            T = self.log2_hashmap_size
            out = (0, None, None, 0)
            if n_sparse and T >= log2B:
                B = 1 << log2B
                bucket = rows >> (T - log2B)  # = level * B + bucket: ascending with the rows
                start = torch.searchsorted(bucket, torch.arange((n_sparse << log2B) + 1, device=rows.device, dtype=torch.int64))
                per_level = (start[1:] - start[:-1]).view(n_sparse, B).max(dim=1).values.tolist()
                k = 0
                while k < n_sparse and per_level[k] <= max_rows:
                    k += 1
                if k:
                    n_rows = int(start[k * B])
                    out = (k, rows[:n_rows].to(torch.int32).contiguous(), start[:k * B + 1].to(torch.int32).contiguous(),
                           int(max(per_level[:k])))
            cache[key] = out
        return cache[key]

    @property
    def spec(self) -> Tuple[torch.Tensor, int, int, int]:
        return (self.scalings, self.n_levels, self.n_features_per_level, self.log2_hashmap_size)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.hashgrid(x.reshape(-1, 3), [self.params], (self.spec,))


class Network(nn.Module):
    """tcnn.Network(n_input_dims, n_output_dims, network_config) -- FullyFusedMLP / CutlassMLP, bias-free."""

    def __init__(self, n_input_dims: int, n_output_dims: int, network_config: Dict, device=None):
        super().__init__()
        self.n_input_dims, self.n_output_dims = int(n_input_dims), int(n_output_dims)
        self.n_neurons = int(network_config["n_neurons"])
        self.n_hidden_layers = int(network_config["n_hidden_layers"])
        if network_config.get("activation", "ReLU") != "ReLU":
            raise ValueError("hidden activation must be ReLU")
        self.output_activation = ops.ACT_BY_NAME[network_config.get("output_activation", "None")]
        dims = [self.n_input_dims] + [self.n_neurons] * self.n_hidden_layers + [self.n_output_dims]
        self.layer_shapes: List[Tuple[int, int]] = [(dims[i + 1], dims[i]) for i in range(len(dims) - 1)]
        total = sum(o * i for o, i in self.layer_shapes)
        device = device or default_device()
        p = torch.empty((total,), device=device, dtype=torch.float32)
        off = 0
        for o, i in self.layer_shapes:
            bound = 1.0 / math.sqrt(i)
            _fill(p[off:off + o * i], -bound, bound)
            off += o * i
        self.params = nn.Parameter(p)

    def weights(self) -> List[torch.Tensor]:
        """Per-layer [out,in] views of the flat parameter (and of its gradient arena, if attached)."""
        out, off = [], 0
        mg = getattr(self.params, "main_grad", None)
        for o, i in self.layer_shapes:
            w = self.params[off:off + o * i].view(o, i)
            if mg is not None:
                w.main_grad = mg[off:off + o * i].view(o, i)
            out.append(w)
            off += o * i
        return out

    def load_weights(self, ws) -> None:
        with torch.no_grad():
            for dst, src in zip(self.weights(), ws):
                dst.copy_(src)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x, ws = x.reshape(-1, self.n_input_dims), self.weights()
        if x.is_cuda and ops.mlp_tiny_supported(self.n_input_dims, ws, self.output_activation):
            return ops.mlp_tiny(x, ws[0], ws[1])  # proposal-network shape: one launch per direction
        return ops.mlp(x, ws, None, self.output_activation)


class NetworkWithInputEncoding(nn.Module):
    """tcnn.NetworkWithInputEncoding(n_input_dims, n_output_dims, encoding_config, network_config)."""

    def __init__(self, n_input_dims: int, n_output_dims: int, encoding_config: Dict, network_config: Dict, device=None):
        super().__init__()
        self.encoding = Encoding(n_input_dims, encoding_config, device=device)
        self.network = Network(self.encoding.n_output_dims, n_output_dims, network_config, device=device)
        self.n_input_dims, self.n_output_dims = n_input_dims, n_output_dims

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.network(self.encoding(x))

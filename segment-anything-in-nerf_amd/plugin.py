"""nerfstudio plugin entry points (nerfstudio/plugins/registry.py:35-51, group `nerfstudio.method_configs`):

    [project.entry-points."nerfstudio.method_configs"]          # pyproject.toml at the repository root
    samnerf_distill_mi355x    = "samnerf_amd.plugin:samnerf_distill"
    samnerf_no_distill_mi355x = "samnerf_amd.plugin:samnerf_no_distill"

`discover_methods()` loads each entry point and expects a `MethodSpecification(config: TrainerConfig, description: str)`
(nerfstudio/plugins/types.py:22-33); it registers `config` under `config.method_name`.

With nerfstudio and the reference's `samnerf` package importable, the specifications wrap the REFERENCE's own
TrainerConfigs (samnerf/samconfigs.py:51-164) with the model config swapped for this package's `SAMModelConfig` (same
field names and defaults, `_target` = the MI355X `SAMModel`), so `ns-train samnerf_distill_mi355x --data ...` runs the
reference's trainer, datamanager, viewer and checkpointing around the HIP hot path.  Without nerfstudio (the GPU boxes of
this build have none) the same names resolve to this package's stand-alone TrainerConfigs (configs.py), wrapped in a
local MethodSpecification of the same shape, so the entry points always load.
"""
from __future__ import annotations

import copy
import dataclasses
from dataclasses import dataclass
from typing import Any

from .configs import method_configs as _own_configs
from .model import SAMModelConfig

try:  # the real registry types, when nerfstudio is installed
    from nerfstudio.plugins.types import MethodSpecification  # type: ignore
    HAVE_NERFSTUDIO = True
except Exception:  # noqa: BLE001  (ImportError, or one of nerfstudio's own optional dependencies missing)
    HAVE_NERFSTUDIO = False

    @dataclass
    class MethodSpecification:  # nerfstudio/plugins/types.py:22-33
        config: Any
        """Trainer configuration"""
        description: str
        """Method description shown in `ns-train` help"""


def _reference_configs():
    """The reference's own method configs, or None when its packages cannot be imported."""
    if not HAVE_NERFSTUDIO:
        return None
    try:
        from samnerf.samconfigs import method_configs as ref  # type: ignore
        return ref
    except Exception:  # noqa: BLE001
        return None


def _specification(name: str) -> MethodSpecification:
    ref = _reference_configs()
    if ref is not None and name in ref:
        cfg = copy.deepcopy(ref[name])
        fields = {f.name: getattr(cfg.pipeline.model, f.name) for f in dataclasses.fields(SAMModelConfig)
                  if hasattr(cfg.pipeline.model, f.name) and f.name != "_target"}
        cfg.pipeline.model = SAMModelConfig(**fields)  # same field names / defaults by construction
    else:
        cfg = copy.deepcopy(_own_configs[name])
    cfg.method_name = f"{name}_mi355x"
    return MethodSpecification(config=cfg, description=f"{name}: SAM-NeRF render-and-distill on the MI355X HIP kernels")


samnerf_distill = _specification("samnerf_distill")
samnerf_no_distill = _specification("samnerf_no_distill")

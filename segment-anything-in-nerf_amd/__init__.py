"""segment-anything-in-nerf_amd: MI355X-native SAM-NeRF render-and-distill hot path.

Import as `samnerf_amd` (the directory name carries hyphens, so the repo-root shim `samnerf_amd.py`
registers this package under that name).  The package needs libsamnerf_hip.so (built by
`__graft_entry__.build()`); there is no CPU fallback for the compute path.
"""
__version__ = "0.1.0"

import os as _os

# One process per GPU over RCCL: the host driver of the MI355X boxes supports dmabuf IPC only; the HIP runtime reads this when it
# starts (first device call), so the default is set as early as the package can (an explicit value in the environment wins).
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from . import _lib  # noqa: F401

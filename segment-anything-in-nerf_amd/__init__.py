"""segment-anything-in-nerf_amd: MI355X-native SAM-NeRF render-and-distill hot path.

Import as `samnerf_amd` (the directory name carries hyphens, so the repo-root shim `samnerf_amd.py`
registers this package under that name).  The package needs libsamnerf_hip.so (built by
`__graft_entry__.build()`); there is no CPU fallback for the compute path.
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401

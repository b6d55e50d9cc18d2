"""segment-anything-in-nerf_amd: MI355X-native SAM-NeRF render-and-distill hot path.

Import as `samnerf_amd` (the directory name carries hyphens, so the repo-root shim `samnerf_amd.py`
registers this package under that name).  The package needs libsamnerf_hip.so (built by
`__graft_entry__.build()`); there is no CPU fallback for the compute path.
"""
__version__ = "0.1.0"

# (Importing the package does not touch the process environment.  One process per GPU over RCCL needs HSA_ENABLE_IPC_MODE_LEGACY=0 on
#  the MI355X hosts -- dmabuf IPC only: `distributed.init_distributed` sets that default before its first device call and warns when the
#  HIP runtime was already up without it; launchers (bench.py) set it before importing torch.)
from . import _lib  # noqa: F401

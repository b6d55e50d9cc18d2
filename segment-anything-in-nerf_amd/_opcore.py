"""Core of the Python face of the C-ABI (library handle, argument helpers, per-launch timing); `ops.py` is the public module.

PyTorch is plumbing here (device memory, streams, autograd bookkeeping); every numeric step runs in a
hand-written gfx950 kernel.  There is no CPU / eager fallback: tensors must live on a ROCm device.

Gradient-arena convention: a parameter tensor may carry a `main_grad` attribute (an fp32 view into the
model's flat gradient arena, see `arena.py`).  When present, the backward kernels accumulate straight into
it and autograd receives `None` for that parameter -- no per-step zero-filled temporaries, and the arena is
the RCCL all-reduce buffer.  Without it the functions behave like ordinary autograd ops.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib

CONTRACT_NONE, CONTRACT_LINF, CONTRACT_L2 = 0, 1, 2
import os as _os

ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_GELU = 0, 1, 2, 3
ACT_BY_NAME = {None: ACT_NONE, "None": ACT_NONE, "none": ACT_NONE, "ReLU": ACT_RELU, "relu": ACT_RELU,
               "Sigmoid": ACT_SIGMOID, "sigmoid": ACT_SIGMOID}


_GEMM_MODE_ENV = [_os.environ.get("SNF_GEMM_MODE")]  # "0" / "1" / "2": initial snf_set_gemm_mode (default 1), for A/B runs


def _L():
    lib = _lib.load()
    if _GEMM_MODE_ENV[0] is not None:
        mode, _GEMM_MODE_ENV[0] = int(_GEMM_MODE_ENV[0]), None
        _lib.check(lib.snf_set_gemm_mode(mode), "snf_set_gemm_mode")
    return lib


def _p(t: Optional[torch.Tensor]):
    """Raw device pointer for a c_void_p argument (ctypes converts a plain int / None itself: no wrapper object)."""
    return None if t is None else t.data_ptr()


_DEVICE_INDEX: Optional[int] = None
_HAS_GPU: Optional[bool] = None


def _has_gpu() -> bool:
    global _HAS_GPU
    if _HAS_GPU is None:
        _HAS_GPU = torch.cuda.is_available()
    return _HAS_GPU


def _stream():
    """Raw handle of the current HIP stream of this process's device (one process per GPU).  The torch.cuda.Stream object
    behind torch.cuda.current_stream() costs ~9 us per call on the host -- 0.6 ms per train step at ~60 launches."""
    global _DEVICE_INDEX
    if _DEVICE_INDEX is None:
        _DEVICE_INDEX = torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(_DEVICE_INDEX)


def _chk(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a ROCm device tensor; the MI355X path has no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        t = t.contiguous()
    return t


# --- per-kernel HIP-event timing (bench.py): events are recorded on torch's current stream, which is the stream
# every kernel is launched on (see _stream()).
_TIMING = {"names": None, "events": {}}


def enable_kernel_timing(names=None) -> None:
    """names: iterable of C-ABI entry-point names (optionally 'name/tag'), or 'all'."""
    _TIMING["names"] = None if names is None else ("all" if names == "all" else set(names))
    _TIMING["events"] = {}


def kernel_timing_summary() -> dict:
    """-> {name: {"launches": n, "total_ms": t, "avg_ms": t/n}} (synchronises)."""
    torch.cuda.synchronize()
    out = {}
    for name, evs in _TIMING["events"].items():
        tot = sum(e[0].elapsed_time(e[1]) for e in evs)
        out[name] = {"launches": len(evs), "total_ms": tot, "avg_ms": tot / max(len(evs), 1),
                     "units": sum(e[3] for e in evs)}
    return out


def kernel_timeline(base_event) -> list:
    """-> [(start_ms, end_ms, stream_id, key)] relative to `base_event` for every timed launch (synchronises)."""
    torch.cuda.synchronize()
    out = []
    for key, evs in _TIMING["events"].items():
        for a, b, sid, _ in evs:
            out.append((base_event.elapsed_time(a), base_event.elapsed_time(b), sid, key))
    return sorted(out)


_FN: dict = {}  # C-ABI entry points by name (one attribute lookup on the ctypes library per name)


def _launch(name: str, *args, tag: str = "", units: float = 0.0) -> None:
    """`units`: algorithmic bytes / flops of this launch when the caller knows them (summed by kernel_timing_summary)."""
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(_L(), name)
    sel = _TIMING["names"]
    if sel is None:  # the hot path: no timing bookkeeping, no key formatting
        rc = fn(*args)
        if rc:
            _lib.check(rc, name)
        return
    key = name + ("/" + tag if tag else "")
    if sel == "all" or key in sel or name in sel:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = fn(*args)
        b.record()
        _TIMING["events"].setdefault(key, []).append((a, b, torch.cuda.current_stream().stream_id, units))
    else:
        rc = fn(*args)
    _lib.check(rc, name)


def set_gemm_mode(mode: str) -> None:
    """'bf16x3' (default: wide layers on the bf16 matrix cores, 3-term split, fp32 accumulate), 'fp32' (exact) or
    'bf16x3+chains' (the fused 64-wide MLPs on the split as well)."""
    _lib.check(_L().snf_set_gemm_mode({"fp32": 0, "bf16x3": 1, "bf16x3+chains": 2}[mode]), "snf_set_gemm_mode")


def _linear_fwd_ws(x, w, b, N: int, I: int, O: int, act: int, y, st, tag: str) -> None:
    """snf_linear_fwd with the split-K scratch buffer the shape asks for (long-K layers with few output tiles)."""
    nbytes = int(_L().snf_linear_fwd_workspace_bytes(N, I, O))
    ws = torch.empty((max(nbytes, 16) // 4,), device=x.device, dtype=torch.float32)
    _launch("snf_linear_fwd_ws", _p(x), _p(w), _p(b), N, I, O, I, O, act, _p(y), _p(ws), nbytes, st, tag=tag)

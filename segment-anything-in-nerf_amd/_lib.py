"""Build + ctypes loader for libsamnerf_hip.so (the C-ABI of include/samnerf_hip.h).

There is NO CPU fallback: if the HIP library cannot be built or loaded, importing the ops raises.
The shared object is built in-tree (segment-anything-in-nerf_amd/lib/) so it travels with the repo
snapshot to the GPU box; a sidecar hash of the sources decides whether a rebuild is needed.
"""
from __future__ import annotations

import ctypes
import hashlib
import os
import shutil
import subprocess
from ctypes import c_char_p, c_float, c_int, c_int64, c_uint64, c_void_p

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libsamnerf_hip.so")
HASH_PATH = LIB_PATH + ".srchash"
SOURCES = ["vit.hip", "gemm_planes.hip", "batch.hip", "sampling.hip", "hashgrid.hip", "fused_head.hip", "linear.hip", "linear_b3.hip", "mlp_chain.hip", "mlp_tiny.hip", "patchconv.hip", "render.hip", "losses.hip", "optim.hip", "streams.hip"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libsamnerf_hip.so (no CPU fallback exists)")


def _source_hash() -> str:
    h = hashlib.sha256()
    files = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, "common.hpp"), os.path.join(CSRC, "grid_device.hpp"),
                                                         os.path.join(CSRC, "mlp_tiny_device.hpp"), os.path.join(INCLUDE, "samnerf_hip.h")]
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH) or not os.path.exists(HASH_PATH):
        return True
    with open(HASH_PATH) as fh:
        return fh.read().strip() != _source_hash()


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every .hip source for gfx950 and link the shared library (hipcc cross-compiles without a GPU)."""
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = _hipcc()
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    procs = []
    for s in SOURCES:
        obj = os.path.join(obj_dir, s.replace(".hip", ".o"))
        cmd = [hipcc, *HIPCC_FLAGS, "-I", INCLUDE, "-c", os.path.join(CSRC, s), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    objs = []
    for s, obj, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{out.decode(errors='replace')}")
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB_PATH]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout.decode(errors='replace')}")
    with open(HASH_PATH, "w") as fh:
        fh.write(_source_hash())
    return LIB_PATH


P, I, F = c_void_p, c_int, c_float
# name -> argtypes (restype is always int unless noted); mirrors include/samnerf_hip.h one to one
SIGNATURES = {
    "snf_sample_spacing": [P, P, P, I, I, P, P, P],
    "snf_positions": [P, P, P, P, I, I, I, I, I, P, P, P],
    "snf_positions_rows": [P, P, P, P, P, P, I, I, I, I, P, P],
    "snf_hashgrid_fwd": [P, P, P, I, I, I, I, P, I, I, P],
    "snf_hashgrid_bwd": [P, P, P, I, I, I, I, I, I, P, P],
    "snf_hashgrid_bwd_sorted": [P, P, P, I, I, I, I, I, I, P, P, c_int64, P],
    "snf_hashgrid_sort": [P, P, I, I, I, P, c_int64, P],
    "snf_hashgrid_bwd_presorted": [P, I, I, I, I, I, I, I, P, P, P, P],
    "snf_hashgrid_sort_xp": [P, P, I, I, I, P, c_int64, P],
    "snf_hashgrid_bwd_presorted_adam_xp": [P, I, I, I, I, I, I, P, P, P, I, P, P, P, F, F, F, F, I, F, P],
    "snf_hashgrid_bwd_presorted_adam": [P, I, I, I, I, I, I, I, P, P, P, I, P, P, P, F, F, F, F, I, F, P],
    "snf_hashgrid_bwd_presorted_adam_sp": [P, I, I, I, I, I, I, I, P, P, P, I, P, P, P, F, F, F, F, I, F, P, P, I, I, I, P, P],
    "snf_hashgrid_bwd_presorted_adam_pair": [P, P, I, I, I, I, P, P, P, P, I, I, P, P, P, P, P, P, P, P, I, P, P, I, I, I, P, F, F, F, F,
                                             I, F, P],
    "snf_split_weights_b3": [P, I, I, P, P, P],
    "snf_grid_head_fused_fwd": [P, P, P, I, P, P, I, I, P, P, I, P, I, P, I, P],
    "snf_hashgrid_bucket_bits": [I, I],
    "snf_hashgrid_sparse_max_rows": [I],
    "snf_hashgrid_bwd_presorted_adam_fx": [P, I, I, I, I, I, I, I, P, P, P, I, P, P, P, F, F, F, F, I, F, P, P],
    "snf_hashgrid_bwd_sorted_ex": [P, P, P, I, I, I, I, I, I, I, P, P, c_int64, P],
    "snf_linear_fwd": [P, P, P, I, I, I, I, I, I, P, P],
    "snf_linear_fwd_ws": [P, P, P, I, I, I, I, I, I, P, P, c_int64, P],
    "snf_linear_bwd_data": [P, P, P, I, I, I, I, I, I, I, P, P],
    "snf_linear_bwd_weight": [P, P, P, I, I, I, I, I, I, I, P, P, P],
    "snf_linear_bwd_weight_ws": [P, P, P, I, I, I, I, I, I, I, P, P, P, c_int64, P],
    "snf_linear_fwd_mean": [P, P, I, I, I, I, P, I, P, P, P, I, P],
    "snf_linear_bwd_data_rows": [P, P, I, P, I, P, I, I, I, I, I, I, I, P, P],
    "snf_linear_bwd_weight_rows": [P, P, I, P, I, P, I, I, I, I, I, I, I, P, P, c_int64, P],
    "snf_mlp64_fwd": [P, I, P, I, P, P, I, I, I, c_int64, P, P, P, I, P],
    "snf_mlp64_fwd_density": [P, I, P, I, P, P, I, I, I, c_int64, P, P, P, I, P, P, P],
    "snf_mlp64_bwd_data": [P, I, I, P, P, I, P, I, P, P, I, I, I, c_int64, P, P, P, P, P, I, P, I, P],
    "snf_mlp64_bwd_fused": [P, I, I, P, P, I, P, I, P, I, P, P, I, I, I, c_int64, P, P, P, I, P, P, P, P, c_int64, P],
    "snf_mlp64_fwd_sh": [P, I, I, P, I, I, P, P, P, I, I, I, P, P, P, I, P],
    "snf_mlp64_bwd_fused_sh": [P, I, P, I, P, I, I, P, I, I, P, P, P, I, I, I, P, I, P, P, P, P, c_int64, P],
    "snf_head_input": [P, P, I, I, I, I, P, I, P],
    "snf_weights_fwd": [P, I, I, P, P, I, I, P, P, P],
    "snf_weights_bwd": [P, I, I, P, P, P, I, I, P, P],
    "snf_trunc_exp_fwd": [P, I, P, c_int64, P, P],
    "snf_prop_density_fwd": [P, P, P, I, I, I, I, P, P, I, P, P, P],
    "snf_trunc_exp_bwd": [P, I, P, P, c_int64, P, I, P],
    "snf_pdf_resample": [P, P, P, P, P, I, I, I, F, F, P, P, P],
    "snf_composite_fwd": [P, P, P, I, I, I, P, P, P, P],
    "snf_composite_bwd": [P, P, P, I, I, P, P, P],
    "snf_topk_sharpen": [P, I, I, I, F, P, P, P],
    "snf_topk_sharpen_rows": [P, P, P, I, I, I, F, P, P, P],
    "snf_feature_mean_fwd": [P, P, I, I, I, P, P],
    "snf_feature_mean_bwd": [P, P, I, I, I, P, P],
    "snf_interlevel": [P, P, P, P, I, I, I, F, P, P, P],
    "snf_distortion": [P, P, I, I, F, P, P, P],
    "snf_add_scaled": [c_int64, F, P, P, P],
    "snf_nerf_loss_summary": [P, P, F, P, F, F, I, P, P],
    "snf_adam_step": [P, P, P, P, c_int64, F, F, F, F, I, F, I, P],
    "snf_adam_step_rows": [P, P, P, P, P, c_int64, I, F, F, F, F, I, F, I, P],
    "snf_set_adam_launch": [I, I, I],
    "snf_step_guard": [P],
    "snf_guard_update": [P, I, P, P],
    "snf_guard_scan": [P, c_int64, P, P],
    "snf_patchify": [P, I, I, I, I, P, P],
    "snf_sam_preprocess": [P, I, I, I, I, I, I, P, P, P, P],
    "snf_layernorm": [P, P, I, I, P, P, F, P, P, P],
    "snf_window_partition": [P, I, I, I, I, I, P, P],
    "snf_window_merge_add": [P, P, I, I, I, I, I, P, P],
    "snf_relpos": [P, I, I, I, I, I, P, P, P, P],
    "snf_attention": [P, P, I, I, I, I, I, F, P, P],
    "snf_split_planes": [P, c_int64, P, P, P],
    "snf_split_planes_kb": [P, I, I, P, P, P],
    "snf_linear_planes_fwd": [P, P, P, P, P, I, I, I, I, P, P, P, P],
    "snf_linear_planes_fwd_shape": [P, P, P, P, P, I, I, I, I, P, P, P, I, I, P],
    "snf_linear_planes_kb_fwd": [P, P, P, P, P, I, I, I, I, P, P, P, P],
    "snf_layernorm_planes": [P, P, I, I, P, P, F, P, P, P, I, I, I, I, P],
    "snf_layernorm_planes_merge": [P, P, I, I, P, P, F, P, P, P, I, I, I, P],
    "snf_attention_planes": [P, P, I, I, I, I, I, F, P, P, P],
    "snf_attention_planes_rp": [P, P, P, I, I, I, I, I, F, P, P, P],
    "snf_pixel_indices": [P, I, I, I, I, I, P, P],
    "snf_generate_rays": [P, I, P, P, I, P, P, P, P, P],
    "snf_gather_nearest": [P, I, I, I, P, I, I, I, I, I, I, P, P],
    "snf_rowmse_loss_fwd": [P, P, I, I, F, I, P, P, P],
    "snf_rowmse_loss_bwd": [P, P, I, I, F, I, P, P, P, P],
    "snf_mlp_tiny_supported": [I, I, I],
    "snf_mlp_tiny_fwd": [P, I, P, P, I, I, c_int64, P, P, P],
    "snf_mlp_tiny_bwd": [P, P, I, P, P, P, I, I, c_int64, P, I, P, P, P],
    "snf_patch_unfold": [P, I, I, I, I, P, P],
    "snf_patch_fold": [P, I, I, I, I, P, P],
    "snf_patch_unfold_mean": [P, I, I, I, I, P, P],
    "snf_patch_fold_mean": [P, I, I, I, I, P, P],
    "snf_fill_uniform": [P, c_int64, c_uint64, F, F, P],
    "snf_stream_create_cu_mask": [I, ctypes.POINTER(c_void_p)],
    "snf_stream_create_priority": [I, ctypes.POINTER(c_void_p)],
    "snf_stream_destroy": [P],
}

_LIB = None


def load(auto_build: bool = True) -> ctypes.CDLL:
    """dlopen the library (building it first when the sources changed). Raises if unavailable."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # torch must come first: it brings its own HIP runtime (libamdhip64); dlopen-ing ours before torch would bind the
    # process to a second runtime copy and torch then sees "no ROCm-capable device".
    import torch  # noqa: F401
    if auto_build and needs_build():
        build()
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = ctypes.CDLL(LIB_PATH)
    lib.snf_version.restype = c_int
    lib.snf_version.argtypes = []
    lib.snf_last_error.restype = c_char_p
    lib.snf_last_error.argtypes = []
    lib.snf_set_gemm_mode.restype = c_int
    lib.snf_set_gemm_mode.argtypes = [c_int]
    lib.snf_get_gemm_mode.restype = c_int
    lib.snf_get_gemm_mode.argtypes = []
    lib.snf_hashgrid_bwd_workspace_bytes.restype = c_int64
    lib.snf_hashgrid_bwd_workspace_bytes.argtypes = [c_int, c_int, c_int]
    lib.snf_linear_fwd_workspace_bytes.restype = c_int64
    lib.snf_linear_fwd_workspace_bytes.argtypes = [c_int, c_int, c_int]
    lib.snf_mlp64_bwd_fused_workspace_bytes.restype = c_int64
    lib.snf_mlp64_bwd_fused_workspace_bytes.argtypes = [c_int]
    lib.snf_linear_bwd_weight_workspace_bytes.restype = c_int64
    lib.snf_linear_bwd_weight_workspace_bytes.argtypes = [c_int, c_int, c_int]
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = c_int
        fn.argtypes = argtypes
    _LIB = lib
    return lib


class SnfError(RuntimeError):
    pass


def check(rc: int, name: str) -> None:
    if rc != 0:
        msg = load().snf_last_error().decode(errors="replace")
        raise SnfError(f"{name} failed (rc={rc}): {msg}")

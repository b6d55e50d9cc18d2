// common.hpp -- shared helpers for the gfx950 kernels of libsamnerf_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <float.h>
#include "samnerf_hip.h"

namespace snf {

void set_error(const char* fmt, ...);

#define SNF_REQUIRE(cond, ...)                  \
    do {                                        \
        if (!(cond)) {                          \
            snf::set_error(__VA_ARGS__);        \
            return SNF_ERR_ARG;                 \
        }                                       \
    } while (0)

#define SNF_LAUNCH_CHECK(name)                                                      \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) {                                                    \
            snf::set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return SNF_ERR_LAUNCH;                                                  \
        }                                                                           \
    } while (0)

constexpr int WAVE = 64;

// ---- non-finite-gradient guard (snf_step_guard, optim.hip) ---------------------------------------------------------------
// A guard is a device record int32[2] = {veto, skipped}: `veto` != 0 means the loss of the step in flight was not finite (written by
// snf_guard_update), `skipped` counts the steps vetoed before it.  Every optimizer-side kernel (snf_adam_step, snf_adam_step_rows, the
// fused backward + Adam epilogues of hashgrid.hip) launched while a guard is bound leaves p / exp_avg / exp_avg_sq untouched on a vetoed
// step and takes its bias corrections from `step - skipped` -- what torch.cuda.amp.GradScaler.step does when it finds an inf / NaN
// gradient (nerfstudio/engine/trainer.py:419-437, engine/optimizers.py:138-149): the optimizer's step is not called, its state['step']
// does not advance.  The host never reads the record inside a step.
const int32_t* current_guard();  // the record bound by snf_step_guard on this host thread, or nullptr

struct GuardAdam {
    bool veto;
    float step_size, inv_sqrt_bc2;
};
// `guard` is a kernel argument (wave-uniform address: scalar loads).  With skipped == 0 -- always, unless a loss has been non-finite
// before -- the host's bias corrections are kept bit for bit.
__device__ __forceinline__ GuardAdam guard_adam(const int32_t* __restrict__ guard, float lr, float b1, float b2, int step,
                                               float step_size, float inv_sqrt_bc2) {
    GuardAdam r{false, step_size, inv_sqrt_bc2};
    if (guard != nullptr) {
        const int veto = guard[0], skipped = guard[1];
        r.veto = veto != 0;
        if (skipped > 0) {
            const float t = (float)(step - skipped > 1 ? step - skipped : 1);
            r.step_size = lr / (1.f - powf(b1, t));
            r.inv_sqrt_bc2 = 1.f / sqrtf(1.f - powf(b2, t));
        }
    }
    return r;
}

// Degree-4 real spherical harmonics of a direction (nerfstudio/utils/math.py:27-73 `components_from_spherical_harmonics`, levels = 4).
// One definition for every kernel that needs it (snf_head_input, the colour net's fused loaders), evaluated WITHOUT FMA contraction:
// the torch reference rounds every product and sum separately, and all kernels then agree bit for bit.
__device__ __forceinline__ void sh16_of(float x, float y, float z, float (&o)[16]) {
#pragma clang fp contract(off)
    const float xx = x * x, yy = y * y, zz = z * z;
    o[0] = 0.28209479177387814f;
    o[1] = 0.4886025119029199f * y;
    o[2] = 0.4886025119029199f * z;
    o[3] = 0.4886025119029199f * x;
    o[4] = 1.0925484305920792f * x * y;
    o[5] = 1.0925484305920792f * y * z;
    o[6] = 0.9461746957575601f * zz - 0.31539156525251999f;
    o[7] = 1.0925484305920792f * x * z;
    o[8] = 0.5462742152960396f * (xx - yy);
    o[9] = 0.5900435899266435f * y * (3.f * xx - yy);
    o[10] = 2.890611442640554f * x * y * z;
    o[11] = 0.4570457994644658f * y * (5.f * zz - 1.f);
    o[12] = 0.3731763325901154f * z * (5.f * zz - 3.f);
    o[13] = 0.4570457994644658f * x * (5.f * zz - 1.f);
    o[14] = 1.445305721320277f * z * (xx - yy);
    o[15] = 0.5900435899266435f * x * (xx - 3.f * yy);
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }

// inclusive prefix sum across the 64 lanes of a wavefront
__device__ __forceinline__ float wave_incl_scan(float v) {
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        float t = __shfl_up(v, d, WAVE);
        if (lane >= d) v += t;
    }
    return v;
}

// inclusive suffix sum (lane i gets sum over lanes >= i)
__device__ __forceinline__ float wave_incl_scan_rev(float v) {
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        float t = __shfl_down(v, d, WAVE);
        if (lane + d < WAVE) v += t;
    }
    return v;
}

// exclusive scans derived from an inclusive one by a lane shift (never by subtraction: inf - inf, cancellation)
__device__ __forceinline__ float shift_up1(float incl) {
    const float t = __shfl_up(incl, 1, WAVE);
    return lane_id() == 0 ? 0.f : t;
}
__device__ __forceinline__ float shift_down1(float incl) {
    const float t = __shfl_down(incl, 1, WAVE);
    return lane_id() == WAVE - 1 ? 0.f : t;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, WAVE);
    return v;
}

__device__ __forceinline__ float nan_to_num(float x) {
    if (x != x) return 0.f;
    if (x == INFINITY) return FLT_MAX;
    if (x == -INFINITY) return -FLT_MAX;
    return x;
}

// spacing_fn / spacing_fn_inv of UniformLinDispPiecewiseSampler (ray_samplers.py:242-243)
__device__ __forceinline__ float spacing_fn(float x) { return x < 1.f ? x / 2.f : 1.f - 1.f / (2.f * x); }
__device__ __forceinline__ float spacing_fn_inv(float y) { return y < 0.5f ? 2.f * y : 1.f / (2.f - 2.f * y); }

// torch.linspace(start, end, steps)[i] as the CPU kernel evaluates it (symmetric around the midpoint)
__device__ __forceinline__ float linspace_at(float start, float end, int steps, int i) {
    const float step = (end - start) / (float)(steps - 1);
    return (i < steps / 2) ? start + step * (float)i : end - step * (float)(steps - 1 - i);
}

inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// linear_b3.hip: bf16x3 path of the wide layers; return 1 when they launched the layer
int b3_try_fwd(const float* X, const float* W, const float* bias, int N, int I, int O, int ldx, int ldy, int act, float* Y,
               snf_stream_t stream);
int b3_try_fwd_splitk(const float* X, const float* W, int N, int I, int O, int ldx, int ksplit, int splits, float* P,
                      snf_stream_t stream);
int b3_try_bwd_weight(const float* dY, const float* Y, const float* X, int N, int I, int O, int lddy, int ldy, int ldx,
                      int act, float* dW, float* dbias, snf_stream_t stream);
int b3_try_bwd_data(const float* dY, const float* Y, const float* W, int N, int I, int O, int lddy, int ldy, int lddx,
                    int act, float* dX, snf_stream_t stream, const float* rscale = nullptr, int rgroup = 1, int aux_bits = 0);
int b3_try_fwd_mean(const float* X, const float* W, int N, int I, int O, int ldx, int ldy, float* Y, const float* wk, int group,
                    float* hbar, uint8_t* ybits, snf_stream_t stream);
long long b3_wgrad_full_workspace_bytes(int N, int I, int O);
int b3_try_bwd_weight_full(const float* dY, const float* Y, const float* X, int N, int I, int O, int lddy, int ldy, int ldx,
                           int act, float* dW, float* dbias, void* workspace, long long workspace_bytes, snf_stream_t stream,
                           const float* rscale = nullptr, int rgroup = 1, int aux_bits = 0);
bool b3_enabled();  // snf_set_gemm_mode(1): bf16 3-term split on the matrix cores (default); 0: exact fp32

}  // namespace snf

// optim.hip -- parameter-arena kernels (gfx950): fused Adam and the counter-based initialiser.
//   snf_adam_step    : torch.optim.Adam semantics (nerfstudio/engine/optimizers.py:100-147; eps 1e-15,
//                      samnerf/samconfigs.py:144-161) over one contiguous slice of the flat fp32 arena;
//                      streams p,g,m,v once (16 B/lane dwordx4), optionally folds the data-parallel 1/world
//                      gradient scale and the zero_grad of the next step into the same pass.
//   snf_fill_uniform : splitmix64 counter hash -> U[lo,hi); the package has the identical numpy expression.
#include "common.hpp"
#include <math.h>
#include <stdarg.h>
#include <string.h>

namespace snf {

static thread_local char g_err[512] = "";

static thread_local const int32_t* g_guard = nullptr;
const int32_t* current_guard() { return g_guard; }

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

__device__ __forceinline__ void adam1(float& p, float& g, float& m, float& v, float gs, float b1, float b2, float step_size,
                                      float inv_sqrt_bc2, float eps) {
    const float gg = g * gs;
    m = m + (gg - m) * (1.f - b1);
    v = v * b2 + (1.f - b2) * gg * gg;
    const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
    p = p - step_size * (m / denom);
}

// Streaming pass: every thread keeps two independent 16-byte groups of p,g,m,v in flight (8 loads) and writes with
// non-temporal stores (the arenas are far larger than the caches and are not re-read before the next step).
template <typename T>
__device__ __forceinline__ T ldnt(const T* p) { return __builtin_nontemporal_load(p); }
template <typename T>
__device__ __forceinline__ void stnt(T* p, T v) { __builtin_nontemporal_store(v, p); }

typedef float f4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, long long n, float b1, float b2, float step_size,
                                              float inv_sqrt_bc2, float eps, float gs, int zero_grad,
                                              const int32_t* __restrict__ guard, float lr, int step) {
    const long long n4 = n >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    f4* p4 = reinterpret_cast<f4*>(p);
    f4* g4 = reinterpret_cast<f4*>(g);
    f4* m4 = reinterpret_cast<f4*>(m);
    f4* v4 = reinterpret_cast<f4*>(v);
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const GuardAdam ga = guard_adam(guard, lr, b1, b2, step, step_size, inv_sqrt_bc2);
    step_size = ga.step_size;
    inv_sqrt_bc2 = ga.inv_sqrt_bc2;
    if (ga.veto) {  // a vetoed step: parameters and moments stay, the gradient is cleared as zero_grad would
        if (zero_grad) {
            for (; i < n4; i += stride) stnt(g4 + i, zero);
            const long long t = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x;
            if (t < n) g[t] = 0.f;
        }
        return;
    }
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        f4 P[U], G[U], M[U], V[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long j = i + u * stride;
            P[u] = ldnt(p4 + j); G[u] = ldnt(g4 + j); M[u] = ldnt(m4 + j); V[u] = ldnt(v4 + j);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float a = P[u][c], bg = G[u][c], cm = M[u][c], dv = V[u][c];
                adam1(a, bg, cm, dv, gs, b1, b2, step_size, inv_sqrt_bc2, eps);
                P[u][c] = a; M[u][c] = cm; V[u][c] = dv;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long j = i + u * stride;
            stnt(p4 + j, P[u]); stnt(m4 + j, M[u]); stnt(v4 + j, V[u]);
            if (zero_grad) stnt(g4 + j, zero);
        }
    }
    for (; i < n4; i += stride) {
        f4 P0 = ldnt(p4 + i), G0 = ldnt(g4 + i), M0 = ldnt(m4 + i), V0 = ldnt(v4 + i);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float a = P0[c], bg = G0[c], cm = M0[c], dv = V0[c];
            adam1(a, bg, cm, dv, gs, b1, b2, step_size, inv_sqrt_bc2, eps);
            P0[c] = a; M0[c] = cm; V0[c] = dv;
        }
        stnt(p4 + i, P0); stnt(m4 + i, M0); stnt(v4 + i, V0);
        if (zero_grad) stnt(g4 + i, zero);
    }
    // tail (n not a multiple of 4)
    const long long t = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        adam1(p[t], g[t], m[t], v[t], gs, b1, b2, step_size, inv_sqrt_bc2, eps);
        if (zero_grad) g[t] = 0.f;
    }
}

// Adam over a list of table rows (F = 2 or 8 consecutive floats each, `rows[i]` = element offset of row i from the arena
// base).  Used for the coarse hash-grid levels, where only the rows the (res+1)^3 lattice hashes to can ever receive a
// gradient: all other rows keep g = m = v = 0 forever, so skipping them is exact (p - lr * 0 / (0 + eps) = p).
template <int F>
__global__ __launch_bounds__(256) void k_adam_rows(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, const int* __restrict__ rows, long long nrows,
                                                   float b1, float b2, float step_size, float inv_sqrt_bc2, float eps,
                                                   float gs, int zero_grad, const int32_t* __restrict__ guard, float lr,
                                                   int step) {
    constexpr int VPR = F == 8 ? 2 : 1;  // vector accesses per row (float4 x 2 or float2 x 1)
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= nrows * VPR) return;
    const long long r = t / VPR;
    const size_t off = (size_t)rows[r] + (size_t)(t - r * VPR) * 4;
    const GuardAdam ga = guard_adam(guard, lr, b1, b2, step, step_size, inv_sqrt_bc2);
    step_size = ga.step_size;
    inv_sqrt_bc2 = ga.inv_sqrt_bc2;
    if (ga.veto) {
        if (zero_grad) {
            if constexpr (F == 8) *reinterpret_cast<f4*>(g + off) = f4{0.f, 0.f, 0.f, 0.f};
            else g[off] = g[off + 1] = 0.f;
        }
        return;
    }
    if constexpr (F == 8) {
        f4 P = *reinterpret_cast<f4*>(p + off), G = *reinterpret_cast<f4*>(g + off);
        f4 M = *reinterpret_cast<f4*>(m + off), V = *reinterpret_cast<f4*>(v + off);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float a = P[c], bg = G[c], cm = M[c], dv = V[c];
            adam1(a, bg, cm, dv, gs, b1, b2, step_size, inv_sqrt_bc2, eps);
            P[c] = a; M[c] = cm; V[c] = dv;
        }
        *reinterpret_cast<f4*>(p + off) = P; *reinterpret_cast<f4*>(m + off) = M; *reinterpret_cast<f4*>(v + off) = V;
        if (zero_grad) *reinterpret_cast<f4*>(g + off) = f4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            adam1(p[off + c], g[off + c], m[off + c], v[off + c], gs, b1, b2, step_size, inv_sqrt_bc2, eps);
            if (zero_grad) g[off + c] = 0.f;
        }
    }
}

// launch shape of k_adam: {max blocks, threads per block, unroll}; snf_set_adam_launch overrides (tuning hook)
static int g_adam_launch[3] = {2048, 256, 2};

__global__ __launch_bounds__(256) void k_fill_uniform(float* __restrict__ x, long long n, unsigned long long seed, float lo,
                                                      float hi) {
#pragma clang fp contract(off)
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        unsigned long long z = seed + (unsigned long long)(i + 1) * 0x9E3779B97F4A7C15ULL;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        z = z ^ (z >> 31);
        const float u = (float)(z >> 40) * (1.0f / 16777216.0f);
        x[i] = lo + (hi - lo) * u;  // separately rounded, as numpy does
    }
}

}  // namespace snf

using namespace snf;

extern "C" int snf_version(void) { return 100; }
extern "C" const char* snf_last_error(void) { return g_err; }

extern "C" int snf_adam_step(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                             float eps, int step, float grad_scale, int zero_grad, snf_stream_t stream) {
    SNF_REQUIRE(p && g && m && v, "snf_adam_step: null pointer");
    SNF_REQUIRE(n > 0 && step >= 1, "snf_adam_step: bad n=%lld step=%d", (long long)n, step);
    SNF_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0,
                "snf_adam_step: arena slices must be 16-byte aligned");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    const int threads = g_adam_launch[1], U = g_adam_launch[2];
    long long blocks = (n / 4 + threads - 1) / threads;
    if (blocks > g_adam_launch[0]) blocks = g_adam_launch[0];
    if (blocks < 1) blocks = 1;
#define SNF_ADAM_LAUNCH(UU)                                                                                             \
    hipLaunchKernelGGL(k_adam<UU>, dim3((unsigned)blocks), dim3(threads), 0, (hipStream_t)stream, p, g, m, v, (long long)n, \
                       beta1, beta2, step_size, inv_sqrt_bc2, eps, grad_scale, zero_grad, current_guard(), lr, step)
    if (U == 1) SNF_ADAM_LAUNCH(1); else if (U == 4) SNF_ADAM_LAUNCH(4); else SNF_ADAM_LAUNCH(2);
#undef SNF_ADAM_LAUNCH
    SNF_LAUNCH_CHECK("snf_adam_step");
    return SNF_OK;
}

extern "C" int snf_adam_step_rows(float* p, float* g, float* m, float* v, const int32_t* rows, int64_t nrows, int F, float lr,
                                  float beta1, float beta2, float eps, int step, float grad_scale, int zero_grad,
                                  snf_stream_t stream) {
    SNF_REQUIRE(p && g && m && v && rows, "snf_adam_step_rows: null pointer");
    SNF_REQUIRE(nrows > 0 && step >= 1 && (F == 2 || F == 8), "snf_adam_step_rows: bad nrows=%lld step=%d F=%d", (long long)nrows,
                step, F);
    SNF_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0,
                "snf_adam_step_rows: arena bases must be 16-byte aligned");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    const long long threads = (long long)nrows * (F == 8 ? 2 : 1);
    const unsigned blocks = (unsigned)((threads + 255) / 256);
    if (F == 8)
        hipLaunchKernelGGL(k_adam_rows<8>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (const int*)rows,
                           (long long)nrows, beta1, beta2, step_size, inv_sqrt_bc2, eps, grad_scale, zero_grad, current_guard(), lr,
                           step);
    else
        hipLaunchKernelGGL(k_adam_rows<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (const int*)rows,
                           (long long)nrows, beta1, beta2, step_size, inv_sqrt_bc2, eps, grad_scale, zero_grad, current_guard(), lr,
                           step);
    SNF_LAUNCH_CHECK("snf_adam_step_rows");
    return SNF_OK;
}

// ---- the non-finite-gradient guard (common.hpp has the device side) ----------------------------------------------------------
// snf_guard_update: one thread.  First commits the previous step's verdict (a vetoed step counts as skipped), then judges this step's:
// veto = any of the n loss values is inf / NaN.
__global__ void k_guard_update(const float* __restrict__ values, int n, int32_t* __restrict__ guard) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (guard[0] != 0) guard[1] += 1;
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        const float x = values[i];
        bad |= !(fabsf(x) <= FLT_MAX);  // false for NaN and for +-inf
    }
    guard[0] = bad;
}

// veto |= any of x[0 .. n) is inf / NaN (grid-stride; one atomic per workgroup that saw one)
__global__ __launch_bounds__(256) void k_guard_scan(const float* __restrict__ x, long long n, int32_t* __restrict__ guard) {
    int bad = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) bad |= !(fabsf(x[i]) <= FLT_MAX);
    if (__syncthreads_or(bad) && threadIdx.x == 0) atomicOr(guard, 1);
}

extern "C" int snf_guard_scan(const float* x, int64_t n, int32_t* guard, snf_stream_t stream) {
    SNF_REQUIRE(x && guard && n >= 1, "snf_guard_scan: bad argument");
    long long blocks = (n + 256 * 8 - 1) / (256 * 8);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_guard_scan, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (long long)n, guard);
    SNF_LAUNCH_CHECK("snf_guard_scan");
    return SNF_OK;
}

extern "C" int snf_step_guard(const int32_t* guard) {
    g_guard = guard;
    return SNF_OK;
}

extern "C" int snf_guard_update(const float* values, int n, int32_t* guard, snf_stream_t stream) {
    SNF_REQUIRE(values && guard && n >= 1, "snf_guard_update: bad argument");
    hipLaunchKernelGGL(k_guard_update, dim3(1), dim3(64), 0, (hipStream_t)stream, values, n, guard);
    SNF_LAUNCH_CHECK("snf_guard_update");
    return SNF_OK;
}

extern "C" int snf_set_adam_launch(int max_blocks, int threads, int unroll) {
    SNF_REQUIRE(max_blocks >= 1 && (threads == 64 || threads == 128 || threads == 256) && (unroll == 1 || unroll == 2 || unroll == 4),
                "snf_set_adam_launch: max_blocks >= 1, threads in {64,128,256}, unroll in {1,2,4}");
    g_adam_launch[0] = max_blocks; g_adam_launch[1] = threads; g_adam_launch[2] = unroll;
    return SNF_OK;
}

extern "C" int snf_fill_uniform(float* x, int64_t n, uint64_t seed, float lo, float hi, snf_stream_t stream) {
    SNF_REQUIRE(x && n > 0, "snf_fill_uniform: bad argument");
    long long blocks = (n + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(k_fill_uniform, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (long long)n,
                       (unsigned long long)seed, lo, hi);
    SNF_LAUNCH_CHECK("snf_fill_uniform");
    return SNF_OK;
}

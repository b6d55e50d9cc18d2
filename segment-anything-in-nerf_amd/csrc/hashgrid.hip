// hashgrid.hip -- multiresolution hash-grid encoding, forward and table-gradient backward (gfx950).
//
// Semantics = the reference's torch path (HashEncoding.pytorch_fwd, field_components/encodings.py:289-349):
//   scaled = u * s_l ; corners ceil/floor ; index = (x ^ y*2654435761 ^ z*805459861) mod 2^T + l*2^T ;
//   trilinear blend in the reference's order.  The int64 hash of the reference equals the uint32-wrapping
//   hash below because 2^T divides 2^32 and all coordinates are non-negative.
//
// Layout in HBM: table rows are level-major ([L][2^T][F] fp32), so one level of an F=8 grid is a 16 MiB
// slab and a corner is one 32-byte row (two dwordx4 loads); F=2 corners are 8-byte rows.
// Launch: grid = (ceil(N/256), L): blockIdx.y is the level, so a workgroup (and its neighbours in x)
// gather from ONE level slab; a thread owns one (sample, level) pair and keeps all 8 corner loads in flight.
#include "common.hpp"

namespace snf {

constexpr uint32_t PRIME_Y = 2654435761u;
constexpr uint32_t PRIME_Z = 805459861u;

template <int F> struct Vec;
template <> struct Vec<2> { float v[2]; };
template <> struct Vec<8> { float v[8]; };

template <int F>
__device__ __forceinline__ void load_row(const float* __restrict__ p, float (&o)[F]) {
    if constexpr (F == 2) {
        const float2 t = *reinterpret_cast<const float2*>(p);
        o[0] = t.x; o[1] = t.y;
    } else {
        const float4 a = *reinterpret_cast<const float4*>(p);
        const float4 b = *reinterpret_cast<const float4*>(p + 4);
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
        o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    }
}

struct Corners {
    uint32_t idx[8];
    float ox, oy, oz;
};

__device__ __forceinline__ Corners corners_of(const float* __restrict__ u, int n, float s, uint32_t mask) {
    // separately rounded product (no FMA into the subtraction below): the reference rounds `scaled` before
    // taking floor / the fractional offset, and at resolution 2047 one ulp of `scaled` is 1e-4 of a cell.
    // (HIP's __fmul_rn is a plain `*`, so contraction has to be switched off with the pragma.)
#pragma clang fp contract(off)
    const float px = u[(size_t)n * 3 + 0] * s;
    const float py = u[(size_t)n * 3 + 1] * s;
    const float pz = u[(size_t)n * 3 + 2] * s;
    const float fxf = floorf(px), fyf = floorf(py), fzf = floorf(pz);
    const uint32_t cx = (uint32_t)(int)ceilf(px), cy = (uint32_t)(int)ceilf(py) * PRIME_Y,
                   cz = (uint32_t)(int)ceilf(pz) * PRIME_Z;
    const uint32_t fx = (uint32_t)(int)fxf, fy = (uint32_t)(int)fyf * PRIME_Y, fz = (uint32_t)(int)fzf * PRIME_Z;
    Corners c;
    c.ox = px - fxf;
    c.oy = py - fyf;
    c.oz = pz - fzf;
    // corner naming of encodings.py:318-325
    c.idx[0] = (cx ^ cy ^ cz) & mask;
    c.idx[1] = (cx ^ fy ^ cz) & mask;
    c.idx[2] = (fx ^ fy ^ cz) & mask;
    c.idx[3] = (fx ^ cy ^ cz) & mask;
    c.idx[4] = (cx ^ cy ^ fz) & mask;
    c.idx[5] = (cx ^ fy ^ fz) & mask;
    c.idx[6] = (fx ^ fy ^ fz) & mask;
    c.idx[7] = (fx ^ cy ^ fz) & mask;
    return c;
}

template <int F>
__global__ __launch_bounds__(256) void k_hashgrid_fwd(const float* __restrict__ u, const float* __restrict__ table,
                                                      const float* __restrict__ scalings, int N, int log2_T,
                                                      float* __restrict__ out, int ld_out, int col_off) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int l = blockIdx.y;
    if (n >= N) return;
    const uint32_t mask = (1u << log2_T) - 1u;
    const Corners c = corners_of(u, n, scalings[l], mask);
    const float* __restrict__ slab = table + ((size_t)l << log2_T) * F;
    float f[8][F];
#pragma unroll
    for (int k = 0; k < 8; ++k) load_row<F>(slab + (size_t)c.idx[k] * F, f[k]);
    const float ox = c.ox, oy = c.oy, oz = c.oz;
    const float mx = 1.f - ox, my = 1.f - oy, mz = 1.f - oz;
    float r[F];
#pragma unroll
    for (int j = 0; j < F; ++j) {
        const float f03 = f[0][j] * ox + f[3][j] * mx;
        const float f12 = f[1][j] * ox + f[2][j] * mx;
        const float f56 = f[5][j] * ox + f[6][j] * mx;
        const float f47 = f[4][j] * ox + f[7][j] * mx;
        const float f0312 = f03 * oy + f12 * my;
        const float f4756 = f47 * oy + f56 * my;
        r[j] = f0312 * oz + f4756 * mz;
    }
    float* __restrict__ o = out + (size_t)n * ld_out + col_off + l * F;
    if constexpr (F == 2) {
        *reinterpret_cast<float2*>(o) = make_float2(r[0], r[1]);
    } else {
        *reinterpret_cast<float4*>(o) = make_float4(r[0], r[1], r[2], r[3]);
        *reinterpret_cast<float4*>(o + 4) = make_float4(r[4], r[5], r[6], r[7]);
    }
}

template <int F>
__global__ __launch_bounds__(256) void k_hashgrid_bwd(const float* __restrict__ u, const float* __restrict__ grad_out,
                                                      const float* __restrict__ scalings, int N, int log2_T,
                                                      int ld_out, int col_off, float* __restrict__ grad_table) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int l = blockIdx.y;
    if (n >= N) return;
    const uint32_t mask = (1u << log2_T) - 1u;
    const Corners c = corners_of(u, n, scalings[l], mask);
    float g[F];
    load_row<F>(grad_out + (size_t)n * ld_out + col_off + l * F, g);
    const float ox = c.ox, oy = c.oy, oz = c.oz;
    const float mx = 1.f - ox, my = 1.f - oy, mz = 1.f - oz;
    // chain-rule weights in autograd's order: ((g*z)*y)*x
    float w[8];
    w[0] = oz * oy * ox;
    w[3] = oz * oy * mx;
    w[1] = oz * my * ox;
    w[2] = oz * my * mx;
    w[4] = mz * oy * ox;
    w[7] = mz * oy * mx;
    w[5] = mz * my * ox;
    w[6] = mz * my * mx;
    float* __restrict__ slab = grad_table + ((size_t)l << log2_T) * F;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (w[k] != 0.f) {
            float* __restrict__ dst = slab + (size_t)c.idx[k] * F;
#pragma unroll
            for (int j = 0; j < F; ++j) unsafeAtomicAdd(dst + j, w[k] * g[j]);
        }
    }
}

}  // namespace snf

using namespace snf;

static int check_common(const char* who, const void* u, const void* a, const void* sc, const void* b, int N, int L,
                        int F, int log2_T, int ld_out, int col_off) {
    SNF_REQUIRE(u && a && sc && b, "%s: null pointer", who);
    SNF_REQUIRE(N > 0 && L > 0 && L <= 65535, "%s: bad N=%d L=%d", who, N, L);
    SNF_REQUIRE(F == 2 || F == 8, "%s: features_per_level must be 2 or 8 (got %d)", who, F);
    SNF_REQUIRE(log2_T >= 1 && log2_T <= 26, "%s: bad log2_T=%d", who, log2_T);
    SNF_REQUIRE(ld_out >= col_off + L * F && col_off >= 0, "%s: ld_out=%d too small for col_off=%d + L*F=%d", who,
                ld_out, col_off, L * F);
    const int al = (F == 2) ? 2 : 4;
    SNF_REQUIRE(ld_out % al == 0 && col_off % al == 0, "%s: ld_out/col_off must be multiples of %d", who, al);
    return SNF_OK;
}

extern "C" int snf_hashgrid_fwd(const float* u, const float* table, const float* scalings, int N, int L, int F,
                                int log2_T, float* out, int ld_out, int col_off, snf_stream_t stream) {
    int rc = check_common("snf_hashgrid_fwd", u, table, scalings, out, N, L, F, log2_T, ld_out, col_off);
    if (rc) return rc;
    SNF_REQUIRE(((uintptr_t)table % 16) == 0 && ((uintptr_t)out % 16) == 0, "snf_hashgrid_fwd: unaligned pointer");
    dim3 grid(ceil_div(N, 256), L);
    if (F == 2)
        hipLaunchKernelGGL(k_hashgrid_fwd<2>, grid, dim3(256), 0, (hipStream_t)stream, u, table, scalings, N, log2_T,
                           out, ld_out, col_off);
    else
        hipLaunchKernelGGL(k_hashgrid_fwd<8>, grid, dim3(256), 0, (hipStream_t)stream, u, table, scalings, N, log2_T,
                           out, ld_out, col_off);
    SNF_LAUNCH_CHECK("snf_hashgrid_fwd");
    return SNF_OK;
}

extern "C" int snf_hashgrid_bwd(const float* u, const float* grad_out, const float* scalings, int N, int L, int F,
                                int log2_T, int ld_out, int col_off, float* grad_table, snf_stream_t stream) {
    int rc = check_common("snf_hashgrid_bwd", u, grad_out, scalings, grad_table, N, L, F, log2_T, ld_out, col_off);
    if (rc) return rc;
    SNF_REQUIRE(((uintptr_t)grad_out % 16) == 0, "snf_hashgrid_bwd: unaligned pointer");
    dim3 grid(ceil_div(N, 256), L);
    if (F == 2)
        hipLaunchKernelGGL(k_hashgrid_bwd<2>, grid, dim3(256), 0, (hipStream_t)stream, u, grad_out, scalings, N,
                           log2_T, ld_out, col_off, grad_table);
    else
        hipLaunchKernelGGL(k_hashgrid_bwd<8>, grid, dim3(256), 0, (hipStream_t)stream, u, grad_out, scalings, N,
                           log2_T, ld_out, col_off, grad_table);
    SNF_LAUNCH_CHECK("snf_hashgrid_bwd");
    return SNF_OK;
}

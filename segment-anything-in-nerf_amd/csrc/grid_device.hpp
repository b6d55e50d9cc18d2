// grid_device.hpp -- device helpers shared by the hash-grid kernels (hashgrid.hip) and the fused grid -> head-layer kernel
// (fused_head.hip): the corner indices / offsets of a sample at one level, and the bf16 hi + lo split of the 3-product GEMMs.
#pragma once
#include "common.hpp"
#include <math.h>

namespace snf {

constexpr uint32_t PRIME_Y = 2654435761u;
constexpr uint32_t PRIME_Z = 805459861u;

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


typedef __bf16 gd_bf16x2 __attribute__((ext_vector_type(2)));
typedef float gd_f32x2 __attribute__((ext_vector_type(2)));

// fp32 pair -> packed bf16 hi pair and packed bf16 lo pair (lo = bf16(x - hi)); v_cvt_pk_bf16_f32, round-to-nearest-even
__device__ __forceinline__ uint32_t gd_cvt_pk_bf16(float a, float b) {
    const gd_f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, gd_bf16x2));
}

__device__ __forceinline__ void gd_split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    hi = gd_cvt_pk_bf16(x0, x1);
    lo = gd_cvt_pk_bf16(x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xFFFF0000u));
}

}  // namespace snf

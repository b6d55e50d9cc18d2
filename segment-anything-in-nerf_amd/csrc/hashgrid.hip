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

// ==========================================================================================
// Bucketed backward: scatter-add without global atomics.
//
// Device-scope fp32 atomics are executed at the memory side of the fabric (the 8 XCD L2s are not coherent), one
// 4-byte request each: ~13 G atomics/s measured, i.e. 3.8 ms for one L12/F8 grid at 65k samples.  Instead the
// (sample, level, corner) contributions are counting-sorted by destination: a level's 2^T rows are cut into B = 256
// buckets of rpb = 2^T/256 consecutive rows; records (4 B: sample<<3 | corner) are partitioned by bucket in three
// streaming passes, then ONE workgroup per (bucket, level) re-derives index and weight from the record, gathers the
// upstream gradient, accumulates the bucket's rpb x F slab in LDS (ds_add_f32) and adds the slab to the gradient
// table with plain coalesced dwordx4 read-modify-writes -- no other workgroup owns those rows.
// ==========================================================================================
constexpr int HG_TILE = 1024;  // samples per workgroup in the count / scatter passes (4 per thread)

__device__ __forceinline__ void bucket_geometry(int log2_T, int& log2B, int& log2rpb) {
    log2B = log2_T < 8 ? log2_T : 8;
    log2rpb = log2_T - log2B;
}

__global__ __launch_bounds__(256) void k_hg_count(const float* __restrict__ u, const float* __restrict__ scalings, int N,
                                                  int log2_T, uint32_t* __restrict__ g_hist) {
    __shared__ uint32_t hist[256];
    int log2B, log2rpb;
    bucket_geometry(log2_T, log2B, log2rpb);
    const int B = 1 << log2B;
    const int tid = threadIdx.x, blk = blockIdx.x, l = blockIdx.y, nblk = gridDim.x;
    hist[tid] = 0;
    __syncthreads();
    const uint32_t mask = (1u << log2_T) - 1u;
    const float s = scalings[l];
#pragma unroll
    for (int j = 0; j < HG_TILE / 256; ++j) {
        const int n = blk * HG_TILE + j * 256 + tid;
        if (n < N) {
            const Corners c = corners_of(u, n, s, mask);
#pragma unroll
            for (int k = 0; k < 8; ++k) atomicAdd(&hist[c.idx[k] >> log2rpb], 1u);
        }
    }
    __syncthreads();
    if (tid < B) g_hist[((size_t)l * nblk + blk) * B + tid] = hist[tid];
}

// one workgroup per level: per-bucket exclusive scan over the tile histograms, then bucket bases
__global__ __launch_bounds__(256) void k_hg_scan(int N, int log2_T, int nblk, uint32_t* __restrict__ g_hist,
                                                 uint32_t* __restrict__ bucket_start) {
    __shared__ uint32_t tot[256];
    int log2B, log2rpb;
    bucket_geometry(log2_T, log2B, log2rpb);
    const int B = 1 << log2B;
    const int b = threadIdx.x, l = blockIdx.x;
    uint32_t total = 0;
    if (b < B) {
        for (int blk = 0; blk < nblk; ++blk) {
            uint32_t* p = &g_hist[((size_t)l * nblk + blk) * B + b];
            const uint32_t v = *p;
            *p = total;
            total += v;
        }
    }
    tot[b] = (b < B) ? total : 0u;
    __syncthreads();
    // Hillis-Steele inclusive scan over 256 entries
    for (int d = 1; d < 256; d <<= 1) {
        const uint32_t v = (b >= d) ? tot[b - d] : 0u;
        __syncthreads();
        tot[b] += v;
        __syncthreads();
    }
    if (b < B) {
        const uint32_t base = (uint32_t)((size_t)l * 8u * (uint32_t)N) + (tot[b] - total);
        bucket_start[l * (B + 1) + b] = base;
        if (b == B - 1) bucket_start[l * (B + 1) + B] = base + total;
        for (int blk = 0; blk < nblk; ++blk) g_hist[((size_t)l * nblk + blk) * B + b] += base;
    }
}

__global__ __launch_bounds__(256) void k_hg_scatter(const float* __restrict__ u, const float* __restrict__ scalings, int N,
                                                    int log2_T, const uint32_t* __restrict__ g_hist,
                                                    uint32_t* __restrict__ records) {
    __shared__ uint32_t cursor[256];
    int log2B, log2rpb;
    bucket_geometry(log2_T, log2B, log2rpb);
    const int B = 1 << log2B;
    const int tid = threadIdx.x, blk = blockIdx.x, l = blockIdx.y, nblk = gridDim.x;
    if (tid < B) cursor[tid] = g_hist[((size_t)l * nblk + blk) * B + tid];
    __syncthreads();
    const uint32_t mask = (1u << log2_T) - 1u;
    const float s = scalings[l];
#pragma unroll
    for (int j = 0; j < HG_TILE / 256; ++j) {
        const int n = blk * HG_TILE + j * 256 + tid;
        if (n < N) {
            const Corners c = corners_of(u, n, s, mask);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t slot = atomicAdd(&cursor[c.idx[k] >> log2rpb], 1u);
                records[slot] = ((uint32_t)n << 3) | (uint32_t)k;
            }
        }
    }
}

template <int F>
__global__ __launch_bounds__(256) void k_hg_reduce(const float* __restrict__ u, const float* __restrict__ grad_out,
                                                   const float* __restrict__ scalings, int log2_T, int ld_out, int col_off,
                                                   const uint32_t* __restrict__ bucket_start,
                                                   const uint32_t* __restrict__ records, float* __restrict__ grad_table) {
    extern __shared__ __attribute__((aligned(16))) float acc[];
    int log2B, log2rpb;
    bucket_geometry(log2_T, log2B, log2rpb);
    const int B = 1 << log2B;
    const int rpb = 1 << log2rpb;
    const int tid = threadIdx.x, b = blockIdx.x, l = blockIdx.y;
    const int lane = tid & 63;
    const uint32_t start = bucket_start[l * (B + 1) + b], end = bucket_start[l * (B + 1) + b + 1];
    if (start == end) return;  // nothing lands in this bucket: leave the slab untouched
    const int nacc = rpb * F;
    for (int i = tid; i < nacc; i += 256) acc[i] = 0.f;
    __syncthreads();
    const uint32_t mask = (1u << log2_T) - 1u;
    const float s = scalings[l];
    for (uint32_t i = start + tid; i < end; i += 256) {
        const uint32_t rec = records[i];
        const int n = (int)(rec >> 3);
        const int k = (int)(rec & 7u);
        // corner naming of encodings.py:318-325: x takes ceil for k in {0,1,4,5}, y for {0,3,4,7}, z for {0,1,2,3}
        const bool xc = (0x33u >> k) & 1u, yc = (0x99u >> k) & 1u, zc = (0x0Fu >> k) & 1u;
        float px, py, pz;
        {
#pragma clang fp contract(off)
            px = u[(size_t)n * 3 + 0] * s;
            py = u[(size_t)n * 3 + 1] * s;
            pz = u[(size_t)n * 3 + 2] * s;
        }
        const float fxf = floorf(px), fyf = floorf(py), fzf = floorf(pz);
        const float ox = px - fxf, oy = py - fyf, oz = pz - fzf;
        const uint32_t ix = (uint32_t)(int)(xc ? ceilf(px) : fxf);
        const uint32_t iy = (uint32_t)(int)(yc ? ceilf(py) : fyf) * PRIME_Y;
        const uint32_t iz = (uint32_t)(int)(zc ? ceilf(pz) : fzf) * PRIME_Z;
        const uint32_t row = (ix ^ iy ^ iz) & mask;
        const float wx = xc ? ox : 1.f - ox, wy = yc ? oy : 1.f - oy, wz = zc ? oz : 1.f - oz;
        const float w = wz * wy * wx;
        if (w != 0.f) {
            float g[F];
            load_row<F>(grad_out + (size_t)n * ld_out + col_off + l * F, g);
            float* dst = acc + (size_t)(row & (uint32_t)(rpb - 1)) * F;
            // feature order rotated by lane: spreads the 64 lanes of a ds_add over all LDS banks
#pragma unroll
            for (int j = 0; j < F; ++j) {
                const int f = (j + lane) & (F - 1);
                atomicAdd(dst + f, w * g[f]);
            }
        }
    }
    __syncthreads();
    float* __restrict__ slab = grad_table + (((size_t)l << log2_T) + ((size_t)b << log2rpb)) * F;
    if ((nacc & 3) == 0) {
        for (int i = tid; i < nacc / 4; i += 256) {
            const float4 a = reinterpret_cast<const float4*>(acc)[i];
            if (a.x != 0.f || a.y != 0.f || a.z != 0.f || a.w != 0.f) {
                float4 t = reinterpret_cast<float4*>(slab)[i];
                t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w;
                reinterpret_cast<float4*>(slab)[i] = t;
            }
        }
    } else {
        for (int i = tid; i < nacc; i += 256)
            if (acc[i] != 0.f) slab[i] += acc[i];
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

// ---- bucketed (atomic-free) backward ---------------------------------------------------------------------
static size_t hg_ws_words(int N, int L, int log2_T) {
    const int log2B = log2_T < 8 ? log2_T : 8;
    const size_t B = (size_t)1 << log2B;
    const size_t nblk = (size_t)ceil_div(N, HG_TILE);
    return (size_t)L * 8 * (size_t)N + (size_t)L * nblk * B + (size_t)L * (B + 1);
}

extern "C" int64_t snf_hashgrid_bwd_workspace_bytes(int N, int L, int log2_T) {
    if (N <= 0 || L <= 0 || log2_T < 1) return 0;
    return (int64_t)(hg_ws_words(N, L, log2_T) * sizeof(uint32_t));
}

extern "C" int snf_hashgrid_bwd_sorted(const float* u, const float* grad_out, const float* scalings, int N, int L, int F,
                                       int log2_T, int ld_out, int col_off, float* grad_table, void* workspace,
                                       int64_t workspace_bytes, snf_stream_t stream) {
    int rc = check_common("snf_hashgrid_bwd_sorted", u, grad_out, scalings, grad_table, N, L, F, log2_T, ld_out, col_off);
    if (rc) return rc;
    SNF_REQUIRE(((uintptr_t)grad_out % 16) == 0 && ((uintptr_t)grad_table % 16) == 0,
                "snf_hashgrid_bwd_sorted: unaligned pointer");
    const int log2B = log2_T < 8 ? log2_T : 8;
    const int log2rpb = log2_T - log2B;
    const size_t lds = ((size_t)F << log2rpb) * sizeof(float);
    SNF_REQUIRE((long long)N * 8 < (1LL << 31) / (L > 0 ? 1 : 1) && (long long)L * 8 * N < (1LL << 32),
                "snf_hashgrid_bwd_sorted: too many records for 32-bit offsets (N=%d L=%d)", N, L);
    SNF_REQUIRE(N < (1 << 29), "snf_hashgrid_bwd_sorted: N too large for the record packing");
    if (lds > 64 * 1024) {
        // slab does not fit LDS (log2_T > 19 at F=8): fall back to the atomic kernel
        return snf_hashgrid_bwd(u, grad_out, scalings, N, L, F, log2_T, ld_out, col_off, grad_table, stream);
    }
    SNF_REQUIRE(workspace && workspace_bytes >= (int64_t)(hg_ws_words(N, L, log2_T) * sizeof(uint32_t)),
                "snf_hashgrid_bwd_sorted: workspace too small (need %lld bytes)",
                (long long)(hg_ws_words(N, L, log2_T) * sizeof(uint32_t)));
    SNF_REQUIRE(((uintptr_t)workspace % 16) == 0, "snf_hashgrid_bwd_sorted: unaligned workspace");
    const int B = 1 << log2B;
    const int nblk = ceil_div(N, HG_TILE);
    uint32_t* records = (uint32_t*)workspace;
    uint32_t* hist = records + (size_t)L * 8 * (size_t)N;
    uint32_t* bstart = hist + (size_t)L * nblk * B;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_hg_count, dim3(nblk, L), dim3(256), 0, st, u, scalings, N, log2_T, hist);
    hipLaunchKernelGGL(k_hg_scan, dim3(L), dim3(256), 0, st, N, log2_T, nblk, hist, bstart);
    hipLaunchKernelGGL(k_hg_scatter, dim3(nblk, L), dim3(256), 0, st, u, scalings, N, log2_T, hist, records);
    if (F == 2)
        hipLaunchKernelGGL(k_hg_reduce<2>, dim3(B, L), dim3(256), lds, st, u, grad_out, scalings, log2_T, ld_out, col_off,
                           bstart, records, grad_table);
    else
        hipLaunchKernelGGL(k_hg_reduce<8>, dim3(B, L), dim3(256), lds, st, u, grad_out, scalings, log2_T, ld_out, col_off,
                           bstart, records, grad_table);
    SNF_LAUNCH_CHECK("snf_hashgrid_bwd_sorted");
    return SNF_OK;
}

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
#include "mlp_tiny_device.hpp"
#include <algorithm>
#include <type_traits>
#include "grid_device.hpp"
#include <math.h>
#include <stdlib.h>

namespace snf {


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

// Adam moments of a fused backward + Adam: read once and written once per step, never gathered -- SNF_HG_NT_MV = 1 marks those
// accesses non-temporal (bit 0: loads, bit 1: stores), a compile-time probe (round 1 measured +9 % for p / m / v together)
#ifndef SNF_HG_NT_MV
#define SNF_HG_NT_MV 0
#endif
typedef float hg_f4v __attribute__((ext_vector_type(4)));
typedef float hg_f2v __attribute__((ext_vector_type(2)));
typedef unsigned int hg_u4v __attribute__((ext_vector_type(4)));
typedef unsigned int hg_u2v __attribute__((ext_vector_type(2)));
// Cache-policy probes (round 5).  The sort records and the Adam state are read once / written once per step, while the staged
// gradient slab of a level (4 MB at N = 2^19: one L2) is gathered from by every workgroup of the level: marked non-temporal, the
// streams should stop evicting the slab.  Bits: 1 record loads, 2 p/m/v loads, 4 p/m/v stores (F = 2 fixed-point reduce).  (Non-temporal
// record STORES in the x-pair scatter measured 40 % slower, non-temporal record loads in the F = 8 float reduce noise: not kept.)
#ifndef SNF_FX_NT
#define SNF_FX_NT 7  // measured, same box: field-grid backward 0.340 -> 0.327 ms serial, step 2.465 -> 2.438 ms (profiles/EXPERIMENTS.md r05)
#endif

__device__ __forceinline__ uint4 hg_ld_rec(const uint4* p, bool nt) {
    if (nt) {
        const hg_u4v v = __builtin_nontemporal_load(reinterpret_cast<const hg_u4v*>(p));
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    return *p;
}
__device__ __forceinline__ uint2 hg_ld_rec(const uint2* p, bool nt) {
    if (nt) {
        const hg_u2v v = __builtin_nontemporal_load(reinterpret_cast<const hg_u2v*>(p));
        return make_uint2(v.x, v.y);
    }
    return *p;
}

template <int F>
__device__ __forceinline__ void load_row_mv(const float* __restrict__ p, float (&o)[F]) {
    if constexpr (F == 8 && (SNF_HG_NT_MV & 1)) {
        const hg_f4v a = __builtin_nontemporal_load(reinterpret_cast<const hg_f4v*>(p));
        const hg_f4v b = __builtin_nontemporal_load(reinterpret_cast<const hg_f4v*>(p + 4));
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
        o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    } else {
        load_row<F>(p, o);
    }
}
__device__ __forceinline__ void store_row8_mv(float* __restrict__ p, const float (&v)[8]) {
    if constexpr ((SNF_HG_NT_MV & 2) != 0) {
        const hg_f4v a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
        __builtin_nontemporal_store(a, reinterpret_cast<hg_f4v*>(p));
        __builtin_nontemporal_store(b, reinterpret_cast<hg_f4v*>(p + 4));
    } else {
        reinterpret_cast<float4*>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
        reinterpret_cast<float4*>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
}

constexpr int HG_MAX_LOG2B = 12;  // most buckets per level of the bucketed backward (its section below)

// The 8 corners of a sample as 4 x-neighbour pairs (corner order of corners_of: (0,3) (1,2) (4,7) (5,6) differ in x only).  The two
// rows of a pair differ in their low bits only (x has the hash's unit prime; dense levels: adjacent rows), so they share a bucket
// unless the pair straddles a bucket boundary (1 in rows-per-bucket): one LDS atomic of 2 instead of two of 1 -- the count and
// scatter passes are bound by their LDS atomics.
__device__ __forceinline__ void hg_count_corners(const Corners& c, int log2rpb, uint32_t* __restrict__ hist) {
    constexpr int PA[4] = {0, 1, 4, 5}, PB[4] = {3, 2, 7, 6};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const uint32_t ba = c.idx[PA[p]] >> log2rpb, bb = c.idx[PB[p]] >> log2rpb;
        atomicAdd(&hist[ba], ba == bb ? 2u : 1u);
        if (ba != bb) atomicAdd(&hist[bb], 1u);
    }
}

// The trilinear blend, a x + b (1 - x) along x, then y, then z (encodings.py:327-337), with its roundings spelled out: the first
// product rounded, the second fused into the sum -- what the compiler made of `a * o + b * m` in k_hashgrid_fwd; written with fmaf so
// that every kernel that blends through it rounds the same way whatever the compiler contracts around it (a second forward kernel
// with the very same source differed from k_hashgrid_fwd in the last bit of 59 % of its outputs, round 5).
__device__ __forceinline__ float hg_blend(float f0, float f1, float f2, float f3, float f4, float f5, float f6, float f7, float ox,
                                          float oy, float oz, float mx, float my, float mz) {
    const float f03 = fmaf(f3, mx, f0 * ox);
    const float f12 = fmaf(f2, mx, f1 * ox);
    const float f56 = fmaf(f6, mx, f5 * ox);
    const float f47 = fmaf(f7, mx, f4 * ox);
    const float f0312 = fmaf(f12, my, f03 * oy);
    const float f4756 = fmaf(f56, my, f47 * oy);
    return fmaf(f4756, mz, f0312 * oz);
}

// one (sample, level): gather the 8 corner rows, trilinear blend in the reference's order, store
template <int F>
__device__ __forceinline__ void hg_eval(const Corners& c, const float* __restrict__ slab, float (&r)[F]) {
    float f[8][F];
#define SNF_HG_FWD_PAIR 1
    if constexpr (F == 2 && SNF_HG_FWD_PAIR) {
        // The x-neighbours of a cell hash to (fx ^ h) and ((fx + 1) ^ h) with the same h = y P1 ^ z P2: for EVEN fx they differ in
        // bit 0 only -- an aligned PAIR of 8-byte rows, one 16-byte load instead of two 8-byte ones.  The kernel is bound by the
        // texture addresser (rocprofv3: TA busy 80 % of the kernel, ~1 scattered address per clock and CU; L1 hits 67 %, L2 hits
        // 74 %, profiles/r03_f2_counters.txt), so half of the samples issue 4 addresses instead of 8.  (px integral: cx == fx,
        // the indices are equal, not a pair -- the 8-load path reads the row twice as before.)
        const bool pair = (c.idx[0] ^ c.idx[3]) == 1u;  // cx ^ fx == 1: the same for the four (y, z) combinations
        if (pair) {
            // corner pairs (cx, fx) with equal y, z: (0,3) (1,2) (5,6) (4,7).  The aligned pair starts at the EVEN one of the two
            // rows -- which of them that is depends on the parity of h, i.e. on the (y, z) combination
            auto load_pair = [&](int kc, int kf) {
                const uint32_t rf = c.idx[kf];
                const float4 t = *reinterpret_cast<const float4*>(slab + (size_t)(rf & ~1u) * 2);
                const bool f_first = !(rf & 1u);
                f[kf][0] = f_first ? t.x : t.z; f[kf][1] = f_first ? t.y : t.w;
                f[kc][0] = f_first ? t.z : t.x; f[kc][1] = f_first ? t.w : t.y;
            };
            load_pair(0, 3);
            load_pair(1, 2);
            load_pair(5, 6);
            load_pair(4, 7);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) load_row<F>(slab + (size_t)c.idx[k] * F, f[k]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) load_row<F>(slab + (size_t)c.idx[k] * F, f[k]);
    }
    const float ox = c.ox, oy = c.oy, oz = c.oz;
    const float mx = 1.f - ox, my = 1.f - oy, mz = 1.f - oz;
#pragma unroll
    for (int j = 0; j < F; ++j)
        r[j] = hg_blend(f[0][j], f[1][j], f[2][j], f[3][j], f[4][j], f[5][j], f[6][j], f[7][j], ox, oy, oz, mx, my, mz);
}

template <int F>
__device__ __forceinline__ void hg_fwd_one(const Corners& c, const float* __restrict__ slab, int n, int l, int N,
                                           float* __restrict__ out, int ld_out, int col_off) {
    float r[F];
    hg_eval<F>(c, slab, r);
    // ld_out == 0: level-major ("planar") output [L][N][F] -- consecutive samples of a level are contiguous, so a wave
    // writes whole lines (row-major [N, L*F] output is 8 / 32 bytes per 128-byte row from a level-at-a-time kernel:
    // rocprofv3 WRITE_SIZE showed 4x the algorithmic bytes for the F=2 field grid)
    float* __restrict__ o = ld_out ? out + (size_t)n * ld_out + col_off + l * F : out + ((size_t)l * N + n) * F;
    if constexpr (F == 2) {
        *reinterpret_cast<float2*>(o) = make_float2(r[0], r[1]);
    } else {
        *reinterpret_cast<float4*>(o) = make_float4(r[0], r[1], r[2], r[3]);
        *reinterpret_cast<float4*>(o + 4) = make_float4(r[4], r[5], r[6], r[7]);
    }
}

template <int F>
__global__ __launch_bounds__(256) void k_hashgrid_fwd(const float* __restrict__ u, const float* __restrict__ table,
                                                      const float* __restrict__ scalings, int N, int log2_T,
                                                      float* __restrict__ out, int ld_out, int col_off, int l0) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int l = blockIdx.y + l0;
    if (n >= N) return;
    const uint32_t mask = (1u << log2_T) - 1u;
    const Corners c = corners_of(u, n, scalings[l], mask);
    hg_fwd_one<F>(c, table + ((size_t)l << log2_T) * F, n, l, N, out, ld_out, col_off);
}

// The eval render's proposal stage in one kernel (density_fields.py:99-127 without gradients): the L = 5 levels of the F = 2 proposal
// grid, the 10 -> 16 -> 1 density net and trunc_exp per sample, in registers -- snf_hashgrid_fwd + snf_mlp_tiny_fwd + snf_trunc_exp_fwd
// wrote and re-read a [N, 10] encoding (8 bytes per 40-byte row from each level's pass) that nothing else looks at.  Same device
// functions, same values.
template <int L>
__global__ __launch_bounds__(256) void k_prop_density_fwd(const float* __restrict__ u, const float* __restrict__ table,
                                                          const float* __restrict__ scalings, int N, int log2_T,
                                                          const float* __restrict__ W0, const float* __restrict__ W1,
                                                          const uint8_t* __restrict__ selector, float* __restrict__ density) {
    constexpr int I = 2 * L, H = 16;
    __shared__ float w0[H * I], w1[H], sc[L];
    for (int i = threadIdx.x; i < H * I; i += 256) w0[i] = W0[i];
    if (threadIdx.x < H) w1[threadIdx.x] = W1[threadIdx.x];
    if (threadIdx.x < L) sc[threadIdx.x] = scalings[threadIdx.x];
    __syncthreads();
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const uint32_t mask = (1u << log2_T) - 1u;
    float x[I];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const Corners c = corners_of(u, n, sc[l], mask);
        float r[2];
        hg_eval<2>(c, table + ((size_t)l << log2_T) * 2, r);
        x[2 * l] = r[0];
        x[2 * l + 1] = r[1];
    }
    float h[H];
    float d = expf(mt_forward<I, H>(x, w0, w1, h));
    if (selector) d *= (float)selector[n];
    density[n] = d;
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
// Bucketed backward: scatter-add without global atomics and without float atomics at all.
//
// Why: device-scope fp32 atomics execute at the memory side of the fabric (the 8 XCD L2s are not coherent), one
// 4-byte request each: ~13 G atomics/s measured = 3.8 ms for ONE L12/F8 grid at 65k samples.  LDS float atomics are
// no way out either: ds_add_f32 retires 0.375 lane-ops/clk/CU on gfx950 against 10.1 for ds_add_u32
// (tools/ubench/lds_atomic_rate.hip).  So contributions are counting-sorted by destination with INTEGER atomics only:
//   stage   upstream gradient [N, ld] -> level-major gT[l][n][F] (compact, L2-resident per level)
//   count   per (tile of samples, level): histogram of the 8 corner rows over B buckets of rpb = 2^T/B consecutive rows
//   scan    per level: exclusive scan over tiles and buckets -> record offsets
//   scatter 4-byte records (sample<<3 | corner) written bucket by bucket
//   reduce  one workgroup per (bucket, level): counting-sort the bucket's records by row inside LDS (rank =
//           ds_add_rtn_u32), then rows with short segments are summed by their owner thread, rows with long segments
//           (coarse levels: hundreds of samples per cell) by a whole wave with a shuffle reduction; each row is added
//           to the gradient table by exactly one lane with a plain read-modify-write (rows are owned, no atomics).
// B is sized so that a bucket holds ~2048 records (8N/B), i.e. one LDS chunk.
// ==========================================================================================
constexpr int HG_RT = 512;       // threads of the reduce workgroup (8 waves; 4 workgroups per CU)
// records sorted per trip through LDS (16 KB of payload at F = 2, 64 KB at F = 8; a 1024-record chunk for F = 8 measured slower)
#define SNF_HG_CHUNK8 2048
template <int F> constexpr int hg_chunk() { return F == 8 ? SNF_HG_CHUNK8 : 2048; }
constexpr int HG_MAX_RPB = 2048; // rows per bucket (HG_ROWS_PT per reduce thread)
constexpr int HG_ROWS_PT = HG_MAX_RPB / HG_RT;
constexpr int HG_LONG = 48;      // segments longer than this are reduced by a wave (16 until the kernel ran two workgroups per CU:
                                 // 32 .. 64 then measured 5-8 % faster alone, tools/sweep_hg_long.sh)
// (round 5: all four rows of a thread in flight -- the row sums parked in the dead sort payload, buffer addressing so that 96 of the
//  128 registers hold rows -- measured SLOWER: 0.900 -> 0.935 ms per step serial with three rows in flight, 1.05 with four (spills),
//  and the concurrent step 2.55 -> 2.62 ms: this pass is not short of bytes in flight, profiles/EXPERIMENTS.md r05)
#ifndef SNF_HG_EPI
#define SNF_HG_EPI 2
#endif
constexpr int HG_EPI = SNF_HG_EPI;  // rows per thread in flight in the fused Adam epilogue of the float reduce

struct HgGeom {
    int log2B, log2rpb, spt, nblk;  // buckets, rows per bucket, samples per thread in count/scatter, tiles
};

inline HgGeom hg_geometry(int N, int log2_T) {
    HgGeom g;
    // 256 buckets per level: a tile then emits long runs per bucket (coalescing record writes) and big-N grids simply
    // take several LDS chunks per bucket; more buckets only when a bucket would exceed 2048 rows.
    int lb = 8;
    if (lb > log2_T) lb = log2_T;
    if (log2_T - lb > 11) lb = log2_T - 11;                             // rpb <= 2048
    g.log2B = lb;
    g.log2rpb = log2_T - lb;
    int spt = (N + 256 * 128 - 1) / (256 * 128);                       // aim for <= 128 tiles
    g.spt = spt < 4 ? 4 : (spt > 16 ? 16 : spt);
    g.nblk = (N + 256 * g.spt - 1) / (256 * g.spt);
    return g;
}

// XP ("x-pair records"): one record per PAIR of x-neighbour corners that share a bucket (two single records when they do not), see
// k_hg_scatter: the count is then one per pair.
template <bool XP>
__global__ __launch_bounds__(256) void k_hg_count(const float* __restrict__ u, const float* __restrict__ scalings, int N,
                                                  int log2_T, int log2B, int spt, uint32_t* __restrict__ g_hist) {
    __shared__ uint32_t hist[1 << HG_MAX_LOG2B];
    const int B = 1 << log2B, log2rpb = log2_T - log2B;
    const int tid = threadIdx.x, blk = blockIdx.x, l = blockIdx.y, nblk = gridDim.x;
    for (int i = tid; i < B; i += 256) hist[i] = 0;
    __syncthreads();
    const uint32_t mask = (1u << log2_T) - 1u;
    const float s = scalings[l];
    for (int j = 0; j < spt; ++j) {
        const int n = (blk * spt + j) * 256 + tid;
        if (n < N) {
            const Corners c = corners_of(u, n, s, mask);
            if constexpr (XP) {
                constexpr int PA[4] = {0, 1, 4, 5}, PB[4] = {3, 2, 7, 6};
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const uint32_t ba = c.idx[PA[p]] >> log2rpb, bb = c.idx[PB[p]] >> log2rpb;
                    atomicAdd(&hist[ba], 1u);
                    if (ba != bb) atomicAdd(&hist[bb], 1u);
                }
            } else {
                hg_count_corners(c, log2rpb, hist);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < B; i += 256) g_hist[((size_t)l * nblk + blk) * B + i] = hist[i];
}

// one workgroup per level: exclusive scan over tiles and buckets (hist -> offs) and the bucket bases.
// 1024 threads = 4 tile-partitions x 256 bucket columns; column c owns the B/256 consecutive buckets [c*bpt, (c+1)*bpt).
// Input and output arrays are distinct so a thread's tile loads are independent and stay in flight together.
__global__ __launch_bounds__(1024) void k_hg_scan(int N, int log2B, int nblk, const uint32_t* __restrict__ g_hist,
                                                  uint32_t* __restrict__ g_offs, uint32_t* __restrict__ bucket_start) {
    __shared__ uint32_t part_tot[4][256][16];  // [tile partition][column][bucket within column]
    __shared__ uint32_t tot[256];
    const int B = 1 << log2B;
    const int c = threadIdx.x & 255, part = threadIdx.x >> 8, l = blockIdx.x;
    const int bpt = (B + 255) / 256;  // 1 (B <= 256) .. 16
    const int t0 = (int)((long long)nblk * part / 4), t1 = (int)((long long)nblk * (part + 1) / 4);
    for (int q = 0; q < bpt; ++q) {
        const int b = c * bpt + q;
        uint32_t total = 0;
        if (b < B) {
#pragma unroll 8
            for (int blk = t0; blk < t1; ++blk) total += g_hist[((size_t)l * nblk + blk) * B + b];
        }
        part_tot[part][c][q] = total;
    }
    __syncthreads();
    uint32_t mine = 0;
    if (part == 0) {
        for (int q = 0; q < bpt; ++q)
            mine += part_tot[0][c][q] + part_tot[1][c][q] + part_tot[2][c][q] + part_tot[3][c][q];
        tot[c] = mine;
    }
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {  // Hillis-Steele inclusive scan over the 256 columns
        uint32_t v = 0;
        if (part == 0 && c >= d) v = tot[c - d];
        __syncthreads();
        if (part == 0) tot[c] += v;
        __syncthreads();
    }
    // column base = exclusive prefix of column totals; within the column, buckets and tile partitions in order
    uint32_t colsum = 0;
    for (int q = 0; q < bpt; ++q)
        colsum += part_tot[0][c][q] + part_tot[1][c][q] + part_tot[2][c][q] + part_tot[3][c][q];
    uint32_t base = (uint32_t)((size_t)l * 8u * (uint32_t)N) + (tot[c] - colsum);
    for (int q = 0; q < bpt; ++q) {
        const int b = c * bpt + q;
        const uint32_t btot = part_tot[0][c][q] + part_tot[1][c][q] + part_tot[2][c][q] + part_tot[3][c][q];
        if (b < B) {
            if (part == 0) {
                bucket_start[l * (B + 1) + b] = base;
                if (b == B - 1) bucket_start[l * (B + 1) + B] = base + btot;
            }
            uint32_t run = base;
            for (int pp = 0; pp < part; ++pp) run += part_tot[pp][c][q];
#pragma unroll 8
            for (int blk = t0; blk < t1; ++blk) {
                const size_t idx = ((size_t)l * nblk + blk) * B + b;
                g_offs[idx] = run;
                run += g_hist[idx];
            }
        }
        base += btot;
    }
}

// record = 8 bytes { key = sample | (row_in_bucket << 21), corner weight w }: one aligned dwordx2 store per record in
// the scatter pass; the reduce pass needs no position gather and no index math, only the staged-gradient gather.
// (Payload-carrying 12/36-byte records were measured: the scatter's store-transaction count made them slower.)
constexpr int HG_SAMPLE_BITS = 21;  // N <= 2^21 per launch (row_in_bucket needs 11 bits)

// The scatter pass stages records through LDS: a batch of HG_SB_SAMPLES samples per thread-block trip (x 8 corners) is
// counting-sorted by bucket in LDS and leaves as one contiguous run per bucket (~128 B at 256 buckets) instead of 8-byte
// stores to 256 different cache lines -- the store-transaction count, not the byte count, bounded the direct version
// (rocprofv3: 84 % of its wave cycles were issue stalls behind the store queue).
#ifndef SNF_HG_SB_SPT
#define SNF_HG_SB_SPT 2
#endif
constexpr int HG_SB_SPT = SNF_HG_SB_SPT;          // samples per thread per batch (1: 64-B runs, 7 workgroups per CU; 4: 256-B runs, 1)
constexpr int HG_SB_REC = 256 * HG_SB_SPT * 8;    // 4096 staged records (32 KB)

inline size_t hg_scatter_lds_bytes(int log2B) {
    return ((size_t)3 << log2B) * sizeof(uint32_t) + (size_t)HG_SB_REC * (sizeof(uint2) + sizeof(uint16_t));
}

// SELF: the tile computes its own record offsets from the tile histograms (bucket totals over all tiles -> exclusive scan over
// the buckets -> plus what the tiles before it put into each bucket) instead of reading them from a scan kernel's output: the
// whole level's histogram is nblk x B words (64 KB at 64 tiles) from the L2, and the sort is count + scatter.  The tile blk == 0
// writes the level's bucket_start row.
template <bool SELF>
__global__ __launch_bounds__(256) void k_hg_scatter(const float* __restrict__ u, const float* __restrict__ scalings, int N,
                                                    int log2_T, int log2B, int spt, const uint32_t* __restrict__ g_offs,
                                                    uint2* __restrict__ records, uint32_t* __restrict__ bucket_start) {
    extern __shared__ __attribute__((aligned(16))) uint32_t hs_lds[];
    const int B = 1 << log2B, log2rpb = log2_T - log2B;
    uint2* stage = reinterpret_cast<uint2*>(hs_lds);                        // [HG_SB_REC]
    uint32_t* cursor = hs_lds + 2 * HG_SB_REC;                               // [B] global write cursor of this tile
    uint32_t* lcnt = cursor + B;                                             // [B] batch-local count -> exclusive offset
    uint32_t* delta = lcnt + B;                                              // [B] cursor - local offset
    uint16_t* sbkt = reinterpret_cast<uint16_t*>(delta + B);                 // [HG_SB_REC] bucket of a staged record
    __shared__ uint32_t wave_tot[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int blk = blockIdx.x, l = blockIdx.y, nblk = gridDim.x;
    const int bpt = (B + 255) >> 8;  // buckets per thread in the scans (1 .. 16): thread t owns buckets [t * bpt, (t + 1) * bpt)
    if constexpr (SELF) {
        // g_offs = the tile histograms g_hist[l][tile][bucket] here
        uint32_t tot[16], bef[16], sum = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int bb = tid * bpt + q;
            uint32_t t = 0, f = 0;
            if (q < bpt && bb < B) {
                const uint32_t* __restrict__ col = g_offs + (size_t)l * nblk * B + bb;
#pragma unroll 16
                for (int t2 = 0; t2 < nblk; ++t2) {
                    const uint32_t h = col[(size_t)t2 * B];
                    t += h;
                    f += t2 < blk ? h : 0u;
                }
            }
            tot[q] = t; bef[q] = f; sum += t;
        }
        uint32_t inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        if (lane == 63) wave_tot[wave] = inc;
        __syncthreads();
        uint32_t run = (uint32_t)((size_t)l * 8u * (uint32_t)N) + inc - sum;
        for (int w2 = 0; w2 < wave; ++w2) run += wave_tot[w2];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int bb = tid * bpt + q;
            if (q < bpt && bb < B) {
                cursor[bb] = run + bef[q];
                lcnt[bb] = 0u;
                if (blk == 0) {
                    bucket_start[l * (B + 1) + bb] = run;
                    if (bb == B - 1) bucket_start[l * (B + 1) + B] = run + tot[q];
                }
            }
            run += tot[q];
        }
        __syncthreads();  // (wave_tot is reused by the batch scans below)
    } else {
        for (int i = tid; i < B; i += 256) {
            cursor[i] = g_offs[((size_t)l * nblk + blk) * B + i];
            lcnt[i] = 0u;
        }
    }
    __syncthreads();
    const uint32_t mask = (1u << log2_T) - 1u, rmask = (1u << log2rpb) - 1u;
    const float s = scalings[l];
    for (int j0 = 0; j0 < spt; j0 += HG_SB_SPT) {
        // ---- 1: records of this batch, ranked within their bucket
        uint2 rec[HG_SB_SPT * 8];
        uint32_t bk[HG_SB_SPT * 8], rk[HG_SB_SPT * 8];
#pragma unroll
        for (int jj = 0; jj < HG_SB_SPT; ++jj) {
            const int j = j0 + jj;
            const int n = (blk * spt + j) * 256 + tid;
            const bool live = (j < spt) && (n < N);
            if (live) {
                const Corners c = corners_of(u, n, s, mask);
                const float ox = c.ox, oy = c.oy, oz = c.oz;
                const float mx = 1.f - ox, my = 1.f - oy, mz = 1.f - oz;
                float w[8];  // chain-rule weights in autograd's order: ((g*z)*y)*x
                w[0] = oz * oy * ox; w[3] = oz * oy * mx; w[1] = oz * my * ox; w[2] = oz * my * mx;
                w[4] = mz * oy * ox; w[7] = mz * oy * mx; w[5] = mz * my * ox; w[6] = mz * my * mx;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int q = jj * 8 + k;
                    bk[q] = c.idx[k] >> log2rpb;
                    rec[q] = make_uint2((uint32_t)n | ((c.idx[k] & rmask) << HG_SAMPLE_BITS), __float_as_uint(w[k]));
                }
                constexpr int PA[4] = {0, 1, 4, 5}, PB[4] = {3, 2, 7, 6};  // x-neighbour pairs: one atomic when they share a bucket
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int qa = jj * 8 + PA[p], qb = jj * 8 + PB[p];
                    const bool same = bk[qa] == bk[qb];
                    rk[qa] = atomicAdd(&lcnt[bk[qa]], same ? 2u : 1u);
                    rk[qb] = rk[qa] + 1u;
                    if (!same) rk[qb] = atomicAdd(&lcnt[bk[qb]], 1u);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) bk[jj * 8 + k] = 0xFFFFFFFFu;
            }
        }
        __syncthreads();
        // ---- 2: exclusive scan of the bucket counts (thread t owns buckets [t*bpt, (t+1)*bpt))
        uint32_t cq[16];
        uint32_t sum = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int bb = tid * bpt + q;
            cq[q] = (q < bpt && bb < B) ? lcnt[bb] : 0u;
            sum += cq[q];
        }
        uint32_t inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        if (lane == 63) wave_tot[wave] = inc;
        __syncthreads();
        uint32_t run = inc - sum;
        for (int w2 = 0; w2 < wave; ++w2) run += wave_tot[w2];
        const uint32_t total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int bb = tid * bpt + q;
            if (q < bpt && bb < B) {
                lcnt[bb] = run;
                delta[bb] = cursor[bb] - run;
            }
            run += cq[q];
        }
        __syncthreads();
        // ---- 3: records to their bucket-sorted slots
#pragma unroll
        for (int q = 0; q < HG_SB_SPT * 8; ++q) {
            if (bk[q] != 0xFFFFFFFFu) {
                const uint32_t slot = lcnt[bk[q]] + rk[q];
                stage[slot] = rec[q];
                sbkt[slot] = (uint16_t)bk[q];
            }
        }
        __syncthreads();
        // ---- 4: contiguous runs out; advance the cursors
        for (uint32_t i = tid; i < total; i += 256) records[delta[sbkt[i]] + i] = stage[i];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int bb = tid * bpt + q;
            if (q < bpt && bb < B) {
                cursor[bb] += cq[q];
                lcnt[bb] = 0u;
            }
        }
        __syncthreads();
    }
}

// ---- x-pair records (F = 2 grids) -----------------------------------------------------------------------------------------------
// The two x-neighbour corners of a sample -- rows h(x) and h(x + 1) of the level: the unit prime of x leaves them different in their
// low bits only -- fall into the same 2048-row bucket for all but 2^-11 of the pairs, and both take the SAME staged gradient in the
// reduce.  One 16-byte record per pair
//     { sample | row_a << 21,   row_b | b_valid << 11,   w_a,   w_b }
// lets one reduce lane serve two rows from ONE gather: the reduce is bound by the texture addresser, which counts gather instructions
// (not their active lanes: sharing a gather between the lanes of a pair gained nothing), and the scatter ranks and stages half
// as many records for the same bytes.  A pair that straddles two buckets leaves as two a-only records.  The level's record region
// holds 8 N records in the worst case (every pair split), as before -- 16 bytes each here.
#ifndef SNF_HG_XP_SPT
#define SNF_HG_XP_SPT SNF_HG_SB_SPT
#endif
constexpr int HG_XP_SPT = SNF_HG_XP_SPT;  // samples per thread per batch of the x-pair scatter (A/B: its own switch)
constexpr int HG_XP_CAP = 256 * HG_XP_SPT * 4 + 256;  // records staged per batch (4 per sample, + room for split pairs); the rest go direct

inline size_t hg_scatter_xp_lds_bytes(int log2B) {
    return ((size_t)3 << log2B) * sizeof(uint32_t) + (size_t)HG_XP_CAP * (sizeof(uint4) + sizeof(uint16_t));
}

template <bool SELF>
__global__ __launch_bounds__(256) void k_hg_scatter_xp(const float* __restrict__ u, const float* __restrict__ scalings, int N,
                                                       int log2_T, int log2B, int spt, const uint32_t* __restrict__ g_offs,
                                                       uint4* __restrict__ records, uint32_t* __restrict__ bucket_start) {
    extern __shared__ __attribute__((aligned(16))) uint32_t hx_lds[];
    const int B = 1 << log2B, log2rpb = log2_T - log2B;
    uint4* stage = reinterpret_cast<uint4*>(hx_lds);                         // [HG_XP_CAP]
    uint32_t* cursor = hx_lds + 4 * HG_XP_CAP;                               // [B] global write cursor of this tile
    uint32_t* lcnt = cursor + B;                                             // [B] batch-local count -> exclusive offset
    uint32_t* delta = lcnt + B;                                              // [B] cursor - local offset
    uint16_t* sbkt = reinterpret_cast<uint16_t*>(delta + B);                 // [HG_XP_CAP] bucket of a staged record
    __shared__ uint32_t wave_tot[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int blk = blockIdx.x, l = blockIdx.y, nblk = gridDim.x;
    const int bpt = (B + 255) >> 8;  // buckets per thread in the scans (1 .. 16): thread t owns buckets [t * bpt, (t + 1) * bpt)
    if constexpr (SELF) {
        uint32_t tot[16], bef[16], sum = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int bb = tid * bpt + q;
            uint32_t t = 0, f = 0;
            if (q < bpt && bb < B) {
                const uint32_t* __restrict__ col = g_offs + (size_t)l * nblk * B + bb;
#pragma unroll 16
                for (int t2 = 0; t2 < nblk; ++t2) {
                    const uint32_t h = col[(size_t)t2 * B];
                    t += h;
                    f += t2 < blk ? h : 0u;
                }
            }
            tot[q] = t; bef[q] = f; sum += t;
        }
        uint32_t inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        if (lane == 63) wave_tot[wave] = inc;
        __syncthreads();
        uint32_t run = (uint32_t)((size_t)l * 8u * (uint32_t)N) + inc - sum;
        for (int w2 = 0; w2 < wave; ++w2) run += wave_tot[w2];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int bb = tid * bpt + q;
            if (q < bpt && bb < B) {
                cursor[bb] = run + bef[q];
                lcnt[bb] = 0u;
                if (blk == 0) {
                    bucket_start[l * (B + 1) + bb] = run;
                    if (bb == B - 1) bucket_start[l * (B + 1) + B] = run + tot[q];
                }
            }
            run += tot[q];
        }
        __syncthreads();
    } else {
        for (int i = tid; i < B; i += 256) {
            cursor[i] = g_offs[((size_t)l * nblk + blk) * B + i];
            lcnt[i] = 0u;
        }
    }
    __syncthreads();
    const uint32_t mask = (1u << log2_T) - 1u, rmask = (1u << log2rpb) - 1u;
    const float s = scalings[l];
    constexpr int NP = HG_XP_SPT * 4;  // pairs per thread and batch
    for (int j0 = 0; j0 < spt; j0 += HG_XP_SPT) {
        // ---- 1: pair records of this batch, ranked within their bucket (a split pair ranks its b corner in b's bucket too)
        uint4 rec[NP];
        uint32_t bka[NP], rka[NP], rkb[NP / 2];  // (rkb: two 16-bit ranks per word; only read for split pairs)
        uint32_t split = 0u;
#pragma unroll
        for (int q = 0; q < NP / 2; ++q) rkb[q] = 0u;
#pragma unroll
        for (int jj = 0; jj < HG_XP_SPT; ++jj) {
            const int j = j0 + jj;
            const int n = (blk * spt + j) * 256 + tid;
            const bool live = (j < spt) && (n < N);
            if (live) {
                const Corners c = corners_of(u, n, s, mask);
                const float ox = c.ox, oy = c.oy, oz = c.oz;
                const float mx = 1.f - ox, my = 1.f - oy, mz = 1.f - oz;
                float w[8];  // chain-rule weights in autograd's order: ((g*z)*y)*x
                w[0] = oz * oy * ox; w[3] = oz * oy * mx; w[1] = oz * my * ox; w[2] = oz * my * mx;
                w[4] = mz * oy * ox; w[7] = mz * oy * mx; w[5] = mz * my * ox; w[6] = mz * my * mx;
                constexpr int PA[4] = {0, 1, 4, 5}, PB[4] = {3, 2, 7, 6};
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int q = jj * 4 + p;
                    const uint32_t ia = c.idx[PA[p]], ib = c.idx[PB[p]];
                    const uint32_t ba = ia >> log2rpb, bb = ib >> log2rpb;
                    const bool same = ba == bb;
                    bka[q] = ba;
                    // y: row_b | b_valid << 11 | bucket_b << 12 (the bucket only matters for the second record of a split pair)
                    rec[q] = make_uint4((uint32_t)n | ((ia & rmask) << HG_SAMPLE_BITS), (ib & rmask) | (same ? 0x800u : 0u) | (bb << 12),
                                        __float_as_uint(w[PA[p]]), __float_as_uint(w[PB[p]]));
                    rka[q] = atomicAdd(&lcnt[ba], 1u);
                    if (!same) {
                        split |= 1u << q;
                        rkb[q >> 1] |= atomicAdd(&lcnt[bb], 1u) << (16 * (q & 1));
                    }
                }
            } else {
#pragma unroll
                for (int p = 0; p < 4; ++p) bka[jj * 4 + p] = 0xFFFFFFFFu;
            }
        }
        __syncthreads();
        // ---- 2: exclusive scan of the bucket counts
        uint32_t cq[16];
        uint32_t sum = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int bb = tid * bpt + q;
            cq[q] = (q < bpt && bb < B) ? lcnt[bb] : 0u;
            sum += cq[q];
        }
        uint32_t inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        if (lane == 63) wave_tot[wave] = inc;
        __syncthreads();
        uint32_t run = inc - sum;
        for (int w2 = 0; w2 < wave; ++w2) run += wave_tot[w2];
        const uint32_t total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int bb = tid * bpt + q;
            if (q < bpt && bb < B) {
                lcnt[bb] = run;
                delta[bb] = cursor[bb] - run;
            }
            run += cq[q];
        }
        __syncthreads();
        // ---- 3: records to their bucket-sorted slots: staged when the slot lies inside the stage, straight to memory otherwise
        auto place = [&](uint32_t bk, uint32_t rk, const uint4& r) {
            const uint32_t slot = lcnt[bk] + rk;
            if (slot < (uint32_t)HG_XP_CAP) {
                stage[slot] = r;
                sbkt[slot] = (uint16_t)bk;
            } else {
                records[delta[bk] + slot] = r;
            }
        };
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            if (bka[q] != 0xFFFFFFFFu) {
                uint4 r = rec[q];
                const bool sp = (split >> q) & 1u;
                const uint32_t bb = r.y >> 12;
                r.y &= 0xFFFu;
                place(bka[q], rka[q], r);
                if (sp) {  // the b corner alone, as the a corner of a record in its own bucket
                    const uint32_t n = r.x & ((1u << HG_SAMPLE_BITS) - 1u);
                    place(bb, (rkb[q >> 1] >> (16 * (q & 1))) & 0xFFFFu, make_uint4(n | ((r.y & 0x7FFu) << HG_SAMPLE_BITS), 0u, r.w, 0u));
                }
            }
        }
        __syncthreads();
        // ---- 4: contiguous runs out; advance the cursors
        const uint32_t staged = total < (uint32_t)HG_XP_CAP ? total : (uint32_t)HG_XP_CAP;
        for (uint32_t i = tid; i < staged; i += 256) records[delta[sbkt[i]] + i] = stage[i];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int bb = tid * bpt + q;
            if (q < bpt && bb < B) {
                cursor[bb] += cq[q];
                lcnt[bb] = 0u;
            }
        }
        __syncthreads();
    }
}

// upstream gradient [N, ld] (this grid's L*F columns) -> level-major staging gT[l][n][F]
template <int F>
__global__ __launch_bounds__(256) void k_hg_stage_grad(const float* __restrict__ grad_out, int N, int L, int ld_out,
                                                       int col_off, float* __restrict__ gT) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)N * L) return;
    const int n = (int)(t / L), l = (int)(t - (long long)n * L);
    float g[F];
    load_row<F>(grad_out + (size_t)n * ld_out + col_off + l * F, g);
    float* o = gT + ((size_t)l * N + n) * F;
    if constexpr (F == 2) {
        *reinterpret_cast<float2*>(o) = make_float2(g[0], g[1]);
    } else {
        *reinterpret_cast<float4*>(o) = make_float4(g[0], g[1], g[2], g[3]);
        *reinterpret_cast<float4*>(o + 4) = make_float4(g[4], g[5], g[6], g[7]);
    }
}

template <int F>
__device__ __forceinline__ void row_rmw(float* __restrict__ dst, const float (&a)[F]) {
    if constexpr (F == 8) {
        float4 t0 = reinterpret_cast<float4*>(dst)[0], t1 = reinterpret_cast<float4*>(dst)[1];
        t0.x += a[0]; t0.y += a[1]; t0.z += a[2]; t0.w += a[3];
        t1.x += a[4]; t1.y += a[5]; t1.z += a[6]; t1.w += a[7];
        reinterpret_cast<float4*>(dst)[0] = t0;
        reinterpret_cast<float4*>(dst)[1] = t1;
    } else {
        float2 t0 = *reinterpret_cast<float2*>(dst);
        t0.x += a[0]; t0.y += a[1];
        *reinterpret_cast<float2*>(dst) = t0;
    }
}

// Reduce pass, one workgroup per (bucket, level); integer LDS atomics only (see the header of this section).
// Per chunk of hg_chunk<F>() records (streamed, coalesced; no index math):
//   1. every thread loads its records {sample|row, w}, gathers the staged gradient of its samples (independent
//      loads) and ranks each record within its row (pos = ds_add_rtn_u32 on the row counter);
//   2. exclusive scan of the row counters -> segment offsets; rows with long segments are queued;
//   3. the record payloads w*g are written to their row-sorted slots of an LDS staging array;
//   4. owner threads sum their rows' (short) segments from LDS into registers, waves sum the long ones with a shuffle
//      reduction; each row is added to the gradient table by exactly one lane with a plain read-modify-write.
// ADAM = true: for the levels >= adam.from_level the reduce pass IS the optimizer step.  A workgroup owns its bucket's rows
// and, after the last chunk, holds their complete gradient in registers, so it reads p / exp_avg / exp_avg_sq of those
// rows, applies torch.optim.Adam's update (optim.hip: adam1) and writes them back: 24 B per parameter instead of the
// 8 B gradient read-modify-write here plus the 32 B of the separate Adam pass (which also re-zeroes the gradient).
// Valid when this launch sees the whole gradient of the table: one backward per step, gradients not exchanged between
// ranks (one rank, or table-parallel levels).  Buckets nothing landed in are still visited (m and v decay, p moves).
struct HgAdam {
    float *p, *m, *v;
    float b1, b2, step_size, inv_sqrt_bc2, eps, gs;
    int from_level;
    // the non-finite-gradient guard bound when the launch was issued (snf_step_guard, common.hpp), or nullptr; resolved once at the top
    // of a kernel by hg_guard: `veto` set = this step leaves p / m / v as they are
    const int32_t* guard;
    float lr;
    int step, veto;
};

__device__ __forceinline__ void hg_guard(HgAdam& a) {
    const GuardAdam ga = guard_adam(a.guard, a.lr, a.b1, a.b2, a.step, a.step_size, a.inv_sqrt_bc2);
    a.step_size = ga.step_size;
    a.inv_sqrt_bc2 = ga.inv_sqrt_bc2;
    a.veto = ga.veto ? 1 : 0;
}

__device__ __forceinline__ void hg_adam1(float& p, float g, float& m, float& v, const HgAdam& a) {  // == optim.hip adam1
    if (a.veto) return;  // (kernel-uniform)
    const float gg = g * a.gs;
    m = m + (gg - m) * (1.f - a.b1);
    v = v * a.b2 + (1.f - a.b2) * gg * gg;
    const float denom = sqrtf(v) * a.inv_sqrt_bc2 + a.eps;
    p = p - a.step_size * (m / denom);
}

// ==========================================================================================
// Reachable-row levels (round 3): fixed-point reduce over COMPACT row indices, Adam fused.
//
// A level of resolution s can only ever address the <= (s + 3)^3 rows its lattice hashes to (Encoding.active_rows).
// On the coarse levels of a T = 19 table that is 19 ... ~900 of a bucket's 2048 rows, each receiving 100 ... 2 records of a
// 65 536-sample step -- the regime where the float reduce above ranks hundreds of records on ONE LDS counter (same-address
// atomics serialise: the eight such levels of a 16 -> 128 feature grid took as long as its four dense levels, which move 4x
// the bytes) and where the plain fixed-point reduce piles its 64-bit atomics on a handful of addresses.
// Here the bucket's reachable rows (a static, sorted list per (level, bucket)) get compact indices 0 .. nrows-1 through a
// 2048-entry LDS lookup, and the accumulator array holds as many COPIES of the nrows x F sums as fit (lane j adds into copy
// j mod copies; 64 copies when nrows <= 16: no two lanes of a wave ever share an address).  Integer sums commute, so the copies
// are added in any order and the result stays exact and order-independent.  The epilogue applies Adam to exactly the reachable
// rows -- zero gradient or not, like snf_adam_step_rows -- so these levels write no gradient and need no separate optimizer
// pass; without ADAM the sums are added to the gradient table.
// ==========================================================================================
constexpr int HG_FX_BITS = 38;  // fixed-point resolution: 2^-38 of a level's largest |g| (see the fixed-point reduce below)
constexpr int HG_SP_T = 512;
template <int F> constexpr int hg_sp_acc_words() { return F == 8 ? 8192 : 4096; }  // 64-bit accumulators: 64 KB (F = 8), 32 KB (F = 2)
template <int F> constexpr int hg_sp_max_rows() { return hg_sp_acc_words<F>() / F; }   // reachable rows per bucket this kernel takes

template <int F> inline size_t hg_sp_lds_bytes(int log2rpb) {
    return (size_t)hg_sp_acc_words<F>() * 8 + (size_t)hg_sp_acc_words<F>() / 32 * 4 + ((size_t)2 << log2rpb);
}

// The body is shared by the stand-alone kernel below and by k_hg_reduce, whose launches take the reachable-row levels of their
// table(s) as ordinary (bucket, level) workgroups: the latency-bound compact reduce then runs BESIDE the bandwidth-bound
// bucket-wide levels instead of in front of them (two launches in a row cost the paired backward 0.12 ms per step).
// acc: ACC 64-bit words [(row * F + f) * copies + copy]; bad: ACC / 32 words; lookup: rpb half-words.
template <int F>
__device__ __forceinline__ void hg_sparse_body(const float* __restrict__ gT, int N, int log2_T, int log2B, int b, int l,
                                               const uint32_t* __restrict__ bucket_start, const uint2* __restrict__ records,
                                               float* __restrict__ grad_table, const uint32_t* __restrict__ reach_rows,
                                               const uint32_t* __restrict__ reach_start,
                                               const uint32_t* __restrict__ lvl_absmax_bits, const HgAdam& adam, bool adam_on,
                                               unsigned long long* acc, uint32_t* bad, uint16_t* lookup) {
    constexpr int ACC = hg_sp_acc_words<F>();
    const int B = 1 << log2B, log2rpb = log2_T - log2B, rpb = 1 << log2rpb;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t rs = reach_start[l * B + b];
    const int nrows = (int)(reach_start[l * B + b + 1] - rs);
    if (nrows <= 0) return;  // no row of this bucket can be addressed: no record lands here, nothing to step
    const uint32_t start = bucket_start[l * (B + 1) + b], end = bucket_start[l * (B + 1) + b + 1];
    if (!adam_on && start == end) return;
    int copies = ACC / (nrows * F);
    copies = copies >= 64 ? 64 : (copies < 1 ? 1 : 1 << (31 - __clz(copies)));  // power of two, at most one per lane
    // layout acc[(row * F + f) * copies + copy]: the copies of one sum are ADJACENT 8-byte words, so the lanes of a wave that hit one
    // row (the common case on a coarse level) spread over all LDS banks; with the copies strided by nrows * F words they fell
    // into the same few banks (rows are 64 B: four bank groups) and the "conflict-free" atomics serialised 16-way
    const uint32_t copy = (uint32_t)(lane & (copies - 1));
    const int csh = 31 - __clz(copies);
    for (int i = tid; i < rpb; i += HG_SP_T) lookup[i] = 0xFFFFu;
    for (int i = tid; i < copies * nrows * F; i += HG_SP_T) acc[i] = 0ull;
    for (int i = tid; i < (nrows * F + 31) / 32; i += HG_SP_T) bad[i] = 0u;
    __syncthreads();
    for (int i = tid; i < nrows; i += HG_SP_T) lookup[reach_rows[rs + i] & (uint32_t)(rpb - 1)] = (uint16_t)i;
    const float* __restrict__ gl = gT + (size_t)l * N * F;
    constexpr int U = 4;
    const uint32_t mask_s = (1u << HG_SAMPLE_BITS) - 1u;
    uint2 rec0[U], rec1[U];
    float g0[U][F], g1[U][F];
    auto load_recs = [&](uint32_t c0, uint2 (&r)[U]) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const uint32_t i = c0 + tid + (uint32_t)HG_SP_T * j;
            r[j] = records[i < end ? i : (start < end ? start : 0u)];
        }
    };
    auto gather = [&](const uint2 (&r)[U], float (&g)[U][F]) {
#pragma unroll
        for (int j = 0; j < U; ++j) load_row<F>(gl + (size_t)(r[j].x & mask_s) * F, g[j]);
    };
    constexpr uint32_t TRIP = HG_SP_T * U;
    // ---- the fixed-point scale: 2^e above the largest finite |w g| of THIS bucket's records.  The rows of a bucket belong to one
    // workgroup, so its own maximum is all the scale has to cover (finer than the level-wide maximum of the bucket-wide fixed-point
    // kernel, and no pre-pass over the staged gradient: two launches less in front of every table backward).  At the step's sizes a
    // bucket's records are one trip, held in registers for the accumulation below; longer buckets stream their records twice.
    auto trip_max = [&](uint32_t c0, const uint2 (&r)[U], const float (&g)[U][F]) {
        float m = 0.f;
#pragma unroll
        for (int j = 0; j < U; ++j) {
            if (c0 + tid + (uint32_t)HG_SP_T * j < end) {
                const float w = fabsf(__uint_as_float(r[j].y));
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    const float v = w * fabsf(g[j][f]);
                    if (v < INFINITY && v > m) m = v;  // (NaN fails both comparisons)
                }
            }
        }
        return m;
    };
    load_recs(start, rec0);
    gather(rec0, g0);
    float lmax = trip_max(start, rec0, g0);
    const bool single = start + TRIP >= end;
    if (!single) {
        for (uint32_t c0 = start + TRIP; c0 < end; c0 += TRIP) {
            load_recs(c0, rec1);
            gather(rec1, g1);
            lmax = fmaxf(lmax, trip_max(c0, rec1, g1));
        }
    }
#pragma unroll
    for (int dlt = 32; dlt > 0; dlt >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, dlt, 64));
    __shared__ float sp_wmax[HG_SP_T / 64];
    if (lane == 0) sp_wmax[tid >> 6] = lmax;
    __syncthreads();  // (also: the lookup table is complete)
#pragma unroll
    for (int q = 0; q < HG_SP_T / 64; ++q) lmax = fmaxf(lmax, sp_wmax[q]);
    int e = 0;
    frexpf(lmax, &e);
    int sh = HG_FX_BITS - e;
    sh = sh > 120 ? 120 : sh;
    const float scale = ldexpf(1.f, sh), inv = ldexpf(1.f, -sh);
    (void)lvl_absmax_bits;
    if (start + TRIP < end) load_recs(start + TRIP, rec1);
    for (uint32_t c0 = start; c0 < end; c0 += TRIP) {
        uint2 rec2[U];
        const bool has1 = c0 + TRIP < end, has2 = c0 + 2 * TRIP < end;
        if (has1) gather(rec1, g1);
        if (has2) load_recs(c0 + 2 * TRIP, rec2);
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const bool live = c0 + tid + (uint32_t)HG_SP_T * j < end;
            const uint32_t idx = lookup[rec0[j].x >> HG_SAMPLE_BITS];
            if (live && idx != 0xFFFFu) {
                const float w = __uint_as_float(rec0[j].y);
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    const float v = w * g0[j][f];
                    if (fabsf(v) < INFINITY) {
                        const long long q = __float2ll_rn(v * scale);
                        if (q != 0) atomicAdd(&acc[((idx * F + f) << csh) + copy], (unsigned long long)q);
                    } else {
                        const uint32_t eb = idx * F + f;
                        atomicOr(&bad[eb >> 5], 1u << (eb & 31));
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            rec0[j] = rec1[j];
            rec1[j] = rec2[j];
#pragma unroll
            for (int f = 0; f < F; ++f) g0[j][f] = g1[j][f];
        }
    }
    __syncthreads();
    // ---- epilogue: one thread per reachable row
    for (int i = tid; i < nrows; i += HG_SP_T) {
        const size_t o = (size_t)reach_rows[rs + i] * F;  // (the list holds (level << T) + row: an offset into the whole table)
        float gg[F];
        bool nz = false;
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const uint32_t eb = (uint32_t)(i * F + f);
            unsigned long long qs = 0ull;
            // (lane i starts at copy i: the lanes' reads then fall into different banks -- their elements are copies * 8 B apart)
            for (int c = 0; c < copies; ++c) qs += acc[(eb << csh) + (((uint32_t)c + (uint32_t)lane) & (uint32_t)(copies - 1))];
            const bool isbad = (bad[eb >> 5] >> (eb & 31)) & 1u;
            gg[f] = isbad ? __uint_as_float(0x7FC00000u) : __ll2float_rn((long long)qs) * inv;
            nz = nz || qs != 0ull || isbad;
        }
        if (adam_on) {
            float pp[F], mm[F], vv[F];
            load_row<F>(adam.p + o, pp);
            load_row<F>(adam.m + o, mm);
            load_row<F>(adam.v + o, vv);
#pragma unroll
            for (int f = 0; f < F; ++f) hg_adam1(pp[f], gg[f], mm[f], vv[f], adam);
            if constexpr (F == 2) {
                *reinterpret_cast<float2*>(adam.p + o) = make_float2(pp[0], pp[1]);
                *reinterpret_cast<float2*>(adam.m + o) = make_float2(mm[0], mm[1]);
                *reinterpret_cast<float2*>(adam.v + o) = make_float2(vv[0], vv[1]);
            } else {
                reinterpret_cast<float4*>(adam.p + o)[0] = make_float4(pp[0], pp[1], pp[2], pp[3]);
                reinterpret_cast<float4*>(adam.p + o)[1] = make_float4(pp[4], pp[5], pp[6], pp[7]);
                reinterpret_cast<float4*>(adam.m + o)[0] = make_float4(mm[0], mm[1], mm[2], mm[3]);
                reinterpret_cast<float4*>(adam.m + o)[1] = make_float4(mm[4], mm[5], mm[6], mm[7]);
                reinterpret_cast<float4*>(adam.v + o)[0] = make_float4(vv[0], vv[1], vv[2], vv[3]);
                reinterpret_cast<float4*>(adam.v + o)[1] = make_float4(vv[4], vv[5], vv[6], vv[7]);
            }
        } else if (nz) {
            row_rmw<F>(grad_table + o, gg);
        }
    }
}


template <int F, bool ADAM>
__global__ __launch_bounds__(HG_SP_T) void k_hg_reduce_sparse(const float* __restrict__ gT, int N, int log2_T, int log2B,
                                                              const uint32_t* __restrict__ bucket_start,
                                                              const uint2* __restrict__ records, float* __restrict__ grad_table,
                                                              const uint32_t* __restrict__ reach_rows,
                                                              const uint32_t* __restrict__ reach_start,
                                                              const uint32_t* __restrict__ lvl_absmax_bits, HgAdam adam) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long sp_lds[];
    constexpr int ACC = hg_sp_acc_words<F>();
    unsigned long long* acc = sp_lds;
    uint32_t* bad = reinterpret_cast<uint32_t*>(acc + ACC);
    uint16_t* lookup = reinterpret_cast<uint16_t*>(bad + ACC / 32);
    if constexpr (ADAM) hg_guard(adam);
    hg_sparse_body<F>(gT, N, log2_T, log2B, (int)blockIdx.x, (int)blockIdx.y, bucket_start, records, grad_table, reach_rows,
                      reach_start, lvl_absmax_bits, adam, ADAM, acc, bad, lookup);
}

// A second grid in the same launch (snf_hashgrid_bwd_presorted_adam_pair: the two F = 8 grids of a feature head, same N / T):
// blockIdx.y covers the levels of both; with `interleave` (equal level counts) even y = first grid, odd y = second, so the
// latency-bound reachable-row levels of one grid are resident together with the bandwidth-bound dense levels of the other.
// the reachable-row levels of a table handled inside a k_hg_reduce launch (levels [0, levels) of that table)
struct HgSparseDev {
    const uint32_t* rows;
    const uint32_t* start;
    const uint32_t* lvlmax;  // per-level largest finite |g| (bits), from k_hg_level_absmax
    int levels;
    int step;                // apply Adam to the listed rows (else: add the sums to the gradient table)
};

struct HgSecond {
    const float* gT;
    const uint32_t* bucket_start;
    const uint2* records;
    float* grad_table;
    HgAdam adam;
    int first_levels;  // levels of the first grid IN THIS LAUNCH; >= gridDim.y: there is no second grid
    int interleave;
    int level0;        // first level of the second grid covered by this launch
    HgSparseDev sp;    // its reachable-row levels
};

#define SNF_HG_RT_MINWG 4
// (HIP: the second launch-bounds value is the minimum WAVES per SIMD.  Two 8-wave workgroups per CU -- LDS allows it at F = 8 --
// are 4 waves per SIMD, i.e. <= 128 VGPRs; without the hint the compiler took 130: ONE workgroup per CU)
template <int F, bool ADAM>
__global__ __launch_bounds__(HG_RT, SNF_HG_RT_MINWG) void k_hg_reduce(const float* __restrict__ gT, int N, int log2_T, int log2B,
                                                     const uint32_t* __restrict__ bucket_start,
                                                     const uint2* __restrict__ records, float* __restrict__ grad_table,
                                                     uint32_t hg_long, int n_run_levels, int level0, HgAdam adam, HgSparseDev sp,
                                                     HgSecond sec) {
    constexpr int CHUNK = hg_chunk<F>();
    constexpr int RPT = CHUNK / HG_RT;
    __shared__ uint32_t cnt[HG_MAX_RPB + 1];  // per-row counts, then exclusive offsets
    __shared__ __attribute__((aligned(16))) float val[CHUNK * F];  // w*g per record, row-sorted
    __shared__ uint32_t wave_tot[HG_RT / 64];
    __shared__ uint32_t long_rows[CHUNK / 8 + 1];
    __shared__ uint32_t n_long;
    const int B = 1 << log2B, log2rpb = log2_T - log2B;
    const int rpb = 1 << log2rpb;
    const int tid = threadIdx.x;
    int b = blockIdx.x, l = blockIdx.y;
    // (round 5: an XCD-aware order of the dense levels -- in rounds of eight levels XCD x owns ONE level, so that a level's staged-gradient
    //  slab is pulled by one L2 instead of all eight -- measured SLOWER: 0.886 -> 0.931 ms per step serial, step 2.51 -> 2.545 ms.  Like the
    //  forward's remaps ("XCD mapping"): every L2 on the same level at the same time is the better order; the slab copies come
    //  out of the Infinity Cache, not out of HBM.  profiles/EXPERIMENTS.md r05)
    if (sec.first_levels < (int)gridDim.y) {  // (uniform over the workgroup)
        bool second;
        if (sec.interleave) {
            second = l & 1;
            l >>= 1;
        } else {
            second = l >= sec.first_levels;
            if (second) l -= sec.first_levels;
        }
        if (second) {
            gT = sec.gT; bucket_start = sec.bucket_start; records = sec.records; grad_table = sec.grad_table; adam = sec.adam;
            level0 = sec.level0;
            sp = sec.sp;
        }
    }
    if constexpr (ADAM) hg_guard(adam);
    l += level0;  // (a launch may cover the level sub-range [level0, level0 + n) of its table; all indices below are absolute)
    if constexpr (F == 8) {
        if (l < sp.levels) {
            // a reachable-row level of this table: the compact fixed-point reduce, in the LDS of the sort machinery (val = the 8192
            // accumulators, cnt = lookup + flags) -- a latency-bound workgroup beside the bandwidth-bound ones of the other levels
            static_assert(sizeof(val) >= (size_t)hg_sp_acc_words<8>() * 8 && sizeof(cnt) >= HG_MAX_RPB * 2 + hg_sp_acc_words<8>() / 8,
                          "the compact reduce does not fit the float reduce's LDS");
            hg_sparse_body<8>(gT, N, log2_T, log2B, b, l, bucket_start, records, grad_table, sp.rows, sp.start, sp.lvlmax, adam,
                              sp.step != 0, reinterpret_cast<unsigned long long*>(val), cnt + HG_MAX_RPB / 2,
                              reinterpret_cast<uint16_t*>(cnt));
            return;
        }
    }
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t start = bucket_start[l * (B + 1) + b], end = bucket_start[l * (B + 1) + b + 1];
    const bool fuse = ADAM && l >= adam.from_level;
    if (start == end && !fuse) return;  // nothing lands in this bucket: leave the slab untouched
    bool any_long = false;
    float* __restrict__ slab = grad_table + (((size_t)l << log2_T) + ((size_t)b << log2rpb)) * F;
    const float* __restrict__ gl = gT + (size_t)l * N * F;
    if (hg_long < 8u) hg_long = 8u;  // capacity of long_rows
    const bool coarse = l < n_run_levels;
    float racc[HG_ROWS_PT][F];       // running sums of the owned rows over all chunks (short segments)
#pragma unroll
    for (int q = 0; q < HG_ROWS_PT; ++q)
#pragma unroll
        for (int f = 0; f < F; ++f) racc[q][f] = 0.f;

    // Software pipeline over chunks: the records of chunk c+1 are requested at the top of chunk c and their staged
    // gradients gathered half-way through it, so both global latencies of a chunk hide under the LDS work of the previous
    // one (workgroup barriers do not wait for outstanding global loads on gfx950).
    // (F = 2 only: at F = 8 the second register set of 4 x 8 gradients costs more occupancy than the overlap returns.)
    constexpr bool PIPE = (F == 2);
    constexpr int NRPT = PIPE ? RPT : 1;
    uint2 rec[RPT], recn[NRPT];
    float g[RPT][F], gn[NRPT][F];
    auto load_recs = [&](uint32_t c0, auto& r) {
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const uint32_t i = c0 + tid + (uint32_t)HG_RT * j;
            r[j] = records[i < end ? i : (start < end ? start : 0u)];  // (an empty bucket is still visited when ADAM)
        }
    };
    auto gather = [&](const auto& r, auto& gg) {
#pragma unroll
        for (int j = 0; j < RPT; ++j) load_row<F>(gl + (size_t)(r[j].x & ((1u << HG_SAMPLE_BITS) - 1u)) * F, gg[j]);
    };
    load_recs(start, rec);
    for (int i = tid; i <= rpb; i += HG_RT) cnt[i] = 0u;
    if (tid == 0) n_long = 0u;
    gather(rec, g);
    __syncthreads();

    for (uint32_t c0 = start; c0 < end; c0 += CHUNK) {
        const bool has_next = c0 + CHUNK < end;
        if constexpr (PIPE) {
            if (has_next) load_recs(c0 + CHUNK, recn);
        }
        // ---- phase 1: rank within row.  At a coarse level neighbouring records of the bucket stream mostly come from
        // neighbouring samples of one ray inside one cell, i.e. carry the same row: such runs are summed across adjacent
        // lanes first (segmented shuffle scan) and only the run tail is ranked, sorted and summed -- it removes the
        // same-address rank atomics and the thousand-record segments that made a coarse level 5-8x dearer than a fine one.
        uint32_t row[RPT], pos[RPT];
        float v[RPT][F];
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            bool live = c0 + tid + (uint32_t)HG_RT * j < end;
            row[j] = rec[j].x >> HG_SAMPLE_BITS;
            const float w = __uint_as_float(rec[j].y);
#pragma unroll
            for (int f = 0; f < F; ++f) v[j][f] = w * g[j][f];
            if (coarse) {
                const uint32_t key = live ? row[j] : (0xFFFFFF00u + (uint32_t)lane);  // dead lanes: singleton runs
                const uint32_t prev = __shfl_up(key, 1, 64);
                const unsigned long long heads = __ballot((lane == 0) || (prev != key));
                const int h = 63 - __builtin_clzll(heads & ((2ull << lane) - 1ull));  // first lane of this lane's run
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const bool take = (lane - d) >= h;
#pragma unroll
                    for (int f = 0; f < F; ++f) {
                        const float t = __shfl_up(v[j][f], d, 64);
                        if (take) v[j][f] += t;
                    }
                }
                const uint32_t next = __shfl_down(key, 1, 64);
                live = live && ((lane == 63) || (next != key));  // the run tail carries the run's sum
            }
            pos[j] = live ? atomicAdd(&cnt[row[j]], 1u) : 0xFFFFFFFFu;
        }
        __syncthreads();
        // ---- phase 2: exclusive scan of cnt[0..rpb) in place (thread t scans HG_ROWS_PT consecutive entries)
        {
            const int i0 = tid * HG_ROWS_PT;
            uint32_t c[HG_ROWS_PT];
            uint32_t sum = 0;
#pragma unroll
            for (int q = 0; q < HG_ROWS_PT; ++q) {
                c[q] = (i0 + q < rpb) ? cnt[i0 + q] : 0u;
                if (c[q] > hg_long) long_rows[atomicAdd(&n_long, 1u)] = (uint32_t)(i0 + q);
                sum += c[q];
            }
            uint32_t inc = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t = __shfl_up(inc, d, 64);
                if (lane >= d) inc += t;
            }
            if (lane == 63) wave_tot[wave] = inc;
            __syncthreads();
            uint32_t base = 0;
            for (int w2 = 0; w2 < wave; ++w2) base += wave_tot[w2];
            uint32_t run = base + inc - sum;
#pragma unroll
            for (int q = 0; q < HG_ROWS_PT; ++q) {
                if (i0 + q < rpb) cnt[i0 + q] = run;
                run += c[q];
            }
            if (tid == HG_RT - 1) cnt[rpb] = base + inc;
        }
        __syncthreads();
        if constexpr (PIPE) {
            if (has_next) gather(recn, gn);  // the next chunk's records have landed by now
        }
        // ---- phase 3: payloads w*g to their row-sorted slots
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            if (pos[j] != 0xFFFFFFFFu) {
                float* d = &val[(size_t)(cnt[row[j]] + pos[j]) * F];
                if constexpr (F == 8) {
                    reinterpret_cast<float4*>(d)[0] = make_float4(v[j][0], v[j][1], v[j][2], v[j][3]);
                    reinterpret_cast<float4*>(d)[1] = make_float4(v[j][4], v[j][5], v[j][6], v[j][7]);
                } else {
                    *reinterpret_cast<float2*>(d) = make_float2(v[j][0], v[j][1]);
                }
            }
        }
        __syncthreads();
        // ---- phase 4a: short segments, one owner thread per row (rows tid, tid + HG_RT, ...), summed from LDS
#pragma unroll
        for (int sidx = 0; sidx < HG_ROWS_PT; ++sidx) {
            const int r = tid + sidx * HG_RT;
            if (r < rpb) {
                const uint32_t e0 = cnt[r], e1 = cnt[r + 1];
                if (e1 - e0 <= hg_long) {
                    for (uint32_t e = e0; e < e1; ++e) {
#pragma unroll
                        for (int f = 0; f < F; ++f) racc[sidx][f] += val[(size_t)e * F + f];
                    }
                }
            }
        }
        // ---- phase 4b: long segments, one wave per row
        const uint32_t nl = n_long;
        any_long |= nl > 0u;
        for (uint32_t q = wave; q < nl; q += HG_RT / 64) {
            const uint32_t r = long_rows[q];
            const uint32_t e0 = cnt[r], e1 = cnt[r + 1];
            float a[F];
#pragma unroll
            for (int f = 0; f < F; ++f) a[f] = 0.f;
            for (uint32_t e = e0 + lane; e < e1; e += 64) {
#pragma unroll
                for (int f = 0; f < F; ++f) a[f] += val[(size_t)e * F + f];
            }
#pragma unroll
            for (int f = 0; f < F; ++f) {
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) a[f] += __shfl_xor(a[f], d, 64);
            }
            if (lane == 0) row_rmw<F>(slab + (size_t)r * F, a);
        }
        __syncthreads();  // also orders this chunk's table updates before the next chunk's
        if (has_next) {
            for (int i = tid; i <= rpb; i += HG_RT) cnt[i] = 0u;
            if (tid == 0) n_long = 0u;
            if constexpr (PIPE) {
#pragma unroll
                for (int j = 0; j < RPT; ++j) {
                    rec[j] = recn[j];
#pragma unroll
                    for (int f = 0; f < F; ++f) g[j][f] = gn[j][f];
                }
            } else {
                load_recs(c0 + CHUNK, rec);
                gather(rec, g);
            }
            __syncthreads();
        }
    }
    if constexpr (ADAM) {
        if (fuse) {
            // ---- fused optimizer epilogue: every owned row, HG_EPI rows (3 row loads each) in flight per thread
            const size_t base = (((size_t)l << log2_T) + ((size_t)b << log2rpb)) * F;
#pragma unroll
            for (int q0 = 0; q0 < HG_ROWS_PT; q0 += HG_EPI) {
                float pp[HG_EPI][F], mm[HG_EPI][F], vv[HG_EPI][F], gg[HG_EPI][F];
                bool on[HG_EPI];
#pragma unroll
                for (int j = 0; j < HG_EPI; ++j) {
                    const int r = tid + (q0 + j) * HG_RT;
                    on[j] = r < rpb;
                    if (on[j]) {
                        const size_t o = base + (size_t)r * F;
                        load_row<F>(adam.p + o, pp[j]);
                        load_row_mv<F>(adam.m + o, mm[j]);
                        load_row_mv<F>(adam.v + o, vv[j]);
                        if (any_long) load_row<F>(slab + (size_t)r * F, gg[j]);  // wave sums of long segments went there
                    }
                }
#pragma unroll
                for (int j = 0; j < HG_EPI; ++j) {
                    if (on[j]) {
                        const int r = tid + (q0 + j) * HG_RT;
                        const size_t o = base + (size_t)r * F;
                        bool had = false;
#pragma unroll
                        for (int f = 0; f < F; ++f) {
                            float g = racc[q0 + j][f];
                            if (any_long) {
                                had |= gg[j][f] != 0.f;
                                g = gg[j][f] + g;
                            }
                            hg_adam1(pp[j][f], g, mm[j][f], vv[j][f], adam);
                        }
                        if constexpr (F == 8) {
                            reinterpret_cast<float4*>(adam.p + o)[0] = make_float4(pp[j][0], pp[j][1], pp[j][2], pp[j][3]);
                            reinterpret_cast<float4*>(adam.p + o)[1] = make_float4(pp[j][4], pp[j][5], pp[j][6], pp[j][7]);
                            store_row8_mv(adam.m + o, mm[j]);
                            store_row8_mv(adam.v + o, vv[j]);
                            if (had) {
                                reinterpret_cast<float4*>(slab + (size_t)r * F)[0] = make_float4(0.f, 0.f, 0.f, 0.f);
                                reinterpret_cast<float4*>(slab + (size_t)r * F)[1] = make_float4(0.f, 0.f, 0.f, 0.f);
                            }
                        } else {
                            *reinterpret_cast<float2*>(adam.p + o) = make_float2(pp[j][0], pp[j][1]);
                            *reinterpret_cast<float2*>(adam.m + o) = make_float2(mm[j][0], mm[j][1]);
                            *reinterpret_cast<float2*>(adam.v + o) = make_float2(vv[j][0], vv[j][1]);
                            if (had) *reinterpret_cast<float2*>(slab + (size_t)r * F) = make_float2(0.f, 0.f);
                        }
                    }
                }
            }
            return;
        }
    }
    // ---- epilogue: one read-modify-write per owned row that received something; the loads issue together
    float cur[HG_ROWS_PT][F];
    bool nz[HG_ROWS_PT];
#pragma unroll
    for (int q = 0; q < HG_ROWS_PT; ++q) {
        const int r = tid + q * HG_RT;
        nz[q] = false;
#pragma unroll
        for (int f = 0; f < F; ++f) nz[q] |= racc[q][f] != 0.f;
        nz[q] = nz[q] && (r < rpb);
        if (nz[q]) load_row<F>(slab + (size_t)r * F, cur[q]);
    }
#pragma unroll
    for (int q = 0; q < HG_ROWS_PT; ++q) {
        if (nz[q]) {
            float* dst = slab + (size_t)(tid + q * HG_RT) * F;
            if constexpr (F == 8) {
                reinterpret_cast<float4*>(dst)[0] = make_float4(cur[q][0] + racc[q][0], cur[q][1] + racc[q][1],
                                                                cur[q][2] + racc[q][2], cur[q][3] + racc[q][3]);
                reinterpret_cast<float4*>(dst)[1] = make_float4(cur[q][4] + racc[q][4], cur[q][5] + racc[q][5],
                                                                cur[q][6] + racc[q][6], cur[q][7] + racc[q][7]);
            } else {
                *reinterpret_cast<float2*>(dst) = make_float2(cur[q][0] + racc[q][0], cur[q][1] + racc[q][1]);
            }
        }
    }
}


// ==========================================================================================
// Fixed-point reduce pass (F = 2 grids): per-row sums as 64-bit INTEGER LDS atomics, no sort inside the bucket.
//
// The reduce pass above spends its time ranking records inside LDS, not moving bytes (rocprofv3, field grid: 66 % of
// the wave cycles parked at barriers / waitcnt, ~90 VALU instructions per record, 1.85 TB/s of algorithmic traffic), all to
// avoid float atomics: ds_add_f32 retires 0.37 lane-ops/clk/CU.  ds_add_u64 retires 6.3 (tools/ubench/lds_atomic_rate2.hip),
// so a contribution w*g is added as a 64-bit fixed-point integer instead:
//     q = rint(w*g * 2^s),  2^s = 2^38 / 2^e,  2^e > max |g| over the level (k_hg_level_absmax)
// |q| < 2^38 leaves 24 bits of headroom for the 2^24 contributions a row can receive at most (2^21 samples x 8 corners), and
// resolves every contribution to 2^-38 of the level's largest gradient -- finer than the fp32 round-off of the products
// themselves for anything above 2^-14 of it, and EXACT in the sum: the result does not depend on the order of the
// records, so the table gradient is bit-reproducible (the float reduce differed from run to run only through its chunking).
// Non-finite contributions cannot be represented: they set a per-row flag and the row's gradient becomes NaN, which is
// what a float sum (autograd's) gives.  A workgroup streams its bucket's records once (record + staged-gradient gather,
// four in flight per thread), one barrier, then converts its rows and applies the Adam step / gradient read-modify-write.
// ==========================================================================================
#define SNF_FX_T 512
constexpr int HG_FX_T = SNF_FX_T;

template <int F>
__global__ __launch_bounds__(256) void k_hg_level_absmax(const float* __restrict__ gT, int N, uint32_t* __restrict__ out_bits) {
    const int l = blockIdx.y;
    const float4* __restrict__ p = reinterpret_cast<const float4*>(gT + (size_t)l * N * F);
    const long long n4 = (long long)N * F / 4;  // N * F is a multiple of 4 (F in {2, 8}; N even checked by the caller for F = 2)
    float m = 0.f;
    const long long stride = (long long)gridDim.x * 256;
    for (long long i0 = (long long)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += 4 * stride) {
        float4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // four independent loads in flight
            const long long i = i0 + j * stride;
            v[j] = i < n4 ? p[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = fabsf(v[j].x), b = fabsf(v[j].y), c = fabsf(v[j].z), d = fabsf(v[j].w);
            // finite values only (NaN fails the comparison, +inf is excluded explicitly): non-finite contributions are flagged per row
            if (a < INFINITY && a > m) m = a;
            if (b < INFINITY && b > m) m = b;
            if (c < INFINITY && c > m) m = c;
            if (d < INFINITY && d > m) m = d;
        }
    }
#pragma unroll
    for (int dlt = 32; dlt > 0; dlt >>= 1) m = fmaxf(m, __shfl_xor(m, dlt, 64));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    // one atomic per workgroup (same-address device atomics serialise at the memory side: one per WAVE cost 100 us here)
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        if (m > 0.f) atomicMax(&out_bits[l], __float_as_uint(m));  // non-negative floats order as uints
    }
}

// SPLIT: workgroups per bucket.  The accumulators of a whole 2048-row bucket are 32 KB at F = 2 but 128 KB at F = 8, so at
// F = 8 two workgroups share a bucket: each streams ALL of its records (8 bytes each, cheap) but gathers and accumulates
// only those whose row falls into its half, and owns the Adam step of that half's rows.
// XP (F = 2, SPLIT = 1): x-pair records (k_hg_scatter_xp): a lane serves the two rows of a pair from one gather.
template <int F, bool ADAM, int SPLIT, bool XP = false>
__global__ __launch_bounds__(HG_FX_T) void k_hg_reduce_fx(const float* __restrict__ gT, int N, int log2_T, int log2B,
                                                          const uint32_t* __restrict__ bucket_start,
                                                          const typename std::conditional<XP, uint4, uint2>::type* __restrict__ records,
                                                          float* __restrict__ grad_table,
                                                          int xcd_from_level, int n_merge_levels, int level0,
                                                          const uint32_t* __restrict__ lvl_absmax_bits, HgAdam adam) {
    static_assert(!XP || (F == 2 && SPLIT == 1), "x-pair records: F = 2 grids, one workgroup per bucket");
    using Rec = typename std::conditional<XP, uint4, uint2>::type;
    if constexpr (ADAM) hg_guard(adam);
    constexpr int MAXROWS = HG_MAX_RPB / SPLIT;
    __shared__ unsigned long long acc[MAXROWS * F];
    __shared__ uint32_t bad[MAXROWS * F / 32];  // one bit per (row, feature): a non-finite contribution landed there
    const int B = 1 << log2B, log2rpb = log2_T - log2B;
    const int rpb_full = 1 << log2rpb;
    const int rpb = rpb_full >= SPLIT ? rpb_full / SPLIT : rpb_full;  // rows of this workgroup
    const int part = rpb_full >= SPLIT ? (int)(blockIdx.x % SPLIT) : 0;
    const int tid = threadIdx.x, lane = tid & 63;
    int b = blockIdx.x / SPLIT, l = blockIdx.y;
    if (SPLIT == 1 && xcd_from_level < (int)gridDim.y) {
        // XCD-aware order for the levels >= xcd_from_level (uniform cost): workgroups go to the 8 XCDs round-robin by linear id,
        // and every (bucket, level) workgroup gathers from the WHOLE staged-gradient slab of its level (4 MB at N = 2^19, the
        // size of one L2).  In launch order ~4 levels are resident at once and every L2 sees all of them; here XCD c takes a
        // CONTIGUOUS run of the level-major (level, bucket) list, so an L2 serves one level at a time.  The coarse levels before
        // xcd_from_level keep the plain order: their workgroups are the long ones and want all CUs.
        const int lin = (int)blockIdx.x + (int)blockIdx.y * B;
        const int plain = xcd_from_level * B;
        if (lin >= plain) {
            const int rem = lin - plain, per_xcd = (((int)gridDim.y - xcd_from_level) * B) >> 3;  // B % 8 == 0
            const int f = (rem & 7) * per_xcd + (rem >> 3);
            l = xcd_from_level + f / B;
            b = f % B;
        }
    }
    if (rpb_full < SPLIT && (blockIdx.x % SPLIT) != 0) return;  // (tiny tables: one workgroup per bucket)
    // (a launch may cover the level sub-range [level0, level0 + gridDim.y) of its table: xcd_from_level / n_merge_levels above are
    //  relative to the launch, every index below is absolute)
    const bool merge_level = l < n_merge_levels;
    l += level0;
    const uint32_t row0 = (uint32_t)part * (uint32_t)rpb;
    const uint32_t start = bucket_start[l * (B + 1) + b], end = bucket_start[l * (B + 1) + b + 1];
    const bool fuse = ADAM && l >= adam.from_level;
    if (start == end && !fuse) return;
    // Small tables (rpb < MAXROWS, e.g. the T = 17 proposal grid) leave accumulator space unused: it holds up to 8 copies of the
    // bucket's rows, lane j adding into copy j % copies.  Integer sums commute, so the epilogue adds the copies in any order; at a
    // coarse level, where runs of adjacent lanes hit ONE row (same-address LDS atomics serialise), this cuts the conflict depth.
    int copies = MAXROWS / rpb;
    copies = copies > 8 ? 8 : copies;  // (both powers of two)
    const uint32_t copy_off = (uint32_t)(lane & (copies - 1)) * (uint32_t)rpb;
    for (int i = tid; i < rpb * copies * F; i += HG_FX_T) acc[i] = 0ull;
    for (int i = tid; i < MAXROWS * F / 32; i += HG_FX_T) bad[i] = 0u;
    // 2^e > M >= every finite |g| of the level; q = rint(c * 2^(BITS - e)).  (M = 0: nothing finite and non-zero lands here)
    int e = 0;
    frexpf(__uint_as_float(lvl_absmax_bits[l]), &e);
    int sh = HG_FX_BITS - e;
    sh = sh > 120 ? 120 : sh;  // keeps 2^sh and 2^-sh normal floats; levels whose largest gradient is below 2^-82 lose nothing that matters
    const float scale = ldexpf(1.f, sh), inv = ldexpf(1.f, -sh);
    const float* __restrict__ gl = gT + (size_t)l * N * F;
    // (wave-uniform.  Small tables already spread a row over `copies` accumulators: merging on top of 4 copies measured slower,
    //  proposal grid 84 -> 96 us; without copies it takes the field grid's five coarse levels from 354 to 219 us)
    const bool merge = merge_level && copies < 4;
    __syncthreads();
    // three-stage software pipeline over trips of U records per thread: the records of trip t+2 and the staged gradients of
    // trip t+1 are in flight while trip t is accumulated, so neither global latency sits on the loop's critical path
#define SNF_FX_U 4
#ifndef SNF_FX_UXP
#define SNF_FX_UXP 2
#endif
    constexpr int U = XP ? SNF_FX_UXP : SNF_FX_U;  // (a pair record is two corners: the same work in flight with half the registers)
    const uint32_t mask_s = (1u << HG_SAMPLE_BITS) - 1u;
    Rec rec0[U], rec1[U];        // records of trip t (being processed), t+1 (gathers in flight)
    float g0[U][F], g1[U][F];
    auto load_recs = [&](uint32_t c0, Rec (&r)[U]) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const uint32_t i = c0 + tid + (uint32_t)HG_FX_T * j;
            r[j] = hg_ld_rec(&records[i < end ? i : (start < end ? start : 0u)], (SNF_FX_NT & 1) != 0);  // (an empty bucket is still visited when ADAM: never read past the array)
        }
    };
    auto gather = [&](const Rec (&r)[U], float (&g)[U][F]) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            // (SPLIT > 1: only the records of this workgroup's rows cost a gather)
            if (SPLIT == 1 || ((r[j].x >> HG_SAMPLE_BITS) - row0) < (uint32_t)rpb)
                load_row<F>(gl + (size_t)(r[j].x & mask_s) * F, g[j]);
            else {
#pragma unroll
                for (int f = 0; f < F; ++f) g[j][f] = 0.f;
            }
        }
    };
    constexpr uint32_t TRIP = HG_FX_T * U;
    load_recs(start, rec0);
    gather(rec0, g0);
    if (start + TRIP < end) load_recs(start + TRIP, rec1);
    for (uint32_t c0 = start; c0 < end; c0 += TRIP) {
        Rec rec2[U];
        const bool has1 = c0 + TRIP < end, has2 = c0 + 2 * TRIP < end;
        if (has1) gather(rec1, g1);
        if (has2) load_recs(c0 + 2 * TRIP, rec2);
#pragma unroll
        for (int j = 0; j < U; ++j) {
            bool live = c0 + tid + (uint32_t)HG_FX_T * j < end;
            if constexpr (XP) {
                // a pair: rows (row_a, row_b) of the bucket, weights (w_a, w_b), one staged gradient; b may be absent
                const uint32_t ra = rec0[j].x >> HG_SAMPLE_BITS, rb = rec0[j].y & 0x7FFu;
                const bool vb = (rec0[j].y & 0x800u) != 0u;
                const float wa = __uint_as_float(rec0[j].z), wb = __uint_as_float(rec0[j].w);
                float v[2 * F];
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    v[f] = wa * g0[j][f];
                    v[F + f] = vb ? wb * g0[j][f] : 0.f;
                }
                long long q[2 * F];
#pragma unroll
                for (int e2 = 0; e2 < 2 * F; ++e2) {
                    const bool ok = fabsf(v[e2]) < INFINITY;
                    q[e2] = ok ? __float2ll_rn(v[e2] * scale) : 0ll;
                    if (live && !ok) {  // NaN / inf cannot be represented: the element becomes NaN, as a float sum would
                        const uint32_t eb = (e2 < F ? ra : rb) * F + (uint32_t)(e2 % F);
                        atomicOr(&bad[eb >> 5], 1u << (eb & 31));
                    }
                }
                int alive = live ? 1 : 0;
                if (merge) {  // (coarse level: equal pairs inside a quad are added before the atomics, as for single records below)
                    const uint32_t key = live ? (ra | ((rec0[j].y & 0xFFFu) << 11)) : (0xFF000000u | (uint32_t)lane);
#define SNF_QUAD_STEP_XP(CTRL, BIT)                                                                                      \
                    {                                                                                                      \
                        const uint32_t pk = (uint32_t)__builtin_amdgcn_mov_dpp((int)key, CTRL, 0xF, 0xF, true);           \
                        const int pa = __builtin_amdgcn_mov_dpp(alive, CTRL, 0xF, 0xF, true);                              \
                        const bool same = alive && pa && pk == key;                                                        \
                        _Pragma("unroll") for (int e2 = 0; e2 < 2 * F; ++e2) {                                             \
                            const int lo = __builtin_amdgcn_mov_dpp((int)(uint32_t)(unsigned long long)q[e2], CTRL, 0xF, 0xF, true);          \
                            const int hi = __builtin_amdgcn_mov_dpp((int)(uint32_t)((unsigned long long)q[e2] >> 32), CTRL, 0xF, 0xF, true);  \
                            const long long pq = (long long)(((unsigned long long)(uint32_t)hi << 32) | (unsigned long long)(uint32_t)lo);    \
                            if (same && !(lane & BIT)) q[e2] += pq;                                                        \
                        }                                                                                                  \
                        if (same && (lane & BIT)) alive = 0;                                                               \
                    }
                    SNF_QUAD_STEP_XP(0xB1, 1)
                    SNF_QUAD_STEP_XP(0x4E, 2)
#undef SNF_QUAD_STEP_XP
                }
                if (alive) {
#pragma unroll
                    for (int f = 0; f < F; ++f) {
                        if (q[f] != 0) atomicAdd(&acc[(copy_off + ra) * F + f], (unsigned long long)q[f]);
                        if (q[F + f] != 0) atomicAdd(&acc[(copy_off + rb) * F + f], (unsigned long long)q[F + f]);
                    }
                }
                continue;
            }
            const uint32_t row = (rec0[j].x >> HG_SAMPLE_BITS) - row0;  // row inside this workgroup's share
            if (SPLIT > 1) live = live && row < (uint32_t)rpb;
            const float w = __uint_as_float(rec0[j].y);
            float v[F];
#pragma unroll
            for (int f = 0; f < F; ++f) v[f] = w * g0[j][f];
            // (no run aggregation here: the segmented shuffle scan the float reduce uses on coarse levels costs more than the
            // same-address integer LDS atomics it saves -- level 16^3 alone 333 -> 188 us, the proposal grid 156 -> 106 us)
            if (F == 2 && merge) {
                // Coarse level: the lanes of a quad mostly carry ONE row (neighbouring samples of a ray in one cell), and
                // same-address LDS atomics serialise.  The integer contributions of equal rows are added inside the quad first --
                // two DPP exchanges (lane ^ 1, then lane ^ 2 among the survivors), exact, so the result stays order-independent --
                // and one lane per group issues the atomic: up to 4x fewer conflicting atomics for ~24 VALU instructions (the
                // segmented 64-lane shuffle scan of the float reduce cost more than it saved here).
                long long q[F];
                bool fin = true;
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    const bool ok = fabsf(v[f]) < INFINITY;
                    fin = fin && ok;
                    q[f] = ok ? __float2ll_rn(v[f] * scale) : 0ll;
                }
                if (live && !fin) {
#pragma unroll
                    for (int f = 0; f < F; ++f) {
                        if (!(fabsf(v[f]) < INFINITY)) {
                            const uint32_t e = row * F + f;
                            atomicOr(&bad[e >> 5], 1u << (e & 31));
                        }
                    }
                }
                uint32_t key = live ? row : (0xFFFF0000u | (uint32_t)lane);  // dead lanes never match a neighbour
                int alive = live ? 1 : 0;
                // step 1: partner lane ^ 1 (quad_perm [1,0,3,2]); step 2: partner lane ^ 2 (quad_perm [2,3,0,1])
#define SNF_QUAD_STEP(CTRL, BIT)                                                                                         \
                {                                                                                                          \
                    const uint32_t pk = (uint32_t)__builtin_amdgcn_mov_dpp((int)key, CTRL, 0xF, 0xF, true);               \
                    const int pa = __builtin_amdgcn_mov_dpp(alive, CTRL, 0xF, 0xF, true);                                  \
                    const bool same = alive && pa && pk == key;                                                            \
                    _Pragma("unroll") for (int f = 0; f < F; ++f) {                                                        \
                        const int lo = __builtin_amdgcn_mov_dpp((int)(uint32_t)(unsigned long long)q[f], CTRL, 0xF, 0xF, true);          \
                        const int hi = __builtin_amdgcn_mov_dpp((int)(uint32_t)((unsigned long long)q[f] >> 32), CTRL, 0xF, 0xF, true);  \
                        const long long pq = (long long)(((unsigned long long)(uint32_t)hi << 32) | (unsigned long long)(uint32_t)lo);   \
                        if (same && !(lane & BIT)) q[f] += pq;                                                             \
                    }                                                                                                      \
                    if (same && (lane & BIT)) alive = 0;                                                                   \
                }
                SNF_QUAD_STEP(0xB1, 1)
                SNF_QUAD_STEP(0x4E, 2)
#undef SNF_QUAD_STEP
                if (alive) {
#pragma unroll
                    for (int f = 0; f < F; ++f)
                        if (q[f] != 0) atomicAdd(&acc[(copy_off + row) * F + f], (unsigned long long)q[f]);
                }
            } else if (live) {
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    if (fabsf(v[f]) < INFINITY) {
                        const long long q = __float2ll_rn(v[f] * scale);
                        if (q != 0) atomicAdd(&acc[(copy_off + row) * F + f], (unsigned long long)q);
                    } else {  // NaN / inf cannot be represented: the element becomes NaN, as a float sum would
                        const uint32_t e = row * F + f;
                        atomicOr(&bad[e >> 5], 1u << (e & 31));
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            rec0[j] = rec1[j];
            rec1[j] = rec2[j];
#pragma unroll
            for (int f = 0; f < F; ++f) g0[j][f] = g1[j][f];
        }
    }
    __syncthreads();
    // ---- epilogue: each thread converts its rows (tid, tid + T, ...), four row sets in flight
    const size_t base = (((size_t)l << log2_T) + ((size_t)b << log2rpb) + row0) * F;
    float* __restrict__ slab = grad_table + base;
    constexpr int RU = F == 2 ? 4 : 2;
    for (int r0 = tid; r0 < rpb; r0 += HG_FX_T * RU) {
        float gg[RU][F], pp[RU][F], mm[RU][F], vv[RU][F];
        bool on[RU], nz[RU];
        // fused step: the optimizer state of all RU rows is requested FIRST (unconditional, the row clamped into the bucket) and the
        // sums come out of LDS under those loads.  With the loads inside the per-row `if (on)` every row's three loads were followed by
        // their own s_waitcnt vmcnt(0): four memory latencies in a row per thread, after the LDS conversion instead of under it.
        if (fuse) {
#pragma unroll
            for (int j = 0; j < RU; ++j) {
                const int rc = min(r0 + HG_FX_T * j, rpb - 1);
                if constexpr (F == 2 && (SNF_FX_NT & 2) != 0) {
                    const hg_f2v a = __builtin_nontemporal_load(reinterpret_cast<const hg_f2v*>(adam.p + base + (size_t)rc * F));
                    const hg_f2v b2 = __builtin_nontemporal_load(reinterpret_cast<const hg_f2v*>(adam.m + base + (size_t)rc * F));
                    const hg_f2v c2 = __builtin_nontemporal_load(reinterpret_cast<const hg_f2v*>(adam.v + base + (size_t)rc * F));
                    pp[j][0] = a.x; pp[j][1] = a.y; mm[j][0] = b2.x; mm[j][1] = b2.y; vv[j][0] = c2.x; vv[j][1] = c2.y;
                } else {
                    load_row<F>(adam.p + base + (size_t)rc * F, pp[j]);
                    load_row<F>(adam.m + base + (size_t)rc * F, mm[j]);
                    load_row<F>(adam.v + base + (size_t)rc * F, vv[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < RU; ++j) {
            const int r = r0 + HG_FX_T * j;
            on[j] = r < rpb;
            nz[j] = false;
            if (on[j]) {
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    const uint32_t e = (uint32_t)(r * F + f);
                    const bool isbad = (bad[e >> 5] >> (e & 31)) & 1u;
                    unsigned long long qs = acc[e];
                    for (int c = 1; c < copies; ++c) qs += acc[(uint32_t)c * (uint32_t)(rpb * F) + e];
                    const long long q = (long long)qs;
                    gg[j][f] = isbad ? __uint_as_float(0x7FC00000u) : __ll2float_rn(q) * inv;
                    nz[j] = nz[j] || (q != 0) || isbad;
                }
                if (!fuse && nz[j]) load_row<F>(slab + (size_t)r * F, pp[j]);  // the running gradient of a level left to the caller's optimizer
            }
        }
#pragma unroll
        for (int j = 0; j < RU; ++j) {
            if (!on[j]) continue;
            const size_t o = base + (size_t)(r0 + HG_FX_T * j) * F;
            if (fuse) {
#pragma unroll
                for (int f = 0; f < F; ++f) hg_adam1(pp[j][f], gg[j][f], mm[j][f], vv[j][f], adam);
                if constexpr (F == 2 && (SNF_FX_NT & 4) != 0) {
                    const hg_f2v a = {pp[j][0], pp[j][1]}, b2 = {mm[j][0], mm[j][1]}, c2 = {vv[j][0], vv[j][1]};
                    __builtin_nontemporal_store(a, reinterpret_cast<hg_f2v*>(adam.p + o));
                    __builtin_nontemporal_store(b2, reinterpret_cast<hg_f2v*>(adam.m + o));
                    __builtin_nontemporal_store(c2, reinterpret_cast<hg_f2v*>(adam.v + o));
                } else if constexpr (F == 2) {
                    *reinterpret_cast<float2*>(adam.p + o) = make_float2(pp[j][0], pp[j][1]);
                    *reinterpret_cast<float2*>(adam.m + o) = make_float2(mm[j][0], mm[j][1]);
                    *reinterpret_cast<float2*>(adam.v + o) = make_float2(vv[j][0], vv[j][1]);
                } else {
                    reinterpret_cast<float4*>(adam.p + o)[0] = make_float4(pp[j][0], pp[j][1], pp[j][2], pp[j][3]);
                    reinterpret_cast<float4*>(adam.p + o)[1] = make_float4(pp[j][4], pp[j][5], pp[j][6], pp[j][7]);
                    reinterpret_cast<float4*>(adam.m + o)[0] = make_float4(mm[j][0], mm[j][1], mm[j][2], mm[j][3]);
                    reinterpret_cast<float4*>(adam.m + o)[1] = make_float4(mm[j][4], mm[j][5], mm[j][6], mm[j][7]);
                    reinterpret_cast<float4*>(adam.v + o)[0] = make_float4(vv[j][0], vv[j][1], vv[j][2], vv[j][3]);
                    reinterpret_cast<float4*>(adam.v + o)[1] = make_float4(vv[j][4], vv[j][5], vv[j][6], vv[j][7]);
                }
            } else if (nz[j]) {
                float* dst = grad_table + o;
                if constexpr (F == 2) {
                    *reinterpret_cast<float2*>(dst) = make_float2(pp[j][0] + gg[j][0], pp[j][1] + gg[j][1]);
                } else {
                    reinterpret_cast<float4*>(dst)[0] = make_float4(pp[j][0] + gg[j][0], pp[j][1] + gg[j][1], pp[j][2] + gg[j][2], pp[j][3] + gg[j][3]);
                    reinterpret_cast<float4*>(dst)[1] = make_float4(pp[j][4] + gg[j][4], pp[j][5] + gg[j][5], pp[j][6] + gg[j][6], pp[j][7] + gg[j][7]);
                }
            }
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
    SNF_REQUIRE((ld_out == 0 && col_off == 0) || (ld_out >= col_off + L * F && col_off >= 0),
                "%s: ld_out=%d too small for col_off=%d + L*F=%d (ld_out = 0: level-major [L][N][F], col_off must be 0)", who,
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
                           out, ld_out, col_off, 0);
    else
        hipLaunchKernelGGL(k_hashgrid_fwd<8>, grid, dim3(256), 0, (hipStream_t)stream, u, table, scalings, N, log2_T,
                           out, ld_out, col_off, 0);
    SNF_LAUNCH_CHECK("snf_hashgrid_fwd");
    return SNF_OK;
}

// density[n] = exp(mlp(hashgrid(u[n]))) * selector[n]: snf_hashgrid_fwd (row-major) + snf_mlp_tiny_fwd + snf_trunc_exp_fwd in one
// launch, for the proposal net's shape (L = 5 levels, F = 2, 10 -> 16 -> 1: snf_mlp_tiny_supported); no intermediate is written
extern "C" int snf_prop_density_fwd(const float* u, const float* table, const float* scalings, int N, int L, int F, int log2_T,
                                    const float* W0, const float* W1, int H, const uint8_t* selector, float* density,
                                    snf_stream_t stream) {
    SNF_REQUIRE(u && table && scalings && W0 && W1 && density && N > 0, "snf_prop_density_fwd: bad argument");
    SNF_REQUIRE(L == 5 && F == 2 && H == 16, "snf_prop_density_fwd: only the 5-level F = 2 grid with the 10 -> 16 -> 1 net is built (L=%d F=%d H=%d)", L, F, H);
    SNF_REQUIRE(log2_T >= 1 && log2_T <= 26 && ((uintptr_t)table % 16) == 0, "snf_prop_density_fwd: bad table");
    hipLaunchKernelGGL(k_prop_density_fwd<5>, dim3(ceil_div(N, 256)), dim3(256), 0, (hipStream_t)stream, u, table, scalings, N, log2_T,
                       W0, W1, selector, density);
    SNF_LAUNCH_CHECK("snf_prop_density_fwd");
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
constexpr size_t HG_FX_SCRATCH = 64;  // words (L <= 64 for the fixed-point reduce)

// fixed-point reduce for the F = 2 grids (SNF_HG_FX=0: the float reduce everywhere)
// first level of the XCD-aware workgroup order of the fixed-point reduce (k_hg_reduce_fx): the leading `n_run_levels` coarse levels
// keep the plain order; SNF_HG_XCD=0 switches it off (returns L)
static int hg_xcd_from(int n_run_levels, int L, int B) {
    static const int on = 1;
    if (!on || (B % 8) != 0 || n_run_levels >= L) return L;
    return n_run_levels < 0 ? 0 : n_run_levels;
}

static HgSecond hg_no_second() {
    HgSecond s{};
    s.first_levels = 1 << 30;
    return s;
}

// leading levels whose equal-row contributions are merged inside lane quads before the LDS atomics (SNF_HG_MERGE=0: none)
static int hg_merge_levels(int n_run_levels) {
    static const int on = 1;
    return on ? n_run_levels : 0;
}

static bool hg_fx_on(int F, int L, int N) {
    static const int on = 1;
    return on && F == 2 && L <= (int)HG_FX_SCRATCH && (N % 2) == 0;
}

static size_t hg_ws_words(int N, int L, int log2_T) {
    const HgGeom g = hg_geometry(N, log2_T);
    const size_t B = (size_t)1 << g.log2B;
    // records (8 N per level: 8 B each, or 16-byte x-pair records whose worst case -- every pair split -- is 8 N as well) | tile
    // histograms | tile offsets | bucket starts (padded to 4 words) | per-level scratch of the fixed-point reduce (HG_FX_SCRATCH
    // words, the only part a BACKWARD writes) | staged gradients
    return (size_t)L * 8 * (size_t)N * 4 + 2 * (size_t)L * g.nblk * B + (((size_t)L * (B + 1) + 3) & ~(size_t)3) +
           HG_FX_SCRATCH + (size_t)L * (size_t)N * 8;
}

extern "C" int64_t snf_hashgrid_bwd_workspace_bytes(int N, int L, int log2_T) {
    if (N <= 0 || L <= 0 || log2_T < 1) return 0;
    return (int64_t)(hg_ws_words(N, L, log2_T) * sizeof(uint32_t));
}

extern "C" int snf_hashgrid_bwd_sorted(const float* u, const float* grad_out, const float* scalings, int N, int L, int F,
                                       int log2_T, int ld_out, int col_off, float* grad_table, void* workspace,
                                       int64_t workspace_bytes, snf_stream_t stream) {
    return snf_hashgrid_bwd_sorted_ex(u, grad_out, scalings, N, L, F, log2_T, ld_out, col_off, 0, grad_table, workspace,
                                      workspace_bytes, stream);
}

// The sorted backward in two halves.  The sort (count / scan / scatter) depends on the positions and the level geometry only --
// not on the upstream gradient, the table or F -- so grids of equal geometry evaluated at the same points (the SAM and
// ClipSeg feature grids) share one sort, and it can run as soon as the positions exist (in the forward pass).
static int hg_sort_checks(const char* who, int N, int L, int log2_T, const void* workspace, int64_t workspace_bytes) {
    SNF_REQUIRE(N > 0 && L > 0 && log2_T >= 1 && log2_T <= 26, "%s: bad N=%d L=%d log2_T=%d", who, N, L, log2_T);
    SNF_REQUIRE((long long)L * 8 * N < (1LL << 32) && N <= (1 << HG_SAMPLE_BITS),
                "%s: at most 2^21 samples per call (N=%d L=%d); split the batch", who, N, L);
    SNF_REQUIRE(workspace && workspace_bytes >= (int64_t)(hg_ws_words(N, L, log2_T) * sizeof(uint32_t)),
                "%s: workspace too small (need %lld bytes)", who, (long long)(hg_ws_words(N, L, log2_T) * sizeof(uint32_t)));
    SNF_REQUIRE(((uintptr_t)workspace % 16) == 0, "%s: unaligned workspace", who);
    return SNF_OK;
}

static int hg_absmax_blocks(int N, int F) {
    long long n = ((long long)N * F / 4 + 255) / 256 / 16;  // ~16 float4 per thread
    return (int)(n < 1 ? 1 : (n > 64 ? 64 : n));
}

struct HgWs {
    uint32_t *records, *hist, *offs, *bstart, *fx;
    float* gT;
};

static HgWs hg_ws_layout(void* workspace, int N, int L, const HgGeom& g) {
    const size_t B = (size_t)1 << g.log2B;
    HgWs w;
    w.records = (uint32_t*)workspace;
    w.hist = w.records + (size_t)L * 8 * (size_t)N * 4;
    w.offs = w.hist + (size_t)L * g.nblk * B;
    w.bstart = w.offs + (size_t)L * g.nblk * B;
    w.fx = w.bstart + (((size_t)L * (B + 1) + 3) & ~(size_t)3);
    w.gT = (float*)(w.fx + HG_FX_SCRATCH);
    return w;
}

// second half of the sort from the tile histograms in w.hist: offsets + record scatter.  Up to HG_SELF_SCAN_TILES tiles per level the
// scatter computes its offsets itself (one launch less: the feature grids' 64-tile sort 76 -> 70 us); at 128 tiles re-reading the
// level's histogram in every tile costs more than the scan kernel (field grid: 270 vs 261 us), which then writes them out.
constexpr int HG_SELF_SCAN_TILES = 64;
static void hg_scatter_from_hist(hipStream_t st, const float* u, const float* scalings, int N, int L, int log2_T, const HgGeom& g,
                                 const HgWs& w, bool xp = false) {
    static int lds_attr[2] = {0, 0};  // largest dynamic-LDS size each scatter instantiation has been opened for
    const bool self = g.nblk <= HG_SELF_SCAN_TILES;
    if (xp) {
        static int lds_attr_xp[2] = {0, 0};
        const int ldx = (int)hg_scatter_xp_lds_bytes(g.log2B);
        if (ldx > 48 * 1024 && ldx > lds_attr_xp[self ? 1 : 0]) {
            lds_attr_xp[self ? 1 : 0] = ldx;
            if (self) (void)hipFuncSetAttribute((const void*)k_hg_scatter_xp<true>, hipFuncAttributeMaxDynamicSharedMemorySize, ldx);
            else (void)hipFuncSetAttribute((const void*)k_hg_scatter_xp<false>, hipFuncAttributeMaxDynamicSharedMemorySize, ldx);
        }
        if (self) {
            hipLaunchKernelGGL(k_hg_scatter_xp<true>, dim3(g.nblk, L), dim3(256), ldx, st, u, scalings, N, log2_T, g.log2B, g.spt, w.hist,
                               (uint4*)w.records, w.bstart);
        } else {
            hipLaunchKernelGGL(k_hg_scan, dim3(L), dim3(1024), 0, st, N, g.log2B, g.nblk, w.hist, w.offs, w.bstart);
            hipLaunchKernelGGL(k_hg_scatter_xp<false>, dim3(g.nblk, L), dim3(256), ldx, st, u, scalings, N, log2_T, g.log2B, g.spt,
                               w.offs, (uint4*)w.records, w.bstart);
        }
        return;
    }
    const int lds = (int)hg_scatter_lds_bytes(g.log2B);
    if (lds > 48 * 1024 && lds > lds_attr[self ? 1 : 0]) {
        lds_attr[self ? 1 : 0] = lds;
        if (self) (void)hipFuncSetAttribute((const void*)k_hg_scatter<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        else (void)hipFuncSetAttribute((const void*)k_hg_scatter<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
    if (self) {
        hipLaunchKernelGGL(k_hg_scatter<true>, dim3(g.nblk, L), dim3(256), lds, st, u, scalings, N, log2_T, g.log2B, g.spt, w.hist,
                           (uint2*)w.records, w.bstart);
    } else {
        hipLaunchKernelGGL(k_hg_scan, dim3(L), dim3(1024), 0, st, N, g.log2B, g.nblk, w.hist, w.offs, w.bstart);
        hipLaunchKernelGGL(k_hg_scatter<false>, dim3(g.nblk, L), dim3(256), lds, st, u, scalings, N, log2_T, g.log2B, g.spt, w.offs,
                           (uint2*)w.records, w.bstart);
    }
}

static int hg_sort_impl(const char* who, const float* u, const float* scalings, int N, int L, int log2_T, void* workspace,
                        int64_t workspace_bytes, bool xp, snf_stream_t stream) {
    SNF_REQUIRE(u && scalings, "%s: null pointer", who);
    int rc = hg_sort_checks(who, N, L, log2_T, workspace, workspace_bytes);
    if (rc) return rc;
    const HgGeom g = hg_geometry(N, log2_T);
    SNF_REQUIRE((1 << g.log2rpb) <= HG_MAX_RPB, "%s: log2_T=%d too large for the bucketed backward", who, log2_T);
    const HgWs w = hg_ws_layout(workspace, N, L, g);
    hipStream_t st = (hipStream_t)stream;
    if (xp) hipLaunchKernelGGL(k_hg_count<true>, dim3(g.nblk, L), dim3(256), 0, st, u, scalings, N, log2_T, g.log2B, g.spt, w.hist);
    else hipLaunchKernelGGL(k_hg_count<false>, dim3(g.nblk, L), dim3(256), 0, st, u, scalings, N, log2_T, g.log2B, g.spt, w.hist);
    hg_scatter_from_hist(st, u, scalings, N, L, log2_T, g, w, xp);
    SNF_LAUNCH_CHECK(who);
    return SNF_OK;
}

extern "C" int snf_hashgrid_sort(const float* u, const float* scalings, int N, int L, int log2_T, void* workspace,
                                 int64_t workspace_bytes, snf_stream_t stream) {
    return hg_sort_impl("snf_hashgrid_sort", u, scalings, N, L, log2_T, workspace, workspace_bytes, false, stream);
}

// The sort with x-pair records (see k_hg_scatter_xp): for F = 2 grids; its workspace is consumed by snf_hashgrid_bwd_presorted_adam_xp
// ONLY (the other backward entry points read 8-byte single-corner records).
extern "C" int snf_hashgrid_sort_xp(const float* u, const float* scalings, int N, int L, int log2_T, void* workspace,
                                    int64_t workspace_bytes, snf_stream_t stream) {
    return hg_sort_impl("snf_hashgrid_sort_xp", u, scalings, N, L, log2_T, workspace, workspace_bytes, true, stream);
}

// What the reduce pass of ONE table needs besides the records: the reachable-row lists of its leading `levels` levels
// (Encoding.reach_lists: sorted (level << T) + row per (level, bucket), and the [levels * B + 1] list offsets).  levels = 0: none,
// every level goes through the bucket-wide kernels.
struct HgSparse {
    const uint32_t* rows;
    const uint32_t* start;
    int levels;
};

static void hg_fill_adam(HgAdam& a, float* param, float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2, float eps,
                         int step, float grad_scale, int from_level) {
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    a.p = param; a.m = exp_avg; a.v = exp_avg_sq;
    a.b1 = beta1; a.b2 = beta2; a.step_size = (float)((double)lr / bc1); a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    a.eps = eps; a.gs = grad_scale; a.from_level = from_level;
    a.guard = current_guard(); a.lr = lr; a.step = step; a.veto = 0;
}

template <int F>
static void hg_launch_sparse(hipStream_t st, const float* stage, int N, int log2_T, const HgGeom& g, const HgWs& w, float* grad_table,
                             const HgSparse& sp, const uint32_t* lvlmax, bool adam_on, const HgAdam& a) {
    const int B = 1 << g.log2B;
    const size_t lds = hg_sp_lds_bytes<F>(g.log2rpb);
    static bool attr_done[2] = {false, false};
    if (lds > 48 * 1024 && !attr_done[adam_on ? 1 : 0]) {
        attr_done[adam_on ? 1 : 0] = true;
        if (adam_on) (void)hipFuncSetAttribute((const void*)k_hg_reduce_sparse<F, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        else (void)hipFuncSetAttribute((const void*)k_hg_reduce_sparse<F, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (adam_on)
        hipLaunchKernelGGL((k_hg_reduce_sparse<F, true>), dim3(B, sp.levels), dim3(HG_SP_T), lds, st, stage, N, log2_T, g.log2B, w.bstart,
                           (const uint2*)w.records, grad_table, sp.rows, sp.start, lvlmax, a);
    else
        hipLaunchKernelGGL((k_hg_reduce_sparse<F, false>), dim3(B, sp.levels), dim3(HG_SP_T), lds, st, stage, N, log2_T, g.log2B, w.bstart,
                           (const uint2*)w.records, grad_table, sp.rows, sp.start, lvlmax, a);
}

// The reduce pass of one table from its staged (level-major) gradient: reachable-row levels [0, sp.levels) through
// k_hg_reduce_sparse (stepped there when adam_on), the rest through the fixed-point reduce (F = 2) or the float reduce (F = 8)
// with Adam fused on the levels >= a.from_level.  lvlmax: per-level scratch of the fixed-point kernels (HG_FX_SCRATCH words).
static void hg_reduce_one(hipStream_t st, int F, const float* stage, int N, int L, int log2_T, const HgGeom& g, const HgWs& w,
                          float* grad_table, int n_run_levels, bool adam_on, bool sparse_step, HgAdam a, HgSparse sp,
                          uint32_t* lvlmax, bool xp = false) {
    const int B = 1 << g.log2B;
    static const uint32_t hg_long = (uint32_t)HG_LONG;
    const bool fx = hg_fx_on(F, L, N);
    const int s0 = sp.levels;  // first level of the bucket-wide kernels
    const int nfx = fx ? L : 0;  // levels whose largest |g| the bucket-wide fixed-point kernel needs (the compact reduce finds its own)
    if (nfx > 0) {
        (void)hipMemsetAsync(lvlmax, 0, nfx * sizeof(uint32_t), st);
        if (F == 2) hipLaunchKernelGGL(k_hg_level_absmax<2>, dim3(hg_absmax_blocks(N, F), nfx), dim3(256), 0, st, stage, N, lvlmax);
        else hipLaunchKernelGGL(k_hg_level_absmax<8>, dim3(hg_absmax_blocks(N, F), nfx), dim3(256), 0, st, stage, N, lvlmax);
    }
    if (F == 8 && s0 > 0) {
        // one launch for the whole table: its reachable-row levels are workgroups of the same grid (compact reduce in the float
        // reduce's LDS), resident beside the bucket-wide levels
        const HgSparseDev sd{sp.rows, sp.start, lvlmax, s0, sparse_step ? 1 : 0};
        if (adam_on || sparse_step)
            hipLaunchKernelGGL((k_hg_reduce<8, true>), dim3(B, L), dim3(HG_RT), 0, st, stage, N, log2_T, g.log2B, w.bstart,
                               (const uint2*)w.records, grad_table, hg_long, 0, 0, a, sd, hg_no_second());
        else
            hipLaunchKernelGGL((k_hg_reduce<8, false>), dim3(B, L), dim3(HG_RT), 0, st, stage, N, log2_T, g.log2B, w.bstart,
                               (const uint2*)w.records, grad_table, hg_long, 0, 0, a, sd, hg_no_second());
        return;
    }
    if (s0 > 0) hg_launch_sparse<2>(st, stage, N, log2_T, g, w, grad_table, sp, lvlmax, sparse_step, a);
    const int Lr = L - s0;
    if (Lr <= 0) return;
    const int nrun = n_run_levels > s0 ? n_run_levels - s0 : 0;  // (relative to the launch)
    if (F == 2 && fx && xp) {
        if (adam_on)
            hipLaunchKernelGGL((k_hg_reduce_fx<2, true, 1, true>), dim3(B, Lr), dim3(HG_FX_T), 0, st, stage, N, log2_T, g.log2B, w.bstart,
                               (const uint4*)w.records, grad_table, hg_xcd_from(nrun, Lr, B), hg_merge_levels(nrun), s0, lvlmax, a);
        else
            hipLaunchKernelGGL((k_hg_reduce_fx<2, false, 1, true>), dim3(B, Lr), dim3(HG_FX_T), 0, st, stage, N, log2_T, g.log2B,
                               w.bstart, (const uint4*)w.records, grad_table, hg_xcd_from(nrun, Lr, B), hg_merge_levels(nrun), s0, lvlmax,
                               HgAdam{});
    } else if (F == 2 && fx) {
        if (adam_on)
            hipLaunchKernelGGL((k_hg_reduce_fx<2, true, 1>), dim3(B, Lr), dim3(HG_FX_T), 0, st, stage, N, log2_T, g.log2B, w.bstart,
                               (const uint2*)w.records, grad_table, hg_xcd_from(nrun, Lr, B), hg_merge_levels(nrun), s0, lvlmax, a);
        else
            hipLaunchKernelGGL((k_hg_reduce_fx<2, false, 1>), dim3(B, Lr), dim3(HG_FX_T), 0, st, stage, N, log2_T, g.log2B, w.bstart,
                               (const uint2*)w.records, grad_table, hg_xcd_from(nrun, Lr, B), hg_merge_levels(nrun), s0, lvlmax, HgAdam{});
    } else if (F == 2) {
        if (adam_on)
            hipLaunchKernelGGL((k_hg_reduce<2, true>), dim3(B, Lr), dim3(HG_RT), 0, st, stage, N, log2_T, g.log2B, w.bstart,
                               (const uint2*)w.records, grad_table, hg_long, nrun, s0, a, HgSparseDev{}, hg_no_second());
        else
            hipLaunchKernelGGL((k_hg_reduce<2, false>), dim3(B, Lr), dim3(HG_RT), 0, st, stage, N, log2_T, g.log2B, w.bstart,
                               (const uint2*)w.records, grad_table, hg_long, nrun, s0, HgAdam{}, HgSparseDev{}, hg_no_second());
    } else {
        if (adam_on)
            hipLaunchKernelGGL((k_hg_reduce<8, true>), dim3(B, Lr), dim3(HG_RT), 0, st, stage, N, log2_T, g.log2B, w.bstart,
                               (const uint2*)w.records, grad_table, hg_long, nrun, s0, a, HgSparseDev{}, hg_no_second());
        else
            hipLaunchKernelGGL((k_hg_reduce<8, false>), dim3(B, Lr), dim3(HG_RT), 0, st, stage, N, log2_T, g.log2B, w.bstart,
                               (const uint2*)w.records, grad_table, hg_long, nrun, s0, HgAdam{}, HgSparseDev{}, hg_no_second());
    }
}

static void hg_stage(hipStream_t st, int F, const float* grad_out, int N, int L, int ld_out, int col_off, float* stage) {
    const int tblocks = ceil_div((long long)N * L, 256);
    if (F == 2) hipLaunchKernelGGL(k_hg_stage_grad<2>, dim3(tblocks), dim3(256), 0, st, grad_out, N, L, ld_out, col_off, stage);
    else hipLaunchKernelGGL(k_hg_stage_grad<8>, dim3(tblocks), dim3(256), 0, st, grad_out, N, L, ld_out, col_off, stage);
}

extern "C" int snf_hashgrid_bucket_bits(int N, int log2_T) {
    if (N <= 0 || log2_T < 1 || log2_T > 26) return -1;
    return hg_geometry(N, log2_T).log2B;
}

extern "C" int snf_hashgrid_sparse_max_rows(int F) { return F == 8 ? hg_sp_max_rows<8>() : (F == 2 ? hg_sp_max_rows<2>() : 0); }

extern "C" int snf_hashgrid_bwd_presorted(const float* grad_out, int N, int L, int F, int log2_T, int ld_out, int col_off,
                                          int n_run_levels, float* grad_table, const void* sorted_workspace, float* stage,
                                          snf_stream_t stream) {
    const bool planar = ld_out == 0;  // grad_out is already level-major [L][N][F]: it IS the staged gradient
    SNF_REQUIRE(grad_out && grad_table && sorted_workspace && (stage || planar), "snf_hashgrid_bwd_presorted: null pointer");
    SNF_REQUIRE(F == 2 || F == 8, "snf_hashgrid_bwd_presorted: features_per_level must be 2 or 8 (got %d)", F);
    SNF_REQUIRE(N > 0 && L > 0 && N <= (1 << HG_SAMPLE_BITS) && ((planar && col_off == 0) || ld_out >= col_off + L * F) &&
                    col_off >= 0,
                "snf_hashgrid_bwd_presorted: bad shape N=%d L=%d ld_out=%d col_off=%d", N, L, ld_out, col_off);
    SNF_REQUIRE(((uintptr_t)grad_out % 16) == 0 && ((uintptr_t)grad_table % 16) == 0 && ((uintptr_t)stage % 16) == 0,
                "snf_hashgrid_bwd_presorted: unaligned pointer");
    if (planar) stage = const_cast<float*>(grad_out);
    const HgGeom g = hg_geometry(N, log2_T);
    const HgWs w = hg_ws_layout(const_cast<void*>(sorted_workspace), N, L, g);
    hipStream_t st = (hipStream_t)stream;
    if (!planar) hg_stage(st, F, grad_out, N, L, ld_out, col_off, stage);
    hg_reduce_one(st, F, stage, N, L, log2_T, g, w, grad_table, n_run_levels, false, false, HgAdam{}, HgSparse{nullptr, nullptr, 0}, w.fx);
    SNF_LAUNCH_CHECK("snf_hashgrid_bwd_presorted");
    return SNF_OK;
}

extern "C" int snf_hashgrid_bwd_presorted_adam_sp(const float* grad_out, int N, int L, int F, int log2_T, int ld_out, int col_off,
                                                  int n_run_levels, float* grad_table, const void* sorted_workspace, float* stage,
                                                  int fuse_from_level, float* param, float* exp_avg, float* exp_avg_sq, float lr,
                                                  float beta1, float beta2, float eps, int step, float grad_scale,
                                                  const uint32_t* reach_rows, const uint32_t* reach_start, int sparse_levels,
                                                  int sparse_max_rows, int sparse_step, void* scratch, snf_stream_t stream) {
    const bool planar = ld_out == 0;  // grad_out is already level-major [L][N][F]: it IS the staged gradient
    // (fuse_from_level = L and sparse_step = 0: nothing is stepped -- plain gradient accumulation, param & co. may be NULL)
    const bool step_sparse = sparse_levels > 0 && sparse_step != 0;
    const bool adam_on = fuse_from_level < L || step_sparse;
    SNF_REQUIRE(grad_out && grad_table && sorted_workspace && (stage || planar) && (!adam_on || (param && exp_avg && exp_avg_sq)),
                "snf_hashgrid_bwd_presorted_adam: null pointer");
    SNF_REQUIRE(F == 2 || F == 8, "snf_hashgrid_bwd_presorted_adam: features_per_level must be 2 or 8 (got %d)", F);
    SNF_REQUIRE(N > 0 && L > 0 && N <= (1 << HG_SAMPLE_BITS) && ((planar && col_off == 0) || ld_out >= col_off + L * F) &&
                    col_off >= 0,
                "snf_hashgrid_bwd_presorted_adam: bad shape N=%d L=%d ld_out=%d col_off=%d", N, L, ld_out, col_off);
    if (planar) stage = const_cast<float*>(grad_out);
    SNF_REQUIRE(fuse_from_level >= 0 && fuse_from_level <= L && step >= 1,
                "snf_hashgrid_bwd_presorted_adam: bad fuse_from_level=%d (L=%d) or step=%d", fuse_from_level, L, step);
    SNF_REQUIRE(sparse_levels >= 0 && sparse_levels <= fuse_from_level && sparse_levels <= (int)HG_FX_SCRATCH &&
                    (sparse_levels == 0 || (reach_rows && reach_start && sparse_max_rows <= snf_hashgrid_sparse_max_rows(F))),
                "snf_hashgrid_bwd_presorted_adam: reachable-row levels %d must be <= fuse_from_level %d, come with their row lists "
                "and hold at most %d rows per bucket (got %d)", sparse_levels, fuse_from_level, snf_hashgrid_sparse_max_rows(F),
                sparse_max_rows);
    SNF_REQUIRE((((uintptr_t)grad_out | (uintptr_t)grad_table | (uintptr_t)stage | (uintptr_t)param | (uintptr_t)exp_avg |
                  (uintptr_t)exp_avg_sq | (uintptr_t)scratch) & 15) == 0, "snf_hashgrid_bwd_presorted_adam: unaligned pointer");
    const HgGeom g = hg_geometry(N, log2_T);
    const HgWs w = hg_ws_layout(const_cast<void*>(sorted_workspace), N, L, g);
    uint32_t* lvlmax = scratch ? (uint32_t*)scratch : w.fx;
    SNF_REQUIRE(scratch || F == 2 || sparse_levels == 0,
                "snf_hashgrid_bwd_presorted_adam: F = 8 reachable-row levels need a scratch buffer (the sort may be shared)");
    SNF_REQUIRE(!hg_fx_on(F, L, N) || L <= (int)HG_FX_SCRATCH, "snf_hashgrid_bwd_presorted_adam: too many levels");
    hipStream_t st = (hipStream_t)stream;
    HgAdam a{};
    hg_fill_adam(a, param, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, grad_scale, fuse_from_level);
    if (!planar) hg_stage(st, F, grad_out, N, L, ld_out, col_off, stage);
    hg_reduce_one(st, F, stage, N, L, log2_T, g, w, grad_table, n_run_levels, fuse_from_level < L, step_sparse, a,
                  HgSparse{reach_rows, reach_start, sparse_levels}, lvlmax);
    SNF_LAUNCH_CHECK("snf_hashgrid_bwd_presorted_adam");
    return SNF_OK;
}

extern "C" int snf_hashgrid_bwd_presorted_adam(const float* grad_out, int N, int L, int F, int log2_T, int ld_out, int col_off,
                                               int n_run_levels, float* grad_table, const void* sorted_workspace, float* stage,
                                               int fuse_from_level, float* param, float* exp_avg, float* exp_avg_sq, float lr,
                                               float beta1, float beta2, float eps, int step, float grad_scale,
                                               snf_stream_t stream) {
    SNF_REQUIRE(param && exp_avg && exp_avg_sq, "snf_hashgrid_bwd_presorted_adam: null pointer");
    return snf_hashgrid_bwd_presorted_adam_sp(grad_out, N, L, F, log2_T, ld_out, col_off, n_run_levels, grad_table, sorted_workspace,
                                              stage, fuse_from_level, param, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step,
                                              grad_scale, nullptr, nullptr, 0, 0, 0, nullptr, stream);
}

// snf_hashgrid_bwd_presorted_adam for an F = 2 grid whose workspace was sorted by snf_hashgrid_sort_xp (x-pair records); param NULL:
// no optimizer step (fuse_from_level is then ignored), the gradient is added to grad_table.
extern "C" int snf_hashgrid_bwd_presorted_adam_xp(const float* grad_out, int N, int L, int log2_T, int ld_out, int col_off,
                                                  int n_run_levels, float* grad_table, const void* sorted_workspace, float* stage,
                                                  int fuse_from_level, float* param, float* exp_avg, float* exp_avg_sq, float lr,
                                                  float beta1, float beta2, float eps, int step, float grad_scale,
                                                  snf_stream_t stream) {
    const bool planar = ld_out == 0;
    if (!param) fuse_from_level = L;
    const bool adam_on = fuse_from_level < L;
    SNF_REQUIRE(grad_out && grad_table && sorted_workspace && (stage || planar) && (!adam_on || (param && exp_avg && exp_avg_sq)),
                "snf_hashgrid_bwd_presorted_adam_xp: null pointer");
    SNF_REQUIRE(N > 0 && L > 0 && N <= (1 << HG_SAMPLE_BITS) && (N % 2) == 0 && L <= (int)HG_FX_SCRATCH &&
                    ((planar && col_off == 0) || ld_out >= col_off + L * 2) && col_off >= 0,
                "snf_hashgrid_bwd_presorted_adam_xp: bad shape N=%d (even) L=%d (<= 64) ld_out=%d col_off=%d", N, L, ld_out, col_off);
    SNF_REQUIRE(fuse_from_level >= 0 && fuse_from_level <= L && step >= 1,
                "snf_hashgrid_bwd_presorted_adam_xp: bad fuse_from_level=%d (L=%d) or step=%d", fuse_from_level, L, step);
    if (planar) stage = const_cast<float*>(grad_out);
    SNF_REQUIRE((((uintptr_t)grad_out | (uintptr_t)grad_table | (uintptr_t)stage | (uintptr_t)param | (uintptr_t)exp_avg |
                  (uintptr_t)exp_avg_sq) & 15) == 0, "snf_hashgrid_bwd_presorted_adam_xp: unaligned pointer");
    SNF_REQUIRE(hg_fx_on(2, L, N), "snf_hashgrid_bwd_presorted_adam_xp: the fixed-point reduce does not take N=%d L=%d", N, L);
    const HgGeom g = hg_geometry(N, log2_T);
    const HgWs w = hg_ws_layout(const_cast<void*>(sorted_workspace), N, L, g);
    hipStream_t st = (hipStream_t)stream;
    HgAdam a{};
    hg_fill_adam(a, param, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, grad_scale, fuse_from_level);
    if (!planar) hg_stage(st, 2, grad_out, N, L, ld_out, col_off, stage);
    hg_reduce_one(st, 2, stage, N, L, log2_T, g, w, grad_table, n_run_levels, adam_on, false, a, HgSparse{nullptr, nullptr, 0}, w.fx, true);
    SNF_LAUNCH_CHECK("snf_hashgrid_bwd_presorted_adam_xp");
    return SNF_OK;
}

// The two F = 8 grids of a feature head (same samples, same table size, level-major gradients): the bucket-wide levels of both in
// ONE float-reduce launch (one tail instead of two), their reachable-row levels in k_hg_reduce_sparse launches in front of it.
extern "C" int snf_hashgrid_bwd_presorted_adam_pair(const float* grad_out0, const float* grad_out1, int N, int L0, int L1, int log2_T,
                                                    float* grad_table0, float* grad_table1, const void* sorted_workspace0,
                                                    const void* sorted_workspace1, int fuse_from_level0, int fuse_from_level1,
                                                    float* param0, float* exp_avg0, float* exp_avg_sq0, float* param1,
                                                    float* exp_avg1, float* exp_avg_sq1, const uint32_t* reach_rows0,
                                                    const uint32_t* reach_start0, int sparse_levels0, const uint32_t* reach_rows1,
                                                    const uint32_t* reach_start1, int sparse_levels1, int sparse_max_rows,
                                                    int sparse_step, void* scratch, float lr, float beta1, float beta2, float eps,
                                                    int step, float grad_scale, snf_stream_t stream) {
    SNF_REQUIRE(grad_out0 && grad_out1 && grad_table0 && grad_table1 && sorted_workspace0 && sorted_workspace1 && param0 && param1 &&
                    exp_avg0 && exp_avg1 && exp_avg_sq0 && exp_avg_sq1, "snf_hashgrid_bwd_presorted_adam_pair: null pointer");
    SNF_REQUIRE(N > 0 && L0 > 0 && L1 > 0 && L0 + L1 <= 65535 && N <= (1 << HG_SAMPLE_BITS) && fuse_from_level0 >= 0 &&
                    fuse_from_level0 <= L0 && fuse_from_level1 >= 0 && fuse_from_level1 <= L1 && step >= 1,
                "snf_hashgrid_bwd_presorted_adam_pair: bad shape N=%d L=%d+%d or fuse levels / step", N, L0, L1);
    SNF_REQUIRE(sparse_levels0 >= 0 && sparse_levels0 <= fuse_from_level0 && sparse_levels1 >= 0 && sparse_levels1 <= fuse_from_level1 &&
                    sparse_levels0 + sparse_levels1 <= (int)HG_FX_SCRATCH &&
                    (sparse_levels0 == 0 || (reach_rows0 && reach_start0)) && (sparse_levels1 == 0 || (reach_rows1 && reach_start1)) &&
                    (sparse_levels0 + sparse_levels1 == 0 || (scratch && sparse_max_rows <= hg_sp_max_rows<8>())),
                "snf_hashgrid_bwd_presorted_adam_pair: reachable-row levels %d / %d need their row lists, a scratch buffer and at most "
                "%d rows per bucket (got %d)", sparse_levels0, sparse_levels1, hg_sp_max_rows<8>(), sparse_max_rows);
    SNF_REQUIRE((((uintptr_t)grad_out0 | (uintptr_t)grad_out1 | (uintptr_t)grad_table0 | (uintptr_t)grad_table1 | (uintptr_t)param0 |
                  (uintptr_t)param1 | (uintptr_t)exp_avg0 | (uintptr_t)exp_avg1 | (uintptr_t)exp_avg_sq0 | (uintptr_t)exp_avg_sq1 |
                  (uintptr_t)scratch) & 15) == 0, "snf_hashgrid_bwd_presorted_adam_pair: unaligned pointer");
    const HgGeom g = hg_geometry(N, log2_T);
    const HgWs w0 = hg_ws_layout(const_cast<void*>(sorted_workspace0), N, L0, g);
    const HgWs w1 = hg_ws_layout(const_cast<void*>(sorted_workspace1), N, L1, g);
    const int B = 1 << g.log2B;
    hipStream_t st = (hipStream_t)stream;
    static const uint32_t hg_long = (uint32_t)HG_LONG;
    const bool on0 = sparse_step != 0, on1 = sparse_step != 0;  // the reachable-row levels are stepped (else: gradients only)
    HgAdam a{};
    hg_fill_adam(a, param0, exp_avg0, exp_avg_sq0, lr, beta1, beta2, eps, step, grad_scale, fuse_from_level0);
    HgSecond s2{};
    s2.gT = grad_out1; s2.bucket_start = w1.bstart; s2.records = (const uint2*)w1.records; s2.grad_table = grad_table1;
    hg_fill_adam(s2.adam, param1, exp_avg1, exp_avg_sq1, lr, beta1, beta2, eps, step, grad_scale, fuse_from_level1);
    // ---- reachable-row levels: fixed-point over compact rows, as workgroups of the SAME launch (their largest |g| per level first)
    uint32_t* lvlmax = (uint32_t*)scratch;  // (unused by the compact reduce since it scales per bucket; kept in the signature)
    const HgSparseDev sd0{reach_rows0, reach_start0, lvlmax, sparse_levels0, on0 ? 1 : 0};
    s2.sp = HgSparseDev{reach_rows1, reach_start1, lvlmax + sparse_levels0, sparse_levels1, on1 ? 1 : 0};
    static const int interleave = 1;
    s2.first_levels = L0;
    s2.interleave = (interleave && L0 == L1) ? 1 : 0;
    s2.level0 = 0;
    // (ADAM instantiation whenever anything is stepped: bucket-wide levels >= fuse_from_level, or reachable-row levels with sparse_step)
    const bool any_step = fuse_from_level0 < L0 || fuse_from_level1 < L1 || (sparse_step && sparse_levels0 + sparse_levels1 > 0);
    if (!any_step)
        hipLaunchKernelGGL((k_hg_reduce<8, false>), dim3(B, L0 + L1), dim3(HG_RT), 0, st, grad_out0, N, log2_T, g.log2B, w0.bstart,
                           (const uint2*)w0.records, grad_table0, hg_long, 0, 0, a, sd0, s2);
    else
        hipLaunchKernelGGL((k_hg_reduce<8, true>), dim3(B, L0 + L1), dim3(HG_RT), 0, st, grad_out0, N, log2_T, g.log2B, w0.bstart,
                           (const uint2*)w0.records, grad_table0, hg_long, 0, 0, a, sd0, s2);
    SNF_LAUNCH_CHECK("snf_hashgrid_bwd_presorted_adam_pair");
    return SNF_OK;
}

extern "C" int snf_hashgrid_bwd_sorted_ex(const float* u, const float* grad_out, const float* scalings, int N, int L, int F,
                                          int log2_T, int ld_out, int col_off, int n_run_levels, float* grad_table,
                                          void* workspace, int64_t workspace_bytes, snf_stream_t stream) {
    int rc = check_common("snf_hashgrid_bwd_sorted", u, grad_out, scalings, grad_table, N, L, F, log2_T, ld_out, col_off);
    if (rc) return rc;
    const HgGeom g = hg_geometry(N, log2_T);
    if ((1 << g.log2rpb) > HG_MAX_RPB) {
        // more than 2048 rows per bucket even at 4096 buckets (log2_T > 23): fall back to the atomic kernel
        return snf_hashgrid_bwd(u, grad_out, scalings, N, L, F, log2_T, ld_out, col_off, grad_table, stream);
    }
    rc = snf_hashgrid_sort(u, scalings, N, L, log2_T, workspace, workspace_bytes, stream);
    if (rc) return rc;
    const HgWs w = hg_ws_layout(workspace, N, L, g);  // the staged gradients use the tail of the same workspace
    return snf_hashgrid_bwd_presorted(grad_out, N, L, F, log2_T, ld_out, col_off, n_run_levels, grad_table, workspace, w.gT,
                                      stream);
}

// The fixed-point reduce with a caller-provided scratch area (>= SNF_HG_FX_SCRATCH_BYTES, private to this launch), for F = 2
// and F = 8 (every level bucket-wide).  The sorted workspace stays read-only here, so heads that SHARE a sort (same positions,
// same level geometry) can run their backward passes concurrently on different streams.  fuse_from_level = L: no level is stepped
// (plain gradient accumulation into grad_table).
extern "C" int snf_hashgrid_bwd_presorted_adam_fx(const float* grad_out, int N, int L, int F, int log2_T, int ld_out, int col_off,
                                                  int n_run_levels, float* grad_table, const void* sorted_workspace,
                                                  float* stage, int fuse_from_level, float* param, float* exp_avg,
                                                  float* exp_avg_sq, float lr, float beta1, float beta2, float eps, int step,
                                                  float grad_scale, void* scratch, snf_stream_t stream) {
    const bool planar = ld_out == 0;
    const bool adam_on = fuse_from_level < L;
    SNF_REQUIRE(grad_out && grad_table && sorted_workspace && (stage || planar) && scratch &&
                    (!adam_on || (param && exp_avg && exp_avg_sq)), "snf_hashgrid_bwd_presorted_adam_fx: null pointer");
    SNF_REQUIRE(F == 2 || F == 8, "snf_hashgrid_bwd_presorted_adam_fx: features_per_level must be 2 or 8 (got %d)", F);
    SNF_REQUIRE(N > 0 && L > 0 && L <= (int)HG_FX_SCRATCH && (N % 2) == 0 && N <= (1 << HG_SAMPLE_BITS) &&
                    ((planar && col_off == 0) || ld_out >= col_off + L * F) && col_off >= 0,
                "snf_hashgrid_bwd_presorted_adam_fx: bad shape N=%d L=%d ld_out=%d col_off=%d (N even, L <= 64)", N, L, ld_out, col_off);
    SNF_REQUIRE(fuse_from_level >= 0 && fuse_from_level <= L && step >= 1,
                "snf_hashgrid_bwd_presorted_adam_fx: bad fuse_from_level=%d (L=%d) or step=%d", fuse_from_level, L, step);
    SNF_REQUIRE((((uintptr_t)grad_out | (uintptr_t)grad_table | (uintptr_t)stage | (uintptr_t)param | (uintptr_t)exp_avg |
                  (uintptr_t)exp_avg_sq | (uintptr_t)scratch) & 15) == 0, "snf_hashgrid_bwd_presorted_adam_fx: unaligned pointer");
    if (planar) stage = const_cast<float*>(grad_out);
    const HgGeom g = hg_geometry(N, log2_T);
    const HgWs w = hg_ws_layout(const_cast<void*>(sorted_workspace), N, L, g);
    const int B = 1 << g.log2B;
    hipStream_t st = (hipStream_t)stream;
    HgAdam a{};
    hg_fill_adam(a, param, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, grad_scale, fuse_from_level);
    uint32_t* lvlmax = (uint32_t*)scratch;
    (void)hipMemsetAsync(lvlmax, 0, L * sizeof(uint32_t), st);
    if (!planar) hg_stage(st, F, grad_out, N, L, ld_out, col_off, stage);
    if (F == 2) {
        hipLaunchKernelGGL(k_hg_level_absmax<2>, dim3(hg_absmax_blocks(N, F), L), dim3(256), 0, st, stage, N, lvlmax);
        hipLaunchKernelGGL((k_hg_reduce_fx<2, true, 1>), dim3(B, L), dim3(HG_FX_T), 0, st, stage, N, log2_T, g.log2B, w.bstart,
                           (const uint2*)w.records, grad_table, hg_xcd_from(n_run_levels, L, B), hg_merge_levels(n_run_levels), 0, lvlmax, a);
    } else {
        hipLaunchKernelGGL(k_hg_level_absmax<8>, dim3(hg_absmax_blocks(N, F), L), dim3(256), 0, st, stage, N, lvlmax);
        hipLaunchKernelGGL((k_hg_reduce_fx<8, true, 2>), dim3(B * 2, L), dim3(HG_FX_T), 0, st, stage, N, log2_T, g.log2B,
                           w.bstart, (const uint2*)w.records, grad_table, L, 0, 0, lvlmax, a);
    }
    SNF_LAUNCH_CHECK("snf_hashgrid_bwd_presorted_adam_fx");
    return SNF_OK;
}

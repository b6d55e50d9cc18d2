// linear_b3.hip -- the 192/256-wide feature-head layers (tcnn CutlassMLP role: samnerf/sam_field.py:51-61,84-94) on the
// bf16 matrix cores with a 3-term split, fp32 accumulate (gfx950).
//
// fp32 operands are split on the fly into bf16 hi + bf16 lo (x = hi + lo + O(2^-17 |x|)); a product is accumulated as
// hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 (the lo*lo term is below fp32 round-off of the sum).  Measured
// against an fp64 reference on head-shaped data (K = 192..256): max abs error 8e-7 at |y| ~ 0.15, i.e. 10x the error of
// the exact-fp32 MFMA path and 100x inside the 1e-4 parity bar -- at 1/5 of the matrix-core cycles
// (3 x 32 cycles per 16 k against 8 x 64 cycles).  snf_set_gemm_mode(0) switches the library back to exact fp32.
//
// Tile: 128 rows x 64 cols x 32 k per workgroup (4 waves x 32 rows x two 32x32 accumulators), operands in LDS as
// row-major bf16 planes [row][k] with an 80-byte pitch: one conflict-free ds_read_b128 per operand fragment.
#include "common.hpp"
#include <type_traits>
#include <stdlib.h>

namespace snf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int B3_BM = 128, B3_BN = 64, B3_BK = 32;
constexpr int B3_PITCH = 40;  // bf16 elements per LDS row (80 B: 16-lane groups of a ds_read_b128 hit 16 distinct slots)

__device__ __forceinline__ float b3_act_deriv(float y, int act) {
    if (act == SNF_ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (act == SNF_ACT_SIGMOID) return y * (1.f - y);
    return 1.f;
}

__device__ __forceinline__ float b3_act_apply(float x, int act) {
    if (act == SNF_ACT_RELU) return fmaxf(x, 0.f);
    if (act == SNF_ACT_SIGMOID) return 1.f / (1.f + expf(-x));
    if (act == SNF_ACT_GELU) return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));  // nn.GELU() (erf form)
    return x;
}

// fp32 -> bf16 hi + bf16 lo with the packed hardware conversion (v_cvt_pk_bf16_f32: round-to-nearest-even, two values per
// instruction).  The integer-arithmetic rounding this replaces made the kernel VALU-bound: ~12 VALU operations per element
// against ~4 now, with every A element split once per column tile.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// two fp32 -> packed hi pair and packed lo pair (lo = bf16(x - hi)): 6 VALU instructions.  +-inf gives lo = NaN (inf - inf),
// i.e. NaN where fp32 arithmetic gives +-inf or NaN; flushing that NaN cost 4 more instructions per pair in kernels whose
// staging is VALU-bound (compile with -DSNF_B3_FLUSH_NAN to restore it).  NaN inputs propagate as NaN either way.
__device__ __forceinline__ void split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    hi = cvt_pk_bf16(x0, x1);
    lo = cvt_pk_bf16(x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xFFFF0000u));
}

__device__ __forceinline__ void split_bf16(float x, uint32_t& hi, uint32_t& lo) {
    uint32_t h, l;
    split2(x, 0.f, h, l);
    hi = h & 0xFFFFu;
    lo = l & 0xFFFFu;
}

// four consecutive fp32 -> 8 bytes of hi bf16 and 8 bytes of lo bf16
__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& lo) {
    uint32_t h0, h1, l0, l1;
    split2(v.x, v.y, h0, l0);
    split2(v.z, v.w, h1, l1);
    hi = make_uint2(h0, h1);
    lo = make_uint2(l0, l1);
}

// rscale != NULL: row m of A is rscale[m] * A[m / rgroup][:] (see k_gemm_ws_b3)
template <bool DERIV>
__device__ __forceinline__ float4 b3_load_a(const float* __restrict__ A, const float* __restrict__ Aux, int row, int k, int M,
                                            int K, int lda, int ldaux, int act, const float* __restrict__ rscale = nullptr,
                                            int rgroup = 1, int aux_bits = 0) {
    const bool ok = (row < M) && (k < K);
    const int rc = row < M ? row : M - 1;
    const int kc = k < K ? k : 0;
    float4 v;
    if (rscale != nullptr) {
        v = *reinterpret_cast<const float4*>(A + (size_t)(rc / rgroup) * lda + kc);
        const float q = rscale[rc];
        v.x *= q; v.y *= q; v.z *= q; v.w *= q;
    } else {
        v = *reinterpret_cast<const float4*>(A + (size_t)rc * lda + kc);
    }
    if constexpr (DERIV) {
        if (act != SNF_ACT_NONE && aux_bits) {  // Aux: ReLU mask bits [M][ldaux bytes]
            const uint32_t m = reinterpret_cast<const uint8_t*>(Aux)[(size_t)rc * ldaux + (kc >> 3)] >> (kc & 4);
            v.x = (m & 1u) ? v.x : 0.f; v.y = (m & 2u) ? v.y : 0.f; v.z = (m & 4u) ? v.z : 0.f; v.w = (m & 8u) ? v.w : 0.f;
        } else if (act != SNF_ACT_NONE) {
            const float4 y = *reinterpret_cast<const float4*>(Aux + (size_t)rc * ldaux + kc);
            v.x *= b3_act_deriv(y.x, act);
            v.y *= b3_act_deriv(y.y, act);
            v.z *= b3_act_deriv(y.z, act);
            v.w *= b3_act_deriv(y.w, act);
        }
    }
    if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
    return v;
}

__device__ __forceinline__ float4 b3_load_b(const float* __restrict__ B, int r, int c, int Rn, int Cn, int ldb) {
    const bool ok = (r < Rn) && (c < Cn);
    const int rc = r < Rn ? r : Rn - 1;
    const int cc = c < Cn ? c : 0;
    float4 v = *reinterpret_cast<const float4*>(B + (size_t)rc * ldb + cc);
    if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
    return v;
}

// the same from a level-major matrix [Cn/8][rows_total][8] (the feature grids' planar output, ld = -8): column quad c..c+3 of
// row r sits at ((c >> 3) * rows_total + r) * 8 + (c & 7)
__device__ __forceinline__ float4 b3_load_b_planar8(const float* __restrict__ B, int r, int c, int Rn, int Cn, int rows_total) {
    const bool ok = (r < Rn) && (c < Cn);
    const int rc = r < Rn ? r : Rn - 1;
    const int cc = c < Cn ? c : 0;
    float4 v = *reinterpret_cast<const float4*>(B + ((size_t)(cc >> 3) * rows_total + rc) * 8 + (cc & 7));
    if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
    return v;
}

// C[M,Nc] = op(A)[M,K] * B   (BT: B = W[Nc,K] row-major, i.e. forward;  !BT: B = W[K,Nc] row-major, i.e. data gradient)
// Requirements (checked by the host wrapper): K % 4 == 0, Nc % 4 == 0, leading dimensions % 4 == 0, 16-byte aligned bases.
// BN = columns per workgroup (64 / 128 / 192 / 256): a wave owns 32 rows x BN columns = BN/32 accumulators.  Wider tiles
// split every A element fewer times and run more MFMAs per staged k-tile; measured on the head shapes 128 is the best
// (+10 % over 64), 192 / 256 lose it again to register pressure and two-workgroup occupancy (SNF_B3_BN overrides the cap).
#define SNF_B3_ROWS_WAVES 1
template <bool BT, bool DERIV, int BN>
__global__ __launch_bounds__(256, SNF_B3_ROWS_WAVES) void k_gemm_rows_b3(const float* __restrict__ A, const float* __restrict__ Aux,
                                                      const float* __restrict__ B, const float* __restrict__ bias, int M,
                                                      int K, int Nc, int lda, int ldaux, int ldb, int ldc, int act_in,
                                                      int act_out, float* __restrict__ C, int ksplit = 0,
                                                      long long c_split_stride = 0) {
    constexpr int NB = BN / 32;  // accumulators per wave; also B-staging passes per thread
    if constexpr (BT && !DERIV) {
        if (ksplit > 0) {  // split-K slice z (see k_gemm_rows)
            const int kb = blockIdx.z * ksplit;
            A += kb; B += kb; K = min(ksplit, K - kb);
            C += (size_t)blockIdx.z * c_split_stride;
        }
    }
    __shared__ __attribute__((aligned(16))) uint16_t Ah[B3_BM * B3_PITCH];
    __shared__ __attribute__((aligned(16))) uint16_t Al[B3_BM * B3_PITCH];
    __shared__ __attribute__((aligned(16))) uint16_t Bh[BN * B3_PITCH];
    __shared__ __attribute__((aligned(16))) uint16_t Bl[BN * B3_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, half = lane >> 5;
    // Workgroups are dispatched round-robin over the 8 XCDs in linear order (x fastest).  Where the output has few column tiles
    // (at most half as many as row tiles: the encoder's 5120 -> 1280 layer, 32 x 10), XCD x takes the row tiles x, x + 8, ... and
    // walks ALL column tiles of a row tile back to back: a row tile of A (128 x K fp32, 2.6 MB at K = 5120) is then read from HBM
    // once by its XCD's L2 instead of once per column tile (that layer 0.317 -> 0.297 ms; with many column tiles the default order,
    // which shares a WEIGHT tile inside an XCD, is the better one: 1280 -> 5120 0.214 -> 0.231 ms swizzled).  SNF_B3_SWIZZLE=0: off.
    int bx = blockIdx.x, by = blockIdx.y;
#if !defined(SNF_B3_SWIZZLE) || SNF_B3_SWIZZLE
    if ((gridDim.x & 7) == 0 && gridDim.z == 1 && 2 * gridDim.y <= gridDim.x) {
        const int id = by * gridDim.x + bx, slot = id >> 3;
        const int r_local = slot / gridDim.y;
        by = slot - r_local * gridDim.y;
        bx = r_local * 8 + (id & 7);
    }
#endif
    const int row0 = bx * B3_BM, col0 = by * BN;
    f32x16 acc[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const int a_kq = tid & 7, a_r = tid >> 3;  // 8 k-quads x 32 rows (A: 4 passes, B^T: NB passes)
    constexpr int BQ = BN / 4;                  // B as [K, Nc]: BQ col-quads x (256 / BQ) k rows per pass, NB passes
    // B as [K, Nc] is transposed while it is staged.  A thread takes the k PAIR b_kp (rows 2 b_kp, 2 b_kp + 1 of the 32-k tile) of column
    // quad b_jq2 (+ 16 per pass) and writes one packed (k, k+1) 32-bit word per column and plane: the 16 k-pairs of a quarter-wave are 16
    // consecutive LDS words and the wave's four column quads start 80 words apart (16 banks): conflict-free.  (Until round 6: one k row x
    // four columns per thread as 16-bit stores 80 B apart across the lanes -- four banks for 32 lanes, 85 % of the kernel's LDS cycles were
    // bank conflicts, profiles/r06_step_counters.txt.)
    static_assert(BT || (NB % 2 == 0 && BQ % 16 == 0), "the transposing stage takes 16 column quads per pass");
    const int b_kp = tid & 15, b_jq2 = tid >> 4;
    float4 av[4], bv[NB];
    // `fetch` only issues loads at clamped (row, k): out-of-range quads are zeroed when the tile is STAGED, a trip later.  (The loaders
    // zero right behind the load -- a select that needs the data: the wait for the NEXT tile's operands then stood in front of THIS
    // tile's MFMAs and the one-tile prefetch hid nothing.  The data gradient's A operand keeps its loader: derivative modes per launch.)
    auto ldq = [&](const float* __restrict__ Mx, int r, int c, int Rn, int Cn, int ld) {
        return *reinterpret_cast<const float4*>(Mx + (size_t)(r < Rn ? r : Rn - 1) * ld + (c < Cn ? c : 0));
    };
    auto fetch = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if constexpr (DERIV) av[p] = b3_load_a<DERIV>(A, Aux, row0 + a_r + 32 * p, k0 + a_kq * 4, M, K, lda, ldaux, act_in);
            else av[p] = ldq(A, row0 + a_r + 32 * p, k0 + a_kq * 4, M, K, lda);
        }
        if constexpr (BT) {
#pragma unroll
            for (int p = 0; p < NB; ++p) bv[p] = ldq(B, col0 + a_r + 32 * p, k0 + a_kq * 4, Nc, K, ldb);
        } else {
            // (thread = k-pair b_kp x column quad b_jq2: two consecutive k rows of four columns per pass, see the staging below)
#pragma unroll
            for (int p = 0; p < NB / 2; ++p)
#pragma unroll
                for (int r = 0; r < 2; ++r) bv[2 * p + r] = ldq(B, k0 + 2 * b_kp + r, col0 + (b_jq2 + 16 * p) * 4, K, Nc, ldb);
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += B3_BK) {
        __syncthreads();
        {
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (!DERIV) {
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    if (!(row0 + a_r + 32 * p < M && k0 + a_kq * 4 < K)) av[p] = z4;
            }
#pragma unroll
            for (int p = 0; p < NB; ++p) {
                const bool ok = BT ? (col0 + a_r + 32 * p < Nc && k0 + a_kq * 4 < K)
                                   : (k0 + 2 * b_kp + (p & 1) < K && col0 + (b_jq2 + 16 * (p >> 1)) * 4 < Nc);
                if (!ok) bv[p] = z4;
            }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            uint2 hi, lo;
            split4(av[p], hi, lo);
            const int off = (a_r + 32 * p) * B3_PITCH + a_kq * 4;
            *reinterpret_cast<uint2*>(&Ah[off]) = hi;
            *reinterpret_cast<uint2*>(&Al[off]) = lo;
        }
        if constexpr (BT) {
#pragma unroll
            for (int p = 0; p < NB; ++p) {
                uint2 hi, lo;
                split4(bv[p], hi, lo);
                const int off = (a_r + 32 * p) * B3_PITCH + a_kq * 4;
                *reinterpret_cast<uint2*>(&Bh[off]) = hi;
                *reinterpret_cast<uint2*>(&Bl[off]) = lo;
            }
        } else {
            // transpose while staging: W[k][j..j+3] -> B planes [j][k]
#pragma unroll
            for (int p = 0; p < NB / 2; ++p) {
                const float e0[4] = {bv[2 * p].x, bv[2 * p].y, bv[2 * p].z, bv[2 * p].w};
                const float e1[4] = {bv[2 * p + 1].x, bv[2 * p + 1].y, bv[2 * p + 1].z, bv[2 * p + 1].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint32_t h0, l0, h1, l1;
                    split_bf16(e0[q], h0, l0);
                    split_bf16(e1[q], h1, l1);
                    const int off = ((b_jq2 + 16 * p) * 4 + q) * B3_PITCH + 2 * b_kp;
                    *reinterpret_cast<uint32_t*>(&Bh[off]) = (h0 & 0xFFFFu) | (h1 << 16);
                    *reinterpret_cast<uint32_t*>(&Bl[off]) = (l0 & 0xFFFFu) | (l1 << 16);
                }
            }
        }
        __syncthreads();
        if (!DERIV || k0 + B3_BK < K) fetch(k0 + B3_BK);  // (forward: unconditional -- clamped addresses, the surplus tile is never staged)
#pragma unroll
        for (int ks = 0; ks < B3_BK / 16; ++ks) {
            const int ko = ks * 16 + half * 8;
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&Ah[(wave * 32 + li) * B3_PITCH + ko]);
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(&Al[(wave * 32 + li) * B3_PITCH + ko]);
#pragma unroll
            for (int t = 0; t < NB; ++t) {
                if (col0 + 32 * t < Nc) {  // workgroup-uniform
                    const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&Bh[(32 * t + li) * B3_PITCH + ko]);
                    const bf16x8 bl = *reinterpret_cast<const bf16x8*>(&Bl[(32 * t + li) * B3_PITCH + ko]);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
                }
            }
        }
    }
    // epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int t = 0; t < NB; ++t) {
        const int c = col0 + 32 * t + li;
        if (c < Nc) {
            const float bb = bias ? bias[c] : 0.f;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = row0 + wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
                if (row < M) C[(size_t)row * ldc + c] = b3_act_apply(acc[t][reg] + bb, act_out);
            }
        }
    }
}

// Weight gradient on the bf16 matrix cores: dW[O,I] += sum_n dZ[n][o] X[n][i], dZ = dY * act'(Y).  The contraction runs over
// the ROWS, so both operands are staged transposed ([o][k = row] and [i][k = row] bf16 planes): a thread loads two consecutive
// rows of four columns and writes four packed (k, k+1) pairs per plane.  64 x 64 output tile, waves 2 x 2, 32 rows per trip;
// partial sums of a row chunk are added to dW with fp32 atomics (as the fp32 kernel does).
// AM: how act'(Y) arrives -- 0 none, 1 the activations Y; both with row-major X, and their `fetch` only ISSUES the loads (raw dY, Y, X
// quads of rows clamped into the chunk): the derivative, the bounds and the split happen when the tile is staged, a trip later.  -1: the
// general loaders (level-major X, any combination decided per launch) -- every load behind a launch-uniform branch with its first use
// right after it: 40 basic blocks and 33 waits per 32-row trip, the four to six loads of a trip going out one after the other IN FRONT
// of the trip's MFMAs.
template <int AM>
__global__ __launch_bounds__(256) void k_gemm_wgrad_b3(const float* __restrict__ dY, const float* __restrict__ Y,
                                                       const float* __restrict__ X, int N, int I, int O, int lddy, int ldy,
                                                       int ldx, int act, int rows_per_block, float* __restrict__ dW,
                                                       float* __restrict__ dbias, int to, int ti, int chunks, int xcd_order) {
    __shared__ __attribute__((aligned(16))) uint16_t Ah[64 * B3_PITCH];
    __shared__ __attribute__((aligned(16))) uint16_t Al[64 * B3_PITCH];
    __shared__ __attribute__((aligned(16))) uint16_t Bh[64 * B3_PITCH];
    __shared__ __attribute__((aligned(16))) uint16_t Bl[64 * B3_PITCH];
    __shared__ float bs[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, half = lane >> 5, wm = wave >> 1, wn = wave & 1;
    // Block order.  The to x ti output tiles of one row chunk read the same rows of dY / Y / X; workgroups go to the 8 XCDs
    // round-robin in linear order, so with the plain (tile, chunk) order the 12 tiles of a chunk land on 8 different L2s and
    // each pulls its own copy of the chunk (rocprofv3 FETCH_SIZE: 2.3 GB per step for the seven launches, 1.65x algorithmic).
    // xcd_order = 1 makes the tiles of a chunk consecutive in ONE XCD's queue (chunk = 8 * (seq / tiles) + xcd): measured
    // 1.05 GB per step -- but the kernel alone gets slower (192x256: 0.175 -> 0.25 ms; the twelve tiles now hit the same L2
    // lines at the same time) and the step time does not move, so the plain order stays the default (SNF_WGRAD_XCD=1).
    const int tiles = to * ti;
    int tile, chunk;
    if (xcd_order) {
        const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
        tile = seq % tiles;
        chunk = (seq / tiles) * 8 + xcd;
    } else {
        tile = blockIdx.x % tiles;
        chunk = blockIdx.x / tiles;
    }
    if (chunk >= chunks) return;
    const int o0 = (tile % to) * 64, i0 = (tile / to) * 64;
    const bool first_i_tile = tile / to == 0;
    const int n_begin = chunk * rows_per_block, n_end = min(N, n_begin + rows_per_block);
    const int q = tid & 15, kp = tid >> 4;  // 16 column quads x 16 row pairs
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    if (tid < 64) bs[tid] = 0.f;
    float4 av[2], bv[2], yv[AM == 1 ? 2 : 1];
    const int oq = o0 + q * 4, iq = i0 + q * 4;
    const bool o_in = oq < O, i_in = iq < I;
    const size_t a_col = o_in ? oq : 0, b_col = i_in ? iq : 0;
    auto fetch = [&](int n0) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int n = n0 + 2 * kp + p;
            if constexpr (AM < 0) {
                av[p] = b3_load_a<true>(dY, Y, n, oq, n_end, O, lddy, ldy, act);
                bv[p] = ldx < 0 ? b3_load_b_planar8(X, n, iq, n_end, I, N) : b3_load_b(X, n, iq, n_end, I, ldx);
            } else {
                const size_t nc = (size_t)min(n, n_end - 1);
                av[p] = *reinterpret_cast<const float4*>(dY + nc * lddy + a_col);
                if constexpr (AM == 1) yv[p] = *reinterpret_cast<const float4*>(Y + nc * ldy + a_col);
                bv[p] = *reinterpret_cast<const float4*>(X + nc * ldx + b_col);
            }
        }
    };
    fetch(n_begin);
    for (int n0 = n_begin; n0 < n_end; n0 += B3_BK) {
        __syncthreads();
        {
            if constexpr (AM >= 0) {  // the rows fetched for THIS trip: derivative and bounds now
                if constexpr (AM == 1) {
                    if (act == SNF_ACT_RELU) {  // (one launch-uniform branch around both rows, not one per element)
#pragma unroll
                        for (int p = 0; p < 2; ++p) {
                            av[p].x = yv[p].x > 0.f ? av[p].x : 0.f; av[p].y = yv[p].y > 0.f ? av[p].y : 0.f;
                            av[p].z = yv[p].z > 0.f ? av[p].z : 0.f; av[p].w = yv[p].w > 0.f ? av[p].w : 0.f;
                        }
                    } else {
#pragma unroll
                        for (int p = 0; p < 2; ++p) {
                            av[p].x *= b3_act_deriv(yv[p].x, act); av[p].y *= b3_act_deriv(yv[p].y, act);
                            av[p].z *= b3_act_deriv(yv[p].z, act); av[p].w *= b3_act_deriv(yv[p].w, act);
                        }
                    }
                }
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const bool row_in = n0 + 2 * kp + p < n_end;
                    if (!(row_in && o_in)) av[p] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (!(row_in && i_in)) bv[p] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            const float a0[4] = {av[0].x, av[0].y, av[0].z, av[0].w}, a1[4] = {av[1].x, av[1].y, av[1].z, av[1].w};
            const float b0[4] = {bv[0].x, bv[0].y, bv[0].z, bv[0].w}, b1[4] = {bv[1].x, bv[1].y, bv[1].z, bv[1].w};
            // The 8-k chunks of a plane row sit XOR-swizzled by (row >> 4) & 3: the 16 column quads of a store instruction are rows
            // 80 words apart (16 banks), i.e. FOUR banks for sixteen lanes in the plain layout (77 % of this kernel's LDS cycles were
            // bank conflicts, profiles/r06_step_counters.txt); with the swizzle the quads q, q + 4, q + 8, q + 12 land in different
            // chunks -- 64 lanes, 64 banks.  The fragment reads below apply the same XOR (the 16 lanes of a read group share row >> 4).
            const int wsw = ((kp >> 2) ^ ((q >> 2) & 3)) * 8 + 2 * (kp & 3);  // element offset of this thread's (k, k + 1) pair in its rows
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                uint32_t h, l;
                split2(a0[c], a1[c], h, l);  // rows (2kp, 2kp+1) of column o0 + 4q + c -> k positions (2kp, 2kp+1)
                *reinterpret_cast<uint32_t*>(&Ah[(q * 4 + c) * B3_PITCH + wsw]) = h;
                *reinterpret_cast<uint32_t*>(&Al[(q * 4 + c) * B3_PITCH + wsw]) = l;
                split2(b0[c], b1[c], h, l);
                *reinterpret_cast<uint32_t*>(&Bh[(q * 4 + c) * B3_PITCH + wsw]) = h;
                *reinterpret_cast<uint32_t*>(&Bl[(q * 4 + c) * B3_PITCH + wsw]) = l;
                bsum[c] += a0[c] + a1[c];
            }
        }
        __syncthreads();
        if (AM >= 0 || n0 + B3_BK < n_end) fetch(n0 + B3_BK);  // (AM >= 0: unconditional, rows clamped -- past the end the last row again)
#pragma unroll
        for (int ks = 0; ks < B3_BK / 16; ++ks) {
            const int ca = ((ks * 2 + half) ^ (((wm * 32 + li) >> 4) & 3)) * 8, cb = ((ks * 2 + half) ^ (((wn * 32 + li) >> 4) & 3)) * 8;
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&Ah[(wm * 32 + li) * B3_PITCH + ca]);
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(&Al[(wm * 32 + li) * B3_PITCH + ca]);
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&Bh[(wn * 32 + li) * B3_PITCH + cb]);
            const bf16x8 bl = *reinterpret_cast<const bf16x8*>(&Bl[(wn * 32 + li) * B3_PITCH + cb]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int o = o0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
        const int i = i0 + wn * 32 + li;
        if (o < O && i < I) unsafeAtomicAdd(&dW[(size_t)o * I + i], acc[reg]);
    }
    if (dbias != nullptr && first_i_tile) {
#pragma unroll
        for (int c = 0; c < 4; ++c) atomicAdd(&bs[q * 4 + c], bsum[c]);
        __syncthreads();
        if (tid < 64 && o0 + tid < O) unsafeAtomicAdd(&dbias[o0 + tid], bs[tid]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Full-width weight gradient for the feature-head layers (O, I <= 256, tens of thousands of rows).
// The tiled kernel above gives every 64 x 64 output tile its own pass over the rows: dY is read I/64 times and X O/64 times
// (rocprofv3 FETCH_SIZE, r02f: 2.2 GB per step for the seven launches against 1.0 GB of operands).  Here a workgroup owns a
// chunk of ROWS and the WHOLE O x I output: every operand element is read from HBM exactly once.  8 waves as 4 (o) x 2 (i),
// a wave accumulates 64 x 128 outputs (2 x 4 accumulator tiles = 128 registers) over its chunk; per 16-row trip both
// operands are staged transposed as bf16 hi / lo planes [column][k = row] (48-byte pitch: the b128 fragment reads of 16
// lanes hit 16 distinct bank quads), one k-step of 3 x 8 MFMAs per wave.  The chunk sums go to a partial buffer
// P[chunk][O][I] with plain coalesced stores and a second small kernel adds them to dW -- no float atomics on the
// O x I x chunks partial sums (deterministic up to the few-way split of that second pass).
constexpr int WF_PITCH = 24;  // bf16 per LDS row: 16 k + 8 pad
constexpr int WF_T = 512;
#ifndef SNF_WS_EVEN_TILES
#define SNF_WS_EVEN_TILES 1
#endif
#ifndef SNF_WS_MIN_ROWS
// fewer rows than this: the tiled kernel (a weight-stationary workgroup stages a whole weight slice first).  8192 since round 5: at 4096
// rows (the heads' last layers on the rendered rows) the weight-stationary launch is 32 workgroups of 131 KB LDS -- 0.033 / 0.035 ms alone
// where the tiled kernel's 128 light workgroups take 0.023 / 0.030, and inside the step each of them waits for an EMPTY CU
#define SNF_WS_MIN_ROWS 8192
#endif
#ifndef SNF_WF_CHUNKS
#define SNF_WF_CHUNKS 256  // row chunks (= workgroups, = partial sums P[chunk][O][I]) of the full-width weight gradient
#endif
#ifndef SNF_WGRAD_RAW_LOADS
#define SNF_WGRAD_RAW_LOADS 1  // 0: the tiled weight gradient through its general loaders (A/B)
#endif
#ifndef SNF_WF_BITS_KERNEL
#define SNF_WF_BITS_KERNEL 1  // 0: the general kernel also for the train step's case (A/B, tests)
#endif
#ifndef SNF_WF_RSPLIT
#define SNF_WF_RSPLIT 4    // k_wgrad_full_reduce: the chunks of one output element are summed by this many workgroups
#endif

__global__ __launch_bounds__(WF_T) void k_wgrad_full_b3(const float* __restrict__ dY, const float* __restrict__ Y,
                                                        const float* __restrict__ X, int N, int I, int O, int lddy, int ldy,
                                                        int ldx, int act, int rows_per_wg, float* __restrict__ P,
                                                        const float* __restrict__ rscale, int rgroup, int aux_bits) {
    // two copies of the four planes: a trip's MFMAs read one while the next trip's rows are split into the other -- one
    // workgroup barrier per 16 rows, and the staging (VALU + LDS stores) of some waves runs under the MFMAs of the others
    __shared__ __attribute__((aligned(16))) uint16_t Ah[2][256 * WF_PITCH];
    __shared__ __attribute__((aligned(16))) uint16_t Al[2][256 * WF_PITCH];
    __shared__ __attribute__((aligned(16))) uint16_t Bh[2][256 * WF_PITCH];
    __shared__ __attribute__((aligned(16))) uint16_t Bl[2][256 * WF_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, half = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int n_begin = blockIdx.x * rows_per_wg, n_end = min(N, n_begin + rows_per_wg);
    const int kp = tid & 7, q = tid >> 3;  // row pair within the 16-row trip, column quad (columns 4q .. 4q+3)
    const int col = q * 4;
    // the I columns go to the two wave columns in equal runs of 32-column tiles (192 = 96 + 96, not 128 + 64: the waves of a
    // SIMD share wn, so an uneven split leaves two SIMDs with twice the MFMAs of the other two)
    const int tiles_i = (I + 31) >> 5, nt = (tiles_i + 1) >> 1, i_base = 32 * nt * wn;
    f32x16 acc[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[u][t][i] = 0.f;
    // Rows are requested TWO trips ahead (two register sets, the loop unrolled by two): with one trip of lead a trip lasted as long as a
    // load takes under the step's traffic (55 us for 16 trips of ~0.6 us of matrix work each, round-4 counters: 60 % of the wave cycles
    // waiting) -- the kernel was a chain of memory latencies.
    float4 av[2][2], bv[2][2];
    auto fetch = [&](int n0, float4 (&a_)[2], float4 (&b_)[2]) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int n = n0 + 2 * kp + p;
            a_[p] = b3_load_a<true>(dY, Y, n, col, n_end, O, lddy, ldy, act, rscale, rgroup, aux_bits);
            b_[p] = ldx < 0 ? b3_load_b_planar8(X, n, col, n_end, I, N) : b3_load_b(X, n, col, n_end, I, ldx);
        }
    };
    auto stage = [&](int buf, const float4 (&a_)[2], const float4 (&b_)[2]) {
        const float a0[4] = {a_[0].x, a_[0].y, a_[0].z, a_[0].w}, a1[4] = {a_[1].x, a_[1].y, a_[1].z, a_[1].w};
        const float b0[4] = {b_[0].x, b_[0].y, b_[0].z, b_[0].w}, b1[4] = {b_[1].x, b_[1].y, b_[1].z, b_[1].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint32_t h, l;
            split2(a0[c], a1[c], h, l);  // rows (2kp, 2kp+1) of column col + c -> k positions (2kp, 2kp+1)
            *reinterpret_cast<uint32_t*>(&Ah[buf][(col + c) * WF_PITCH + 2 * kp]) = h;
            *reinterpret_cast<uint32_t*>(&Al[buf][(col + c) * WF_PITCH + 2 * kp]) = l;
            split2(b0[c], b1[c], h, l);
            *reinterpret_cast<uint32_t*>(&Bh[buf][(col + c) * WF_PITCH + 2 * kp]) = h;
            *reinterpret_cast<uint32_t*>(&Bl[buf][(col + c) * WF_PITCH + 2 * kp]) = l;
        }
    };
    fetch(n_begin, av[0], bv[0]);
    stage(0, av[0], bv[0]);
    __syncthreads();
    if (n_begin + 16 < n_end) fetch(n_begin + 16, av[0], bv[0]);
    if (n_begin + 32 < n_end) fetch(n_begin + 32, av[1], bv[1]);
    const bool m_on[2] = {64 * wm < O, 64 * wm + 32 < O};
    // one trip: fragments of copy `buf`, the NEXT trip's rows (register set `a_`, fetched two trips ago) split into the other copy,
    // the fetch for the trip after next into the freed set, this trip's MFMAs, barrier
    auto trip = [&](int n0, int buf, float4 (&a_)[2], float4 (&b_)[2]) {
        bf16x8 bh[4], bl[4], ah[2], al[2];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            bh[t] = *reinterpret_cast<const bf16x8*>(&Bh[buf][(i_base + 32 * t + li) * WF_PITCH + 8 * half]);
            bl[t] = *reinterpret_cast<const bf16x8*>(&Bl[buf][(i_base + 32 * t + li) * WF_PITCH + 8 * half]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            ah[u] = *reinterpret_cast<const bf16x8*>(&Ah[buf][(64 * wm + 32 * u + li) * WF_PITCH + 8 * half]);
            al[u] = *reinterpret_cast<const bf16x8*>(&Al[buf][(64 * wm + 32 * u + li) * WF_PITCH + 8 * half]);
        }
        if (n0 + 16 < n_end) {
            stage(buf ^ 1, a_, b_);
            if (n0 + 48 < n_end) fetch(n0 + 48, a_, b_);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (m_on[u]) {  // wave-uniform
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (t < nt && i_base + 32 * t < I) {  // wave-uniform
                        acc[u][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[u], bh[t], acc[u][t], 0, 0, 0);
                        acc[u][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[u], bl[t], acc[u][t], 0, 0, 0);
                        acc[u][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[u], bh[t], acc[u][t], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    };
    for (int n0 = n_begin; n0 < n_end; n0 += 32) {
        trip(n0, 0, av[0], bv[0]);
        if (n0 + 16 < n_end) trip(n0 + 16, 1, av[1], bv[1]);
    }
    float* __restrict__ Pw = P + (size_t)blockIdx.x * O * I;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int i = i_base + 32 * t + li;
            if (t < nt && i < I) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int o = 64 * wm + 32 * u + (reg & 3) + 8 * (reg >> 2) + 4 * half;
                    if (o < O) Pw[(size_t)o * I + i] = acc[u][t][reg];
                }
            }
        }
}

// The same kernel for the case the train step launches (the head's first layer: dY rows = rscale[n] * dYbar[n >> rshift][:] masked by
// ReLU bits, X level-major [I/8][N][8], whole 16-row trips, O and I multiples of 32), with the loader taken apart: `fetch` only ISSUES
// loads -- one dYbar row quad (rows 2kp and 2kp+1 share their group: rshift >= 1), the two rows' scales as one 8-byte load, two mask
// bytes, two X quads -- and `stage`, a trip later, scales, masks and splits.  In the general kernel above every load sits behind a
// launch-uniform branch (rscale? bits? level-major?) with its first use right behind it: 135 basic blocks and 84 waits per pair of trips,
// i.e. the six loads of a trip went out one after the other, each waiting for the previous one (a trip took ~8 000 cycles for ~800 of
// matrix work; requesting rows two trips ahead could not help).  Waves whose 32 columns lie past I skip the X half (wave-uniform).
__global__ __launch_bounds__(WF_T) void k_wgrad_full_b3_bits(const float* __restrict__ dYbar, const uint8_t* __restrict__ bits,
                                                             const float* __restrict__ X, int N, int I, int O, int lddy, int ldbits,
                                                             int rows_per_wg, float* __restrict__ P,
                                                             const float* __restrict__ rscale, int rshift) {
    __shared__ __attribute__((aligned(16))) uint16_t Ah[2][256 * WF_PITCH];
    __shared__ __attribute__((aligned(16))) uint16_t Al[2][256 * WF_PITCH];
    __shared__ __attribute__((aligned(16))) uint16_t Bh[2][256 * WF_PITCH];
    __shared__ __attribute__((aligned(16))) uint16_t Bl[2][256 * WF_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, half = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int n_begin = blockIdx.x * rows_per_wg, n_end = min(N, n_begin + rows_per_wg);  // (n_end - n_begin: a multiple of 16)
    const int kp = tid & 7, q = tid >> 3;
    const int col = q * 4;
    const bool a_on = col < O, b_on = col < I;  // wave-uniform (a wave covers 32 columns; O, I multiples of 32)
    const int tiles_i = I >> 5, nt = (tiles_i + 1) >> 1, i_base = 32 * nt * wn;
    f32x16 acc[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[u][t][i] = 0.f;
    // lane-constant parts of the addresses (32-bit element / byte offsets; the host checks the ranges)
    const uint32_t a_col = (uint32_t)(a_on ? col : 0);
    const uint32_t m_col = (uint32_t)((a_on ? col : 0) >> 3);
    const int m_shift = col & 4;
    const uint32_t x_off = (uint32_t)(((size_t)((b_on ? col : 0) >> 3) * N) * 8 + (col & 7));  // + 8 n
    struct Raw { float4 a; float2 s; uint32_t m0, m1; float4 b0, b1; };
    Raw raw[2];
    auto fetch = [&](int n0, Raw& r) {  // rows n0 + 2 kp, + 1 (inside [n_begin, n_end): whole trips)
        const uint32_t n = (uint32_t)(n0 + 2 * kp);
        r.a = *reinterpret_cast<const float4*>(dYbar + ((size_t)(n >> rshift) * lddy + a_col));
        r.s = *reinterpret_cast<const float2*>(rscale + n);
        r.m0 = bits[(size_t)n * ldbits + m_col];
        r.m1 = bits[(size_t)(n + 1) * ldbits + m_col];
        r.b0 = *reinterpret_cast<const float4*>(X + (x_off + 8u * n));
        r.b1 = *reinterpret_cast<const float4*>(X + (x_off + 8u * n + 8u));
    };
    auto stage = [&](int buf, const Raw& r) {
        if (a_on) {
            const float g[4] = {r.a.x, r.a.y, r.a.z, r.a.w};
            const uint32_t m0 = r.m0 >> m_shift, m1 = r.m1 >> m_shift;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float a0 = ((m0 >> c) & 1u) ? g[c] * r.s.x : 0.f;  // (the product, then the mask: the bits of the general loader)
                const float a1 = ((m1 >> c) & 1u) ? g[c] * r.s.y : 0.f;
                uint32_t h, l;
                split2(a0, a1, h, l);  // rows (2kp, 2kp+1) of column col + c -> k positions (2kp, 2kp+1)
                *reinterpret_cast<uint32_t*>(&Ah[buf][(col + c) * WF_PITCH + 2 * kp]) = h;
                *reinterpret_cast<uint32_t*>(&Al[buf][(col + c) * WF_PITCH + 2 * kp]) = l;
            }
        }
        if (b_on) {
            const float b0[4] = {r.b0.x, r.b0.y, r.b0.z, r.b0.w}, b1[4] = {r.b1.x, r.b1.y, r.b1.z, r.b1.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                uint32_t h, l;
                split2(b0[c], b1[c], h, l);
                *reinterpret_cast<uint32_t*>(&Bh[buf][(col + c) * WF_PITCH + 2 * kp]) = h;
                *reinterpret_cast<uint32_t*>(&Bl[buf][(col + c) * WF_PITCH + 2 * kp]) = l;
            }
        }
    };
    const int last = n_end - 16;  // first row of the last trip; fetches past it re-read it (unconditional loads, results unused)
    fetch(n_begin, raw[0]);
    fetch(min(n_begin + 16, last), raw[1]);
    stage(0, raw[0]);
    __syncthreads();
    fetch(min(n_begin + 32, last), raw[0]);
    const bool m_on[2] = {64 * wm < O, 64 * wm + 32 < O};
    // one trip: fragments of copy `buf`; the NEXT trip's rows (register set r, requested two trips ago) split into the other copy; the
    // request for the trip after that into the freed set; this trip's MFMAs; barrier
    auto trip = [&](int n0, int buf, Raw& r) {
        bf16x8 bh[4], bl[4], ah[2], al[2];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            bh[t] = *reinterpret_cast<const bf16x8*>(&Bh[buf][(i_base + 32 * t + li) * WF_PITCH + 8 * half]);
            bl[t] = *reinterpret_cast<const bf16x8*>(&Bl[buf][(i_base + 32 * t + li) * WF_PITCH + 8 * half]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            ah[u] = *reinterpret_cast<const bf16x8*>(&Ah[buf][(64 * wm + 32 * u + li) * WF_PITCH + 8 * half]);
            al[u] = *reinterpret_cast<const bf16x8*>(&Al[buf][(64 * wm + 32 * u + li) * WF_PITCH + 8 * half]);
        }
        stage(buf ^ 1, r);  // (past the last trip: the last trip's rows once more, into the copy nobody reads again)
        fetch(min(n0 + 48, last), r);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (m_on[u]) {  // wave-uniform
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (t < nt && i_base + 32 * t < I) {  // wave-uniform
                        acc[u][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[u], bh[t], acc[u][t], 0, 0, 0);
                        acc[u][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[u], bl[t], acc[u][t], 0, 0, 0);
                        acc[u][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[u], bh[t], acc[u][t], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    };
    for (int n0 = n_begin; n0 < n_end; n0 += 32) {
        trip(n0, 0, raw[1]);
        if (n0 + 16 < n_end) trip(n0 + 16, 1, raw[0]);
    }
    float* __restrict__ Pw = P + (size_t)blockIdx.x * O * I;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int i = i_base + 32 * t + li;
            if (t < nt && i < I) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int o = 64 * wm + 32 * u + (reg & 3) + 8 * (reg >> 2) + 4 * half;
                    if (o < O) Pw[(size_t)o * I + i] = acc[u][t][reg];
                }
            }
        }
}

// dW[e] += sum over the chunks [c0, c1) of this block's slice of P[chunk][e]  (blockIdx.y splits the chunks 4 ways)
__global__ __launch_bounds__(256) void k_wgrad_full_reduce(const float* __restrict__ P, int chunks, int OI,
                                                           float* __restrict__ dW) {
    const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= OI) return;
    const int c0 = (int)((long long)chunks * blockIdx.y / gridDim.y), c1 = (int)((long long)chunks * (blockIdx.y + 1) / gridDim.y);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int c = c0; c < c1; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(P + (size_t)c * OI + e);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    unsafeAtomicAdd(&dW[e], s.x);
    unsafeAtomicAdd(&dW[e + 1], s.y);
    unsafeAtomicAdd(&dW[e + 2], s.z);
    unsafeAtomicAdd(&dW[e + 3], s.w);
}

static int wf_rows_per_wg(int N) {
    static const int chunks = SNF_WF_CHUNKS;
    int rows = ceil_div(N, chunks);  // default: one workgroup per CU
    rows = ((rows + 15) / 16) * 16;
    if (rows < 64) rows = 64;
    return rows;
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight-stationary variant for the feature-head shapes (K <= 256, tens of thousands of rows): C[M,Nc] = op(A)[M,K] * B.
// A head layer is 3x-5x above its HBM floor in the tiled kernel above: every 128-row tile re-stages and re-splits the
// weights, pays two workgroup barriers per 32 k, and every A element is split once per column tile.  Here a workgroup
// keeps its BN-column slice of the weights in LDS for its whole life (bf16 hi/lo planes [n][k], split once), and the
// activations never touch LDS: a wave owns 32*RB rows, loads its A fragments straight from global memory in MFMA operand
// layout (lane (m, half) holds A[m][16s + 8*half .. +8], two dwordx4), splits them in registers and runs 3 MFMAs per
// (row block, column tile, k-step) against B fragments read with one ds_read_b128 per plane.  No barrier after the
// prologue; the loop over row tiles is persistent.  The split is the VALU cost that matters (a wave64 VALU instruction
// occupies its SIMD for 4 cycles): ~6 instructions per pair of elements here, once per BN columns.
//   BT  : B[k][n] = W[n][k]  (W [Nc,K] row-major: forward)         !BT : B[k][n] = W[k][n]  (W [K,Nc] row-major: data gradient)
// LDS: 2 planes x BN x (K + 8) bf16 (pitch K + 8 keeps the b128 fragment reads of 16 lanes on distinct banks for K = 192, 256).
// +-inf inputs give NaN here (lo = inf - inf), where fp32 arithmetic gives +-inf or NaN; NaN inputs propagate as NaN.
__device__ __forceinline__ void ws_split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    hi = cvt_pk_bf16(x0, x1);
    lo = cvt_pk_bf16(x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xFFFF0000u));
}

// PA: A is level-major [K/8][M][8] (lda = -8): the fragment of lane (m, half) for k-step s -- A[m][16s + 8*half .. +8] -- is level
//     2s + half of row m, 32 contiguous bytes, and the 32 lanes of a half-wave read 1 KB in one piece (row-major rows are
//     lda*4 bytes apart).
// CT: the product is accumulated transposed (C^T = W^T A^T: the weight fragment is the first MFMA operand), so a lane owns
//     ONE row m and, per accumulator register quad, four consecutive output features; C is written level-major
//     [Nc/8][M][8] (ldc = -8): lane (m, half) stores 16 bytes of level (col0 + 32t)/8 + q, and a wave store covers 32 rows x
//     32 bytes = 1 KB contiguous.  This is the staged-gradient layout of the hash-grid backward (snf_hashgrid_bwd_presorted
//     with ld_out = 0): the data gradient of a head's first layer lands where the table backward reads it.
// rscale != NULL ("grouped rows", data gradient only): row m of A is rscale[m] * A[m / rgroup][:] -- the gradient of a weighted
//     mean over rgroup consecutive rows (MeanRenderer over a ray's samples), never materialised: A holds one row per group.
// aux_bits (data gradient): Aux is the ReLU mask of the layer's output as BITS, [M][ldaux bytes], bit c of a row = (y[c] > 0).
// hbar != NULL (forward, ReLU, rgroup == 16): the epilogue renders the activations -- hbar[m / 16][c] = sum_k rscale[16 (m/16) + k]
//     * y[16 (m/16) + k][c] -- and writes the ReLU mask bits to ybits; C may then be NULL: the activations themselves are not
//     needed again (the layer after it runs on the rendered rows, the backward needs the mask only).
// AM (data gradient): how the derivative mask arrives, fixed at compile time -- 0 none, 1 the activations (fp32), 2 mask bits;
//     -1: decided per launch from act_in / aux_bits (loads under launch-uniform branches: the compiler then waits for ALL loads in
//     flight before every k-step, s_waitcnt vmcnt(0), and the D-deep prefetch hides nothing).  All loads of the k loop are
//     unconditional for that reason too: the refill past the last k-step re-reads the last one.
#ifndef SNF_WS_DEPTH
#define SNF_WS_DEPTH 4  // k-steps of raw A fragments in flight per wave (512-thread instances)
#endif
template <bool BT, bool DERIV, int BN, int RB, int THREADS, int DEPTH, bool PA = false, bool CT = false, int AM = -1>
__global__ __launch_bounds__(THREADS) void k_gemm_ws_b3(const float* __restrict__ A, const float* __restrict__ Aux,
                                                        const float* __restrict__ W, const float* __restrict__ bias, int M,
                                                        int K, int Nc, int lda, int ldaux, int ldw, int ldc, int act_in,
                                                        int act_out, float* __restrict__ C,
                                                        const float* __restrict__ rscale = nullptr, int rgroup = 1,
                                                        int aux_bits = 0, float* __restrict__ hbar = nullptr,
                                                        uint8_t* __restrict__ ybits = nullptr) {
    constexpr int NB = BN / 32;
    constexpr int TILE_ROWS = (THREADS / 64) * 32 * RB;
    extern __shared__ __attribute__((aligned(16))) uint16_t ws_lds[];
    const int pitch = K + 8;
    uint16_t* __restrict__ Bh = ws_lds;
    uint16_t* __restrict__ Bl = ws_lds + (size_t)BN * pitch;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, half = lane >> 5;
    const int col0 = blockIdx.y * BN;
    // ---- prologue: this workgroup's weight slice, split once; loads go out in batches of independent float4 per thread
    constexpr int PB = 8;
    if constexpr (BT) {
        const int kq = K >> 2;  // float4 per row
        const int total = BN * kq;
        for (int e0 = tid; e0 < total; e0 += THREADS * PB) {
            float4 v[PB];
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int e = e0 + THREADS * u;
                const int n = e / kq, k = (e - n * kq) * 4;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < total && col0 + n < Nc) v[u] = *reinterpret_cast<const float4*>(W + (size_t)(col0 + n) * ldw + k);
            }
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int e = e0 + THREADS * u;
                if (e < total) {
                    const int n = e / kq, k = (e - n * kq) * 4;
                    uint2 hi, lo;
                    split4(v[u], hi, lo);
                    *reinterpret_cast<uint2*>(&Bh[(size_t)n * pitch + k]) = hi;
                    *reinterpret_cast<uint2*>(&Bl[(size_t)n * pitch + k]) = lo;
                }
            }
        }
    } else {
        // an item = (column quad q, k pair): two float4 of W in, four (column, k pair) dwords per plane out.  LDS stores bank on
        // (dword address) mod 32 within 32-lane groups and a column is 132 (mod 32: 4) dwords from the next, so a group of lanes
        // takes 8 quads x 4 k pairs and lane (qq, kk) writes its four columns rotated by qq / 2: the banks 16 (q & 1) +
        // 4 ((c + q / 2) & 3) + kk are all distinct (quads fastest over the lanes: 16-way conflicts, ~7 us of a 50 us launch)
        constexpr int NQ = BN / 4;  // column quads
        const int kp = K >> 1;      // k pairs (K % 16 == 0: a multiple of 8)
        const int total = NQ * kp;
        auto item = [&](int e, int& q, int& k, int& rot) {
            const int l5 = e & 31, blk = e >> 5;
            const int qq = l5 >> 2, kk = l5 & 3;
            q = (blk % (NQ / 8)) * 8 + qq;
            k = ((blk / (NQ / 8)) * 4 + kk) * 2;
            rot = qq >> 1;
        };
        for (int e0 = tid; e0 < total; e0 += THREADS * (PB / 2)) {
            float4 v0[PB / 2], v1[PB / 2];
#pragma unroll
            for (int u = 0; u < PB / 2; ++u) {
                const int e = e0 + THREADS * u;
                int q, k, rot;
                item(e, q, k, rot);
                v0[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                v1[u] = v0[u];
                if (e < total && col0 + q * 4 < Nc) {  // Nc % 4 == 0
                    v0[u] = *reinterpret_cast<const float4*>(W + (size_t)k * ldw + col0 + q * 4);
                    v1[u] = *reinterpret_cast<const float4*>(W + (size_t)(k + 1) * ldw + col0 + q * 4);
                }
            }
#pragma unroll
            for (int u = 0; u < PB / 2; ++u) {
                const int e = e0 + THREADS * u;
                if (e < total) {
                    int q, k, rot;
                    item(e, q, k, rot);
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        const int c = (cc + rot) & 3;
                        const float a0 = c == 0 ? v0[u].x : c == 1 ? v0[u].y : c == 2 ? v0[u].z : v0[u].w;
                        const float a1 = c == 0 ? v1[u].x : c == 1 ? v1[u].y : c == 2 ? v1[u].z : v1[u].w;
                        uint32_t h, l;
                        split2(a0, a1, h, l);  // (k, k+1) of column n = 4q + c
                        *reinterpret_cast<uint32_t*>(&Bh[(size_t)(q * 4 + c) * pitch + k]) = h;
                        *reinterpret_cast<uint32_t*>(&Bl[(size_t)(q * 4 + c) * pitch + k]) = l;
                    }
                }
            }
        }
    }
    __syncthreads();
    float bcol[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) bcol[t] = (bias != nullptr && col0 + 32 * t + li < Nc) ? bias[col0 + 32 * t + li] : 0.f;
    const int ksteps = K >> 4;
    const int tiles = (M + TILE_ROWS - 1) / TILE_ROWS;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int r0 = tile * TILE_ROWS + wave * 32 * RB;
        if (r0 >= M) continue;  // wave-uniform; nothing below synchronises
        // rows of this lane in its row blocks (clamped: out-of-range rows are computed and dropped)
        const float* __restrict__ pa[RB];
        const float* __restrict__ ya[RB];
        float rs[RB];
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const int r = min(r0 + 32 * b + li, M - 1);
            const int ra = (DERIV && rscale != nullptr) ? r / rgroup : r;
            pa[b] = PA ? A + ((size_t)half * M + r) * 8 : A + (size_t)ra * lda + half * 8;
            ya[b] = DERIV ? Aux + (size_t)r * ldaux + half * 8 : nullptr;
            rs[b] = (DERIV && rscale != nullptr) ? rscale[r] : 1.f;
        }
        const bool scaled = DERIV && rscale != nullptr;
        const size_t a_step = PA ? (size_t)16 * M : (size_t)16;  // elements between the fragments of consecutive k-steps
        f32x16 acc[RB][NB];
#pragma unroll
        for (int b = 0; b < RB; ++b)
#pragma unroll
            for (int t = 0; t < NB; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[b][t][i] = 0.f;
        // raw A (and, for the data gradient, the activations whose derivative masks it) of the next D k-steps in flight;
        // the mask is applied when a fragment is built, so nothing waits on a load before its k-step comes up
        constexpr int D = DEPTH;
        float4 raw[D][RB][2], rawy[DERIV ? D : 1][RB][2];
        uint32_t rawm[DERIV ? D : 1][RB];
        const bool masked = DERIV && (AM < 0 ? (act_in != SNF_ACT_NONE && !aux_bits) : AM == 1);
        const bool bitmask = DERIV && (AM < 0 ? (act_in != SNF_ACT_NONE && aux_bits) : AM == 2);
        const uint8_t* __restrict__ mb[RB];
#pragma unroll
        for (int b = 0; b < RB; ++b)
            mb[b] = bitmask ? reinterpret_cast<const uint8_t*>(Aux) + (size_t)min(r0 + 32 * b + li, M - 1) * ldaux + half : nullptr;
        auto load8 = [&](int b, int s, int u) {
            raw[u][b][0] = *reinterpret_cast<const float4*>(pa[b] + s * a_step);
            raw[u][b][1] = *reinterpret_cast<const float4*>(pa[b] + s * a_step + 4);
            if constexpr (DERIV) {
                if (masked) {
                    rawy[u][b][0] = *reinterpret_cast<const float4*>(ya[b] + s * 16);
                    rawy[u][b][1] = *reinterpret_cast<const float4*>(ya[b] + s * 16 + 4);
                }
                if (bitmask) rawm[u][b] = mb[b][2 * s];  // columns 16 s + 8 half .. + 8 of the row: one byte
            }
        };
        auto frag = [&](int b, int u, bf16x8& hi, bf16x8& lo) {
            float4 v0 = raw[u][b][0], v1 = raw[u][b][1];
            if constexpr (DERIV) {
                if (scaled) {  // (the product the separate broadcast kernel would have written, then masked: same bits)
                    const float q = rs[b];
                    v0.x *= q; v0.y *= q; v0.z *= q; v0.w *= q;
                    v1.x *= q; v1.y *= q; v1.z *= q; v1.w *= q;
                }
                if (bitmask) {
                    const uint32_t m = rawm[u][b];
                    v0.x = (m & 1u) ? v0.x : 0.f; v0.y = (m & 2u) ? v0.y : 0.f; v0.z = (m & 4u) ? v0.z : 0.f; v0.w = (m & 8u) ? v0.w : 0.f;
                    v1.x = (m & 16u) ? v1.x : 0.f; v1.y = (m & 32u) ? v1.y : 0.f; v1.z = (m & 64u) ? v1.z : 0.f; v1.w = (m & 128u) ? v1.w : 0.f;
                }
                if (masked) {
                    const float4 y0 = rawy[u][b][0], y1 = rawy[u][b][1];
                    v0.x *= b3_act_deriv(y0.x, act_in); v0.y *= b3_act_deriv(y0.y, act_in);
                    v0.z *= b3_act_deriv(y0.z, act_in); v0.w *= b3_act_deriv(y0.w, act_in);
                    v1.x *= b3_act_deriv(y1.x, act_in); v1.y *= b3_act_deriv(y1.y, act_in);
                    v1.z *= b3_act_deriv(y1.z, act_in); v1.w *= b3_act_deriv(y1.w, act_in);
                }
            }
            uint32_t h[4], l[4];
            ws_split2(v0.x, v0.y, h[0], l[0]);
            ws_split2(v0.z, v0.w, h[1], l[1]);
            ws_split2(v1.x, v1.y, h[2], l[2]);
            ws_split2(v1.z, v1.w, h[3], l[3]);
            hi = __builtin_bit_cast(bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
            lo = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
        };
#pragma unroll
        for (int u = 0; u < D; ++u) {
#pragma unroll
            for (int b = 0; b < RB; ++b) load8(b, min(u, ksteps - 1), u);
        }
        for (int s = 0; s < ksteps; s += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                // this k-step's weight fragments first: their LDS latency passes under the split of the A fragments below
                bf16x8 bh[NB], bl[NB];
                {
                    const int ko = min(s + u, ksteps - 1) * 16 + half * 8;
#pragma unroll
                    for (int t = 0; t < NB; ++t) {
                        bh[t] = *reinterpret_cast<const bf16x8*>(&Bh[(size_t)(32 * t + li) * pitch + ko]);
                        bl[t] = *reinterpret_cast<const bf16x8*>(&Bl[(size_t)(32 * t + li) * pitch + ko]);
                    }
                }
                bf16x8 ah[RB], al[RB];
#pragma unroll
                for (int b = 0; b < RB; ++b) frag(b, u, ah[b], al[b]);
                // refill this buffer D k-steps ahead (unconditional, clamped: see AM above)
#pragma unroll
                for (int b = 0; b < RB; ++b) load8(b, min(s + u + D, ksteps - 1), u);
                if (s + u < ksteps) {
#pragma unroll
                    for (int t = 0; t < NB; ++t) {
                        if constexpr (CT) {
#pragma unroll
                            for (int b = 0; b < RB; ++b) acc[b][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[t], al[b], acc[b][t], 0, 0, 0);
#pragma unroll
                            for (int b = 0; b < RB; ++b) acc[b][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[t], ah[b], acc[b][t], 0, 0, 0);
#pragma unroll
                            for (int b = 0; b < RB; ++b) acc[b][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[t], ah[b], acc[b][t], 0, 0, 0);
                        } else {
#pragma unroll
                            for (int b = 0; b < RB; ++b) acc[b][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[b], bh[t], acc[b][t], 0, 0, 0);
#pragma unroll
                            for (int b = 0; b < RB; ++b) acc[b][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[b], bl[t], acc[b][t], 0, 0, 0);
#pragma unroll
                            for (int b = 0; b < RB; ++b) acc[b][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[b], bh[t], acc[b][t], 0, 0, 0);
                        }
                    }
                }
            }
        }
        if constexpr (CT) {
            // transposed accumulators: lane (m = li, half) holds, in registers 4q .. 4q+3, the output features
            // col0 + 32t + 8q + 4*half + {0..3} of row m -> one 16-byte store into level (col0 + 32t)/8 + q of the level-major C
#pragma unroll
            for (int b = 0; b < RB; ++b) {
                const int row = r0 + 32 * b + li;
                if (row < M) {
#pragma unroll
                    for (int t = 0; t < NB; ++t)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int c0 = col0 + 32 * t + 8 * q + 4 * half;
                            if (c0 < Nc) {  // Nc % 4 == 0
                                float4 v = make_float4(acc[b][t][4 * q], acc[b][t][4 * q + 1], acc[b][t][4 * q + 2], acc[b][t][4 * q + 3]);
                                if (bias != nullptr) { v.x += bias[c0]; v.y += bias[c0 + 1]; v.z += bias[c0 + 2]; v.w += bias[c0 + 3]; }
                                v.x = b3_act_apply(v.x, act_out); v.y = b3_act_apply(v.y, act_out);
                                v.z = b3_act_apply(v.z, act_out); v.w = b3_act_apply(v.w, act_out);
                                *reinterpret_cast<float4*>(C + ((size_t)(c0 >> 3) * M + row) * 8 + (c0 & 7)) = v;
                            }
                        }
                }
            }
        } else if (BT && !DERIV && hbar != nullptr) {
            // rendered epilogue (see the header): registers 0..7 of a lane are rows of the block's first group of 16, 8..15 of its
            // second; the other half-wave holds the other 8 rows of each group
            // (the activations themselves are only stored when the caller wants them: decided ONCE -- as a per-element `C != nullptr &&`
            //  it was 64 exec-mask round trips per wave and tile in the step's launches, which pass C = NULL)
            auto rendered = [&](auto store_c) {
            // the mask words of a row (32 columns each, one per column tile) are gathered in lane (reg, half) of the row's register
            // and leave as ONE store of NB words per row (a 4-byte store per row and column tile by lane 0 before: 64 store
            // instructions and as many exec-mask round trips per wave and tile)
            for (int b = 0; b < RB; ++b) {
                float wr[16];
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int row = r0 + 32 * b + (reg & 3) + 8 * (reg >> 2) + 4 * half;
                    wr[reg] = row < M ? rscale[row] : 0.f;
                }
                uint32_t mw[NB];
#pragma unroll
                for (int t = 0; t < NB; ++t) {
                    const int c = col0 + 32 * t + li;
                    float s0 = 0.f, s1 = 0.f;
                    mw[t] = 0u;
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int row = r0 + 32 * b + (reg & 3) + 8 * (reg >> 2) + 4 * half;
                        const float v = fmaxf(acc[b][t][reg], 0.f);  // (no bias on this path: b3_try_fwd_mean)
                        const unsigned long long bal = __ballot(v > 0.f && c < Nc);
                        const uint32_t mine = half ? (uint32_t)(bal >> 32) : (uint32_t)bal;
                        mw[t] = li == reg ? mine : mw[t];
                        if (reg < 8) s0 += wr[reg] * v; else s1 += wr[reg] * v;
                        if constexpr (decltype(store_c)::value) {
                            if (row < M && c < Nc) C[(size_t)row * ldc + c] = v;
                        }
                    }
                    s0 += __shfl_xor(s0, 32, 64);
                    s1 += __shfl_xor(s1, 32, 64);
                    const int g = ((r0 + 32 * b) >> 4) + half;  // half 0 writes the first group, half 1 the second
                    if (c < Nc && g * 16 < M) hbar[(size_t)g * Nc + c] = half ? s1 : s0;
                }
                if (li < 16) {  // lane (li, half) holds the words of row (li & 3) + 8 (li >> 2) + 4 half
                    const int row = r0 + 32 * b + (li & 3) + 8 * (li >> 2) + 4 * half;
                    if (row < M) {
                        uint32_t* __restrict__ dst = reinterpret_cast<uint32_t*>(ybits + (size_t)row * (Nc >> 3) + (col0 >> 3));
                        if constexpr (NB == 4) {
                            if (col0 + 128 <= Nc && ((Nc >> 3) & 15) == 0) *reinterpret_cast<uint4*>(dst) = make_uint4(mw[0], mw[1], mw[2], mw[3]);
                            else {
#pragma unroll
                                for (int t = 0; t < NB; ++t)
                                    if (col0 + 32 * t < Nc) dst[t] = mw[t];
                            }
                        } else {
#pragma unroll
                            for (int t = 0; t < NB; ++t)
                                if (col0 + 32 * t < Nc) dst[t] = mw[t];
                        }
                    }
                }
            }
            };
            if (C != nullptr) rendered(std::true_type{});
            else rendered(std::false_type{});
        } else {
        // epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
        for (int b = 0; b < RB; ++b)
#pragma unroll
            for (int t = 0; t < NB; ++t) {
                const int c = col0 + 32 * t + li;
                if (c < Nc) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int row = r0 + 32 * b + (reg & 3) + 8 * (reg >> 2) + 4 * half;
                        if (row < M) C[(size_t)row * ldc + c] = b3_act_apply(acc[b][t][reg] + bcol[t], act_out);
                    }
                }
            }
        }
    }
}

template <bool BT, bool DERIV, int BN, int RB, int THREADS, int DEPTH, bool PA = false, bool CT = false, int AM = -1>
static void ws_launch(dim3 grid, size_t lds, snf_stream_t stream, const float* A, const float* Aux, const float* W,
                      const float* bias, int M, int K, int Nc, int lda, int ldaux, int ldw, int ldc, int act_in, int act_out,
                      float* C, const float* rscale = nullptr, int rgroup = 1, int aux_bits = 0, float* hbar = nullptr,
                      uint8_t* ybits = nullptr) {
    auto kern = k_gemm_ws_b3<BT, DERIV, BN, RB, THREADS, DEPTH, PA, CT, AM>;
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(THREADS), lds, (hipStream_t)stream, A, Aux, W, bias, M, K, Nc, lda, ldaux, ldw, ldc,
                       act_in, act_out, C, rscale, rgroup, aux_bits, hbar, ybits);
}

// SNF_GEMM_WS_SMALL_LDS=<bytes>: weight slices whose 128-column LDS image exceeds <bytes> take the 64-column kernel (half the
// LDS, two workgroups per CU) -- a 135 KB workgroup (K = 256) can only start on a CU no co-running kernel occupies
static int ws_small_lds(size_t lds128) {
#ifndef SNF_GEMM_WS_SMALL_LDS
#define SNF_GEMM_WS_SMALL_LDS (1LL << 40)
#endif
    static const long long limit = SNF_GEMM_WS_SMALL_LDS;
    return (long long)lds128 > limit;
}

// takes the launch when the shape fits the weight-stationary kernel (K % 16 == 0, K <= 256, many rows); SNF_GEMM_WS=0 disables
template <bool BT, bool DERIV>
static int ws_try(const float* A, const float* Aux, const float* W, const float* bias, int M, int K, int Nc, int lda, int ldaux,
                  int ldw, int ldc, int act_in, int act_out, float* C, snf_stream_t stream, const float* rscale = nullptr,
                  int rgroup = 1, int aux_bits = 0, float* hbar = nullptr, uint8_t* ybits = nullptr) {
    static const int on = 1;
    const bool pa = lda < 0, ct = ldc < 0;  // level-major operands (F = 8): only this kernel reads / writes them
    if (pa || ct) {
        if ((pa && (lda != -8 || !BT || DERIV)) || (ct && (ldc != -8 || BT || !DERIV || (Nc % 8))) || (pa && ct) || (K % 16) ||
            K > 256 || K < 64 || Nc < 64 || (Nc % 4))
            return -1;
        const int small = ws_small_lds((size_t)2 * 128 * (K + 8) * sizeof(uint16_t));
        // 96-column slices where they tile the output exactly and 128 does not (192 = 2 x 96: the second 128-column slice would
        // run half its MFMAs on columns that do not exist) -- the data gradient from mask bits only (the step's head layers)
        static const int bn96_on = 1;
        const bool bn96 = bn96_on && ct && !small && (Nc % 96) == 0 && (Nc % 128) != 0 && act_in != SNF_ACT_NONE && aux_bits;
        const int bn = small ? 64 : bn96 ? 96 : 128, tile_rows = 256;
        const size_t lds = (size_t)2 * bn * (K + 8) * sizeof(uint16_t);
        const int gy = ceil_div(Nc, bn), tiles = ceil_div(M, tile_rows);
        const int per_cu = lds > 80 * 1024 ? 1 : 2;
        int gx = (256 * per_cu) / gy;
        if (gx < 1) gx = 1;
        if (gx > tiles) gx = tiles;
        dim3 grid(gx, gy);
        if constexpr (BT && !DERIV) {
            if (pa && !small) ws_launch<true, false, 128, 1, 512, SNF_WS_DEPTH, true, false>(grid, lds, stream, A, Aux, W, bias, M, K, Nc, lda, ldaux, ldw, ldc, act_in, act_out, C, rscale, rgroup, 0, hbar, ybits);
            if (pa && small) ws_launch<true, false, 64, 2, 256, 2, true, false>(grid, lds, stream, A, Aux, W, bias, M, K, Nc, lda, ldaux, ldw, ldc, act_in, act_out, C, rscale, rgroup, 0, hbar, ybits);
        }
        if constexpr (!BT && DERIV) {
            const int am = act_in == SNF_ACT_NONE ? 0 : aux_bits ? 2 : 1;
            if (ct && !small && am == 0) ws_launch<false, true, 128, 1, 512, SNF_WS_DEPTH, false, true, 0>(grid, lds, stream, A, Aux, W, bias, M, K, Nc, lda, ldaux, ldw, ldc, act_in, act_out, C, rscale, rgroup, aux_bits);
            if (ct && !small && am == 1) ws_launch<false, true, 128, 1, 512, SNF_WS_DEPTH, false, true, 1>(grid, lds, stream, A, Aux, W, bias, M, K, Nc, lda, ldaux, ldw, ldc, act_in, act_out, C, rscale, rgroup, aux_bits);
            if (ct && bn96) ws_launch<false, true, 96, 1, 512, SNF_WS_DEPTH, false, true, 2>(grid, lds, stream, A, Aux, W, bias, M, K, Nc, lda, ldaux, ldw, ldc, act_in, act_out, C, rscale, rgroup, aux_bits);
            if (ct && !small && am == 2 && !bn96) ws_launch<false, true, 128, 1, 512, SNF_WS_DEPTH, false, true, 2>(grid, lds, stream, A, Aux, W, bias, M, K, Nc, lda, ldaux, ldw, ldc, act_in, act_out, C, rscale, rgroup, aux_bits);
            if (ct && small) ws_launch<false, true, 64, 2, 256, 2, false, true>(grid, lds, stream, A, Aux, W, bias, M, K, Nc, lda, ldaux, ldw, ldc, act_in, act_out, C, rscale, rgroup, aux_bits);
        }
        return 1;
    }
    static const int min_rows = SNF_WS_MIN_ROWS;
    if (!on || (K % 16) || K > 256 || K < 64 || M < min_rows || Nc < 64 || (Nc % 4) || (lda % 4)) return 0;
    // BN 128, 8 waves x 32 rows, 4 k-steps of A loads in flight (2 and 8 measured the same: the loads are not latency-bound);
    // narrow outputs (Nc <= 64) and SNF_GEMM_WS_VARIANT=1 take BN 64 with 4 waves x 64 rows (two workgroups per CU)
    static const int variant = 0;
    const int v = (Nc <= 64 || variant == 1 || ws_small_lds((size_t)2 * 128 * (K + 8) * sizeof(uint16_t))) ? 1 : 0;
    const int bn = v == 1 ? 64 : 128, tile_rows = 256;
    const size_t lds = (size_t)2 * bn * (K + 8) * sizeof(uint16_t);
    const int gy = ceil_div(Nc, bn), tiles = ceil_div(M, tile_rows);
    const int per_cu = lds > 80 * 1024 ? 1 : 2;
    int gx = (256 * per_cu) / gy;
    if (gx < 1) gx = 1;
    if (gx > tiles) gx = tiles;
#if SNF_WS_EVEN_TILES
    // every workgroup the same number of row tiles: 16 tiles over 14 workgroups is two rounds for two of them and one for the rest -- the
    // launch lasts two rounds either way, on 8 workgroups per column slice it leaves the other CUs to the co-running streams
    while (gx > 1 && (tiles % gx) != 0) --gx;
#endif
    dim3 grid(gx, gy);
    const int am = !DERIV ? -1 : act_in == SNF_ACT_NONE ? 0 : aux_bits ? 2 : 1;
    if (v == 0 && am == -1) ws_launch<BT, DERIV, 128, 1, 512, SNF_WS_DEPTH>(grid, lds, stream, A, Aux, W, bias, M, K, Nc, lda, ldaux, ldw, ldc, act_in, act_out, C, rscale, rgroup, aux_bits, hbar, ybits);
    if constexpr (DERIV) {
        if (v == 0 && am == 0) ws_launch<BT, DERIV, 128, 1, 512, SNF_WS_DEPTH, false, false, 0>(grid, lds, stream, A, Aux, W, bias, M, K, Nc, lda, ldaux, ldw, ldc, act_in, act_out, C, rscale, rgroup, aux_bits, hbar, ybits);
        if (v == 0 && am == 1) ws_launch<BT, DERIV, 128, 1, 512, SNF_WS_DEPTH, false, false, 1>(grid, lds, stream, A, Aux, W, bias, M, K, Nc, lda, ldaux, ldw, ldc, act_in, act_out, C, rscale, rgroup, aux_bits, hbar, ybits);
        if (v == 0 && am == 2) ws_launch<BT, DERIV, 128, 1, 512, SNF_WS_DEPTH, false, false, 2>(grid, lds, stream, A, Aux, W, bias, M, K, Nc, lda, ldaux, ldw, ldc, act_in, act_out, C, rscale, rgroup, aux_bits, hbar, ybits);
    }
    if (v != 0) ws_launch<BT, DERIV, 64, 2, 256, 2>(grid, lds, stream, A, Aux, W, bias, M, K, Nc, lda, ldaux, ldw, ldc, act_in, act_out, C, rscale, rgroup, aux_bits, hbar, ybits);
    return 1;
}

// widest column tile (up to the SNF_B3_BN cap, default 128) that still leaves >= 256 workgroups (one per CU); 64 otherwise
static int b3_pick_bn(int M, int Nc) {
    static const int cap = 128;
    const int row_tiles = ceil_div(M, B3_BM);
    const int cands[3] = {256, 192, 128};
    for (int bn : cands) {
        if (bn > cap || Nc <= bn / 2) continue;
        if ((long long)row_tiles * ceil_div(Nc, bn) >= 256) return bn;
    }
    return 64;
}

#define B3_LAUNCH(BT_, DERIV_, BN_, grid, st, ...) \
    hipLaunchKernelGGL((k_gemm_rows_b3<BT_, DERIV_, BN_>), grid, dim3(256), 0, (hipStream_t)st, __VA_ARGS__)

static int g_gemm_mode = 1;  // 1: bf16x3 for the wide layers (default), 0: exact fp32 everywhere

bool b3_enabled() { return g_gemm_mode >= 1; }

}  // namespace snf

using namespace snf;

extern "C" int snf_set_gemm_mode(int mode) {
    SNF_REQUIRE(mode >= 0 && mode <= 2,
                "snf_set_gemm_mode: mode must be 0 (fp32), 1 (bf16x3 on the wide layers) or 2 (1 + bf16x3 64-wide chains)");
    g_gemm_mode = mode;
    return SNF_OK;
}

extern "C" int snf_get_gemm_mode(void) { return g_gemm_mode; }

// called by snf_linear_fwd / snf_linear_bwd_data for wide, aligned layers; returns 1 if it took the launch
int snf::b3_try_fwd(const float* X, const float* W, const float* bias, int N, int I, int O, int ldx, int ldy,
                              int act, float* Y, snf_stream_t stream) {
    if (ldx < 0)  // level-major X [I/8][N][8]: only the weight-stationary bf16x3 kernel reads it (-1: not supported)
        return (!b3_enabled() || (((uintptr_t)X | (uintptr_t)W) & 15))
                   ? -1 : ws_try<true, false>(X, nullptr, W, bias, N, I, O, ldx, 0, I, ldy, SNF_ACT_NONE, act, Y, stream);
    if (!b3_enabled() || I < 128 || O < 64 || (I % 4) || (O % 4) || (ldx % 4) || ((uintptr_t)X & 15) || ((uintptr_t)W & 15))
        return 0;
    if (ws_try<true, false>(X, nullptr, W, bias, N, I, O, ldx, 0, I, ldy, SNF_ACT_NONE, act, Y, stream)) return 1;
    const int bn = b3_pick_bn(N, O);
    dim3 grid(ceil_div(N, B3_BM), ceil_div(O, bn));
    const float* none = nullptr;
    if (bn == 256) B3_LAUNCH(true, false, 256, grid, stream, X, none, W, bias, N, I, O, ldx, 0, I, ldy, SNF_ACT_NONE, act, Y);
    else if (bn == 192) B3_LAUNCH(true, false, 192, grid, stream, X, none, W, bias, N, I, O, ldx, 0, I, ldy, SNF_ACT_NONE, act, Y);
    else if (bn == 128) B3_LAUNCH(true, false, 128, grid, stream, X, none, W, bias, N, I, O, ldx, 0, I, ldy, SNF_ACT_NONE, act, Y);
    else B3_LAUNCH(true, false, 64, grid, stream, X, none, W, bias, N, I, O, ldx, 0, I, ldy, SNF_ACT_NONE, act, Y);
    return 1;
}

// forward + ReLU with the rendered epilogue (hbar, mask bits; Y may be NULL): the weight-stationary kernel or nothing (-1)
int snf::b3_try_fwd_mean(const float* X, const float* W, int N, int I, int O, int ldx, int ldy, float* Y, const float* wk, int group,
                         float* hbar, uint8_t* ybits, snf_stream_t stream) {
    if (!b3_enabled() || group != 16 || (N % 16) || (O % 32) || (((uintptr_t)X | (uintptr_t)W | (uintptr_t)hbar | (uintptr_t)ybits) & 15))
        return -1;
    return ws_try<true, false>(X, nullptr, W, nullptr, N, I, O, ldx, 0, I, ldy, SNF_ACT_NONE, SNF_ACT_RELU, Y, stream, wk, group, 0,
                               hbar, ybits) > 0 ? 1 : -1;
}

int snf::b3_try_fwd_splitk(const float* X, const float* W, int N, int I, int O, int ldx, int ksplit, int splits, float* P,
                           snf_stream_t stream) {
    if (!b3_enabled() || I < 128 || O < 64 || (I % 4) || (O % 4) || (ldx % 4) || ((uintptr_t)X & 15) || ((uintptr_t)W & 15))
        return 0;
    dim3 grid(ceil_div(N, B3_BM), ceil_div(O, B3_BN), splits);
    const float* none = nullptr;
    B3_LAUNCH(true, false, 64, grid, stream, X, none, W, none, N, I, O, ldx, 0, I, O, SNF_ACT_NONE, SNF_ACT_NONE, P, ksplit,
              (long long)N * O);
    return 1;
}

int snf::b3_try_bwd_data(const float* dY, const float* Y, const float* W, int N, int I, int O, int lddy, int ldy,
                                   int lddx, int act, float* dX, snf_stream_t stream, const float* rscale, int rgroup,
                                   int aux_bits) {
    const bool y_bad = act != SNF_ACT_NONE && !aux_bits && ((ldy % 4) || ((uintptr_t)Y & 15));
    if (lddx < 0)  // level-major dX [I/8][N][8]: only the weight-stationary bf16x3 kernel writes it (-1: not supported)
        return (!b3_enabled() || (lddy % 4) || (((uintptr_t)dY | (uintptr_t)W | (uintptr_t)dX) & 15) || y_bad)
                   ? -1 : ws_try<false, true>(dY, Y, W, nullptr, N, O, I, lddy, ldy, I, lddx, act, SNF_ACT_NONE, dX, stream, rscale, rgroup, aux_bits);
    if (rscale != nullptr || aux_bits)  // grouped rows / mask bits: the weight-stationary kernel or nothing
        return (!b3_enabled() || (I % 4) || (O % 4) || (lddy % 4) || (((uintptr_t)dY | (uintptr_t)W) & 15) || y_bad)
                   ? -1 : (ws_try<false, true>(dY, Y, W, nullptr, N, O, I, lddy, ldy, I, lddx, act, SNF_ACT_NONE, dX, stream, rscale, rgroup, aux_bits) ? 1 : -1);
    if (!b3_enabled() || O < 128 || I < 64 || (I % 4) || (O % 4) || (lddy % 4) || ((uintptr_t)dY & 15) ||
        ((uintptr_t)W & 15) || (act != SNF_ACT_NONE && ((ldy % 4) || ((uintptr_t)Y & 15))))
        return 0;
    if (ws_try<false, true>(dY, Y, W, nullptr, N, O, I, lddy, ldy, I, lddx, act, SNF_ACT_NONE, dX, stream)) return 1;
    const int bn = b3_pick_bn(N, I);
    dim3 grid(ceil_div(N, B3_BM), ceil_div(I, bn));
    const float* none = nullptr;
    if (bn == 256) B3_LAUNCH(false, true, 256, grid, stream, dY, Y, W, none, N, O, I, lddy, ldy, I, lddx, act, SNF_ACT_NONE, dX);
    else if (bn == 192) B3_LAUNCH(false, true, 192, grid, stream, dY, Y, W, none, N, O, I, lddy, ldy, I, lddx, act, SNF_ACT_NONE, dX);
    else if (bn == 128) B3_LAUNCH(false, true, 128, grid, stream, dY, Y, W, none, N, O, I, lddy, ldy, I, lddx, act, SNF_ACT_NONE, dX);
    else B3_LAUNCH(false, true, 64, grid, stream, dY, Y, W, none, N, O, I, lddy, ldy, I, lddx, act, SNF_ACT_NONE, dX);
    return 1;
}

// full-width variant with a caller-provided partial buffer; returns 1 if it took the launch
long long snf::b3_wgrad_full_workspace_bytes(int N, int I, int O) {
    if (!b3_enabled() || O < 64 || O > 256 || I < 64 || I > 256 || (I % 4) || (O % 4) || N < 8192) return 0;
    return (long long)ceil_div(N, wf_rows_per_wg(N)) * O * I * (long long)sizeof(float);
}

int snf::b3_try_bwd_weight_full(const float* dY, const float* Y, const float* X, int N, int I, int O, int lddy, int ldy, int ldx,
                                int act, float* dW, float* dbias, void* workspace, long long workspace_bytes,
                                snf_stream_t stream, const float* rscale, int rgroup, int aux_bits) {
    static const int on = 1;
    const long long need = b3_wgrad_full_workspace_bytes(N, I, O);
    if (!on || need == 0 || dbias != nullptr || workspace == nullptr || workspace_bytes < need || (lddy % 4) ||
        (ldx < 0 ? (ldx != -8 || (I % 8)) : (ldx < I || (ldx % 4))) || (((uintptr_t)dY | (uintptr_t)X | (uintptr_t)workspace) & 15) ||
        (act != SNF_ACT_NONE && !aux_bits && ((ldy % 4) || ((uintptr_t)Y & 15))) || (aux_bits && (O % 8)))
        return 0;
    const int rows = wf_rows_per_wg(N), chunks = ceil_div(N, rows);
    float* P = (float*)workspace;
    // the train step's case (grouped rows, mask bits, level-major X, whole trips): the kernel whose loads are issued together
    const bool bits_case = SNF_WF_BITS_KERNEL && rscale != nullptr && aux_bits && act == SNF_ACT_RELU && ldx == -8 && rgroup >= 2 &&
                           (rgroup & (rgroup - 1)) == 0 && (N % 16) == 0 && (O % 32) == 0 && (I % 32) == 0 &&
                           ((uintptr_t)rscale & 7) == 0 && (long long)I * N < (1LL << 31);
    if (bits_case) {
        int rshift = 0;
        while ((1 << rshift) < rgroup) ++rshift;
        hipLaunchKernelGGL(k_wgrad_full_b3_bits, dim3(chunks), dim3(WF_T), 0, (hipStream_t)stream, dY, (const uint8_t*)Y, X, N, I, O,
                           lddy, ldy, rows, P, rscale, rshift);
    } else {
        hipLaunchKernelGGL(k_wgrad_full_b3, dim3(chunks), dim3(WF_T), 0, (hipStream_t)stream, dY, Y, X, N, I, O, lddy, ldy, ldx, act,
                           rows, P, rscale, rgroup, aux_bits);
    }
    hipLaunchKernelGGL(k_wgrad_full_reduce, dim3(ceil_div(O * I, 1024), SNF_WF_RSPLIT), dim3(256), 0, (hipStream_t)stream, P, chunks, O * I, dW);
    return 1;
}

int snf::b3_try_bwd_weight(const float* dY, const float* Y, const float* X, int N, int I, int O, int lddy, int ldy, int ldx,
                           int act, float* dW, float* dbias, snf_stream_t stream) {
    if (!b3_enabled() || O < 64 || I < 64 || (I % 4) || (O % 4) || (lddy % 4) || (ldx % 4) || ldx == 0 ||
        (ldx < 0 && (ldx != -8 || (I % 8))) || ((uintptr_t)dY & 15) ||
        ((uintptr_t)X & 15) || (act != SNF_ACT_NONE && ((ldy % 4) || ((uintptr_t)Y & 15))))
        return 0;
    const int to = ceil_div(O, 64), ti = ceil_div(I, 64);
    int chunks = 1024 / (to * ti);
    if (chunks < 1) chunks = 1;
    int rows = ceil_div(N, chunks);
    rows = ((rows + B3_BK - 1) / B3_BK) * B3_BK;
    if (rows < 4 * B3_BK) rows = 4 * B3_BK;
    chunks = ceil_div(N, rows);
    static const int xcd_order = 0;
    const int chunks8 = (chunks + 7) / 8 * 8;  // whole rounds of the 8 XCDs (surplus workgroups exit at once)
    const int am = (ldx > 0 && SNF_WGRAD_RAW_LOADS) ? (act == SNF_ACT_NONE ? 0 : 1) : -1;
    if (am == 0)
        hipLaunchKernelGGL(k_gemm_wgrad_b3<0>, dim3(to * ti * chunks8), dim3(256), 0, (hipStream_t)stream, dY, Y, X, N, I, O, lddy, ldy,
                           ldx, act, rows, dW, dbias, to, ti, chunks, xcd_order);
    else if (am == 1)
        hipLaunchKernelGGL(k_gemm_wgrad_b3<1>, dim3(to * ti * chunks8), dim3(256), 0, (hipStream_t)stream, dY, Y, X, N, I, O, lddy, ldy,
                           ldx, act, rows, dW, dbias, to, ti, chunks, xcd_order);
    else
        hipLaunchKernelGGL(k_gemm_wgrad_b3<-1>, dim3(to * ti * chunks8), dim3(256), 0, (hipStream_t)stream, dY, Y, X, N, I, O, lddy, ldy,
                           ldx, act, rows, dW, dbias, to, ti, chunks, xcd_order);
    return 1;
}

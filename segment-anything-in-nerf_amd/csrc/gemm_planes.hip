// gemm_planes.hip -- the image encoder's token GEMMs (ImageEncoderViT blocks: attn.qkv / attn.proj / mlp.lin1 / mlp.lin2,
// samnerf/segment_anything/modeling/image_encoder.py:164-236, common.py:13-28) on operands that arrive ALREADY split.
//
// Same arithmetic as linear_b3.hip (x = hi + lo in bf16, a product = hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16, fp32
// accumulate) but the split is not this kernel's work any more: the tiled kernel there re-split both operands per tile on the
// VALU and staged both through LDS behind two barriers per 32 k, which held it at 0.2 of the instruction's rate.  Here
//   * the weights are constant at inference: split ONCE into two bf16 planes [Nc][K] (snf_split_planes), staged into LDS by plain
//     16-byte copies, 64 k per chunk, double buffered -- one barrier per 4 MFMA k-steps;
//   * the activations are split by their PRODUCER (LayerNorm, attention, the GELU epilogue of lin1) into "k-blocked" planes
//     [K/8][M][8] bf16: the MFMA operand fragment of lane (row, half) for k-step s is the 16 bytes of block 2s + half at that row,
//     so a wave reads its fragments straight from global memory as two contiguous 512-byte runs -- no LDS, no VALU, no barrier --
//     and a wave owns its rows for the whole k loop;
//   * a workgroup is 4 waves x (32 RB) rows x (32 NB) columns; a wave issues 3 RB NB MFMAs per k-step against 2 NB LDS fragment
//     reads and 2 RB global fragment loads.
// CT ("transposed accumulate", C^T = W A^T: the weight fragment is the first MFMA operand): a lane then owns ONE row and four
// consecutive output features per accumulator quad, which is what writing the OUTPUT as k-blocked planes needs (8-byte stores, 512
// contiguous bytes per 32 rows): lin1's GELU output is born as lin2's operand.  Without CT the fp32 output leaves as 128-byte runs.
#include "common.hpp"

namespace snf {

typedef __bf16 gp_bf16x8 __attribute__((ext_vector_type(8)));
typedef float gp_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 gp_bf16x2 __attribute__((ext_vector_type(2)));
typedef float gp_f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t gp_u32x4 __attribute__((ext_vector_type(4)));  // (native vector: arrays of HIP's uint4 struct stayed in scratch memory)

constexpr int GP_KC = 64;            // k per staged weight chunk (4 MFMA k-steps)
constexpr int GP_PITCH = GP_KC + 8;  // bf16 per LDS row: 144 B, the b128 fragment reads of 16 lanes fall on distinct banks

__device__ __forceinline__ uint32_t gp_cvt_pk(float a, float b) {
    const gp_f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, gp_bf16x2));
}

// two fp32 -> packed bf16 hi pair and lo pair (lo = bf16(x - hi)); the split of linear_b3.hip
__device__ __forceinline__ void gp_split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    hi = gp_cvt_pk(x0, x1);
    lo = gp_cvt_pk(x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xFFFF0000u));
}

__device__ __forceinline__ float gp_act(float x, int act) {
    if (act == SNF_ACT_GELU) return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));  // nn.GELU() (erf form)
    if (act == SNF_ACT_RELU) return fmaxf(x, 0.f);
    return x;
}

// x [n] fp32 -> hi [n], lo [n] bf16 (row-major planes: the constant weights)
__global__ __launch_bounds__(256) void k_split_planes(const float* __restrict__ x, long long n4, uint2* __restrict__ hi,
                                                      uint2* __restrict__ lo) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    uint32_t h0, h1, l0, l1;
    gp_split2(v.x, v.y, h0, l0);
    gp_split2(v.z, v.w, h1, l1);
    hi[i] = make_uint2(h0, h1);
    lo[i] = make_uint2(l0, l1);
}

// x [M][K] fp32 row-major -> k-blocked planes [K/8][M][8] (the GEMM's activation operand; producers normally write it themselves)
__global__ __launch_bounds__(256) void k_split_planes_kb(const float* __restrict__ x, int M, int K, uint4* __restrict__ hi,
                                                         uint4* __restrict__ lo) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // (row, k-block), k-block fastest: coalesced reads
    const int kb = K >> 3;
    if (i >= (long long)M * kb) return;
    const int r = (int)(i / kb), q = (int)(i - (long long)r * kb);
    const float4 a = *reinterpret_cast<const float4*>(x + (size_t)r * K + q * 8);
    const float4 b = *reinterpret_cast<const float4*>(x + (size_t)r * K + q * 8 + 4);
    uint32_t h[4], l[4];
    gp_split2(a.x, a.y, h[0], l[0]);
    gp_split2(a.z, a.w, h[1], l[1]);
    gp_split2(b.x, b.y, h[2], l[2]);
    gp_split2(b.z, b.w, h[3], l[3]);
    hi[(size_t)q * M + r] = make_uint4(h[0], h[1], h[2], h[3]);
    lo[(size_t)q * M + r] = make_uint4(l[0], l[1], l[2], l[3]);
}

// C[M][Nc] = act(A W^T + bias).  A: k-blocked planes [K/8][M][8]; W: row-major planes [Nc][K]; K % 64 == 0, Nc % 8 == 0.
// Outputs (either or both): C fp32 row-major [M][Nc]; Chi / Clo k-blocked planes [Nc/8][M][8] (CT only).
// Tile order: workgroups reach the 8 XCDs round-robin by linear id, so id -> (xcd = id & 7, slot = id >> 3) -> tile xcd * per + slot:
// an XCD walks a CONTIGUOUS range of the tile list, and the list runs over groups of 8 column tiles, row tiles inside a group,
// the group's columns fastest -- the ~64 workgroups an XCD has in flight form an 8 x 8 block of tiles and share their operand rows
// and weight columns in that XCD's L2.
template <int RB, int NB, bool CT>
__global__ __launch_bounds__(256, (RB * NB <= 4 ? 2 : 1)) void k_gemm_planes(
    const uint16_t* Ahi, const uint16_t* Alo, const uint16_t* Whi, const uint16_t* Wlo,  // (no __restrict__: see GP_PIN)
    const float* __restrict__ bias, int M, int K, int Nc, int act, float* __restrict__ C,
    uint16_t* __restrict__ Chi, uint16_t* __restrict__ Clo, int row_tiles, int col_tiles) {
    constexpr int BN = 32 * NB, TR = 128 * RB;
    constexpr int BUF = 2 * BN * GP_PITCH;  // bf16 elements per LDS buffer (hi plane, lo plane)
    extern __shared__ __attribute__((aligned(16))) uint16_t gp_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, half = lane >> 5;
    // ---- tile of this workgroup
    const int total = row_tiles * col_tiles, per = (total + 7) >> 3;
    const int slot = (int)blockIdx.x >> 3, t = ((int)blockIdx.x & 7) * per + slot;
    if (slot >= per || t >= total) return;
    int rt, ct;
    {
        const int full = col_tiles >> 3, in_full = full * 8 * row_tiles;  // tiles inside complete groups of 8 column tiles
        if (t < in_full) {
            const int g = t / (8 * row_tiles), r = t - g * 8 * row_tiles;
            rt = r >> 3;
            ct = g * 8 + (r & 7);
        } else {
            const int w = col_tiles - full * 8, r = t - in_full;  // the last, narrower group
            rt = r / w;
            ct = full * 8 + (r - rt * w);
        }
    }
    const int row0 = rt * TR + wave * 32 * RB, col0 = ct * BN;
    // ---- operand addresses = wave-uniform base (scalar registers) + one 32-bit lane offset per row block / for all weight pieces
    int aoff[RB];  // elements; M K < 2^31 checked by the host
#pragma unroll
    for (int b = 0; b < RB; ++b) {
        const int r = min(row0 + 32 * b + li, M - 1);  // rows past M are computed on row M - 1 and dropped
        aoff[b] = (half * M + r) * 8;
    }
    const size_t a_step = (size_t)16 * M;  // elements between the fragments of consecutive k-steps (two k-blocks)
    // a weight chunk = 2 planes x BN rows x 8 pieces of 16 bytes; piece j of thread tid is row (tid >> 3) + 32 (j % NB), k piece
    // tid & 7 of plane j / NB (Nc % BN == 0: no clamp, so the row of piece j is a uniform distance from the row of piece 0)
    constexpr int PIECES = 2 * NB;
    const int woff = (tid >> 3) * K + (tid & 7) * 8;            // elements, < 2^31 (host)
    const int wl = (tid >> 3) * GP_PITCH + (tid & 7) * 8;       // LDS element offset of piece 0
    const uint16_t* wtile_h = Whi + (size_t)col0 * K;
    const uint16_t* wtile_l = Wlo + (size_t)col0 * K;
    gp_u32x4 wreg[PIECES];
    gp_u32x4 ra[4][RB][2];
    gp_f32x16 acc[RB][NB];
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
        for (int tt = 0; tt < NB; ++tt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[b][tt][i] = 0.f;
    const int ksteps = K >> 4, chunks = K / GP_KC;
    // ---- prologue: weight chunk 0 into buffer 0, the activation fragments of the first 4 k-steps in flight
    // GP_PIN: nothing moves across -- sched_barrier stops the machine scheduler, the empty asm with a memory clobber stops the passes
    // that sink loads towards their first use (IR level and MachineSink).  The operand pointers are NOT __restrict__ for that
    // reason: loads through a const __restrict__ kernel argument count as invariant and cross the clobber anyway.
#define GP_PIN()                                \
    do {                                        \
        asm volatile("" ::: "memory");          \
        __builtin_amdgcn_sched_barrier(0);      \
    } while (0)
    // (macros, not lambdas: with the arrays captured by reference the compiler kept `wreg` in scratch memory)
#define GP_WLOAD(chunk_)                                                                                                        \
    _Pragma("unroll") for (int j = 0; j < PIECES; ++j) {                                                                        \
        const uint16_t* ub = (j < NB ? wtile_h : wtile_l) + (size_t)(32 * (j % NB)) * K + (size_t)(chunk_) * GP_KC; \
        wreg[j] = *reinterpret_cast<const gp_u32x4*>(ub + woff);                                                                   \
    }
#define GP_WSTORE(buf_)                                                                                                         \
    _Pragma("unroll") for (int j = 0; j < PIECES; ++j)                                                                          \
        *reinterpret_cast<gp_u32x4*>(&gp_lds[(buf_) * BUF + (j < NB ? 0 : BN * GP_PITCH) + 32 * (j % NB) * GP_PITCH + wl]) = wreg[j];
#define GP_ALOAD(s_, u_)                                                                                                        \
    {                                                                                                                           \
        const uint16_t* uh = Ahi + (size_t)(s_) * a_step;                                                          \
        const uint16_t* ul = Alo + (size_t)(s_) * a_step;                                                          \
        _Pragma("unroll") for (int b = 0; b < RB; ++b) {                                                                        \
            ra[u_][b][0] = *reinterpret_cast<const gp_u32x4*>(uh + aoff[b]);                                                       \
            ra[u_][b][1] = *reinterpret_cast<const gp_u32x4*>(ul + aoff[b]);                                                       \
        }                                                                                                                       \
    }
#define GP_READ_B(Bb_, u_, h_, l_)                                                                                              \
    _Pragma("unroll") for (int tt = 0; tt < NB; ++tt) {                                                                         \
        h_[tt] = *reinterpret_cast<const gp_bf16x8*>(&(Bb_)[(32 * tt + li) * GP_PITCH + (u_) * 16 + half * 8]);                 \
        l_[tt] = *reinterpret_cast<const gp_bf16x8*>(&(Bb_)[BN * GP_PITCH + (32 * tt + li) * GP_PITCH + (u_) * 16 + half * 8]); \
    }
    GP_WLOAD(0)
#pragma unroll
    for (int u = 0; u < 3; ++u) GP_ALOAD(min(u, ksteps - 1), u)  // (slot 3 is filled by the loop's first step)
    GP_WSTORE(0)
    __syncthreads();
    // Program order is pinned with sched_barrier: left alone, the compiler sinks every global load of the loop body to its end and
    // waits for it there (the weight chunk's latency then sits between the last MFMA and the LDS write, once per chunk, and the
    // fragment ring buffer degenerates into load-then-use).
    // One k-step = one scheduling region (GP_PIN on both sides): its 3 RB NB MFMAs and, issued IN THEIR SHADOW (a wave that is alone
    // on its SIMD hides up to ~5 single-issue instructions per 32-cycle MFMA), this step's "fillers":
    //   every step : the refill of the fragment slot the PREVIOUS step consumed (same registers: the MFMAs that read them have issued),
    //                with the fragments 3 k-steps ahead;
    //   steps 0..2 : the LDS reads of the next step's weight fragments;      step 0: the global loads of the next weight chunk;
    //   step 3     : that chunk's LDS writes (other buffer), then the barrier and the first fragments of the next chunk.
    // sched_group_barrier spells the interleave out (one filler after each MFMA); left alone the scheduler issues the fillers in a
    // block before or after the MFMAs and the matrix pipe runs dry meanwhile (~200 of ~970 cycles per step at RB NB = 8).
    constexpr int NM = 3 * RB * NB;
#define GP_MFMAS()                                                                                                              \
    if constexpr (CT) {                                                                                                         \
        _Pragma("unroll") for (int tt = 0; tt < NB; ++tt)                                                                       \
            _Pragma("unroll") for (int b = 0; b < RB; ++b)                                                                      \
                acc[b][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[tt], al[b], acc[b][tt], 0, 0, 0);                       \
        _Pragma("unroll") for (int tt = 0; tt < NB; ++tt)                                                                       \
            _Pragma("unroll") for (int b = 0; b < RB; ++b)                                                                      \
                acc[b][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[tt], ah[b], acc[b][tt], 0, 0, 0);                       \
        _Pragma("unroll") for (int tt = 0; tt < NB; ++tt)                                                                       \
            _Pragma("unroll") for (int b = 0; b < RB; ++b)                                                                      \
                acc[b][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[tt], ah[b], acc[b][tt], 0, 0, 0);                       \
    } else {                                                                                                                    \
        _Pragma("unroll") for (int tt = 0; tt < NB; ++tt)                                                                       \
            _Pragma("unroll") for (int b = 0; b < RB; ++b)                                                                      \
                acc[b][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[b], bh[tt], acc[b][tt], 0, 0, 0);                       \
        _Pragma("unroll") for (int tt = 0; tt < NB; ++tt)                                                                       \
            _Pragma("unroll") for (int b = 0; b < RB; ++b)                                                                      \
                acc[b][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[b], bl[tt], acc[b][tt], 0, 0, 0);                       \
        _Pragma("unroll") for (int tt = 0; tt < NB; ++tt)                                                                       \
            _Pragma("unroll") for (int b = 0; b < RB; ++b)                                                                      \
                acc[b][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[b], bh[tt], acc[b][tt], 0, 0, 0);                       \
    }
    // (masks: 0x008 MFMA, 0x020 VMEM read, 0x100 DS read, 0x200 DS write)
#define GP_WEAVE(mask_, n_)                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < (n_); ++i) {                                                                          \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                      \
        __builtin_amdgcn_sched_group_barrier(mask_, 1, 0);                                                                      \
    }
    gp_bf16x8 bh[NB], bl[NB];
    GP_READ_B(gp_lds, 0, bh, bl)
    for (int c = 0; c < chunks; ++c) {
        const uint16_t* __restrict__ Bb = gp_lds + (c & 1) * BUF;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            GP_PIN();
            gp_bf16x8 ah[RB], al[RB], nh[NB], nl[NB];
#pragma unroll
            for (int b = 0; b < RB; ++b) {
                ah[b] = __builtin_bit_cast(gp_bf16x8, ra[u][b][0]);
                al[b] = __builtin_bit_cast(gp_bf16x8, ra[u][b][1]);
            }
            // (all loads unconditional, clamped: loads under branches make the compiler drain every load in flight first)
            if (u == 0) GP_WLOAD(min(c + 1, chunks - 1))
            if (u < 3) GP_READ_B(Bb, u + 1, nh, nl)
            GP_ALOAD(min(4 * c + u + 3, ksteps - 1), (u + 3) & 3)
            GP_MFMAS()
            if (u == 3) { GP_WSTORE((c + 1) & 1) }  // (that buffer was last read in chunk c - 1, behind a barrier)
            if (u < 3) GP_WEAVE(0x100, 2 * NB)
            GP_WEAVE(0x020, (u == 0 ? 2 * NB : 0) + 2 * RB)
            if (u == 3) GP_WEAVE(0x200, 2 * NB)
            if (u < 3) {
#pragma unroll
                for (int tt = 0; tt < NB; ++tt) { bh[tt] = nh[tt]; bl[tt] = nl[tt]; }
            }
        }
        GP_PIN();
        __syncthreads();
        GP_READ_B(gp_lds + ((c + 1) & 1) * BUF, 0, bh, bl)
    }
#undef GP_MFMAS
#undef GP_WEAVE
#undef GP_WLOAD
#undef GP_WSTORE
#undef GP_ALOAD
#undef GP_READ_B
#undef GP_PIN
    // ---- epilogue
    if constexpr (CT) {
        // lane (m = li, half), registers 4q .. 4q+3: output features col0 + 32 tt + 8 q + 4 half + {0..3} of row m
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const int row = row0 + 32 * b + li;
            if (row >= M) continue;
#pragma unroll
            for (int tt = 0; tt < NB; ++tt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c0 = col0 + 32 * tt + 8 * q + 4 * half;
                    if (c0 >= Nc) continue;  // Nc % 8 == 0
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gp_act(acc[b][tt][4 * q + e] + (bias ? bias[c0 + e] : 0.f), act);
                    if (C) *reinterpret_cast<float4*>(C + (size_t)row * Nc + c0) = make_float4(v[0], v[1], v[2], v[3]);
                    if (Chi) {
                        uint32_t h0, h1, l0, l1;
                        gp_split2(v[0], v[1], h0, l0);
                        gp_split2(v[2], v[3], h1, l1);
                        const size_t o = ((size_t)(c0 >> 3) * M + row) * 8 + (c0 & 7);
                        *reinterpret_cast<uint2*>(Chi + o) = make_uint2(h0, h1);
                        *reinterpret_cast<uint2*>(Clo + o) = make_uint2(l0, l1);
                    }
                }
        }
    } else {
        // C/D layout: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 half
#pragma unroll
        for (int b = 0; b < RB; ++b)
#pragma unroll
            for (int tt = 0; tt < NB; ++tt) {
                const int cc = col0 + 32 * tt + li;
                if (cc >= Nc) continue;
                const float bb = bias ? bias[cc] : 0.f;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int row = row0 + 32 * b + (reg & 3) + 8 * (reg >> 2) + 4 * half;
                    if (row < M) C[(size_t)row * Nc + cc] = gp_act(acc[b][tt][reg] + bb, act);
                }
            }
    }
}

template <int RB, int NB, bool CT>
static void gp_launch(hipStream_t st, const uint16_t* Ahi, const uint16_t* Alo, const uint16_t* Whi, const uint16_t* Wlo,
                      const float* bias, int M, int K, int Nc, int act, float* C, uint16_t* Chi, uint16_t* Clo) {
    constexpr int BN = 32 * NB, TR = 128 * RB;
    const int row_tiles = (M + TR - 1) / TR, col_tiles = (Nc + BN - 1) / BN;
    const int total = row_tiles * col_tiles, per = (total + 7) >> 3;
    const size_t lds = (size_t)2 * 2 * BN * GP_PITCH * sizeof(uint16_t);
    auto kern = k_gemm_planes<RB, NB, CT>;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL(kern, dim3(8 * per), dim3(256), lds, st, Ahi, Alo, Whi, Wlo, bias, M, K, Nc, act, C, Chi, Clo, row_tiles,
                       col_tiles);
}

}  // namespace snf

using namespace snf;

extern "C" int snf_split_planes(const float* x, int64_t n, uint16_t* hi, uint16_t* lo, snf_stream_t stream) {
    SNF_REQUIRE(x && hi && lo && n > 0 && (n % 4) == 0, "snf_split_planes: null pointer or n=%lld not a multiple of 4", (long long)n);
    SNF_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)hi % 8) == 0 && ((uintptr_t)lo % 8) == 0, "snf_split_planes: unaligned pointer");
    const long long n4 = n / 4;
    hipLaunchKernelGGL(k_split_planes, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n4, (uint2*)hi,
                       (uint2*)lo);
    SNF_LAUNCH_CHECK("snf_split_planes");
    return SNF_OK;
}

extern "C" int snf_split_planes_kb(const float* x, int M, int K, uint16_t* hi, uint16_t* lo, snf_stream_t stream) {
    SNF_REQUIRE(x && hi && lo && M > 0 && K > 0 && (K % 8) == 0, "snf_split_planes_kb: null pointer or K=%d not a multiple of 8", K);
    SNF_REQUIRE((((uintptr_t)x | (uintptr_t)hi | (uintptr_t)lo) % 16) == 0, "snf_split_planes_kb: unaligned pointer");
    const long long n = (long long)M * (K / 8);
    hipLaunchKernelGGL(k_split_planes_kb, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, M, K, (uint4*)hi,
                       (uint4*)lo);
    SNF_LAUNCH_CHECK("snf_split_planes_kb");
    return SNF_OK;
}

// tile shape by the number of workgroups it makes: the 256 x 128 tile (a wave's MFMAs against the fewest operand loads) where it
// still gives every CU two rounds of work, the 128 x 128 tile (two workgroups per CU) otherwise
static int gp_forward(const uint16_t* a_hi, const uint16_t* a_lo, const uint16_t* w_hi, const uint16_t* w_lo, const float* bias, int M,
                      int K, int Nc, int act, float* C, uint16_t* c_hi, uint16_t* c_lo, int force_rb, int force_nb,
                      snf_stream_t stream) {
    SNF_REQUIRE(a_hi && a_lo && w_hi && w_lo && (C || c_hi), "snf_linear_planes_fwd: null pointer");
    SNF_REQUIRE(M > 0 && K >= GP_KC && (K % GP_KC) == 0 && Nc > 0 && (Nc % 64) == 0,
                "snf_linear_planes_fwd: M=%d K=%d Nc=%d (K and Nc must be multiples of 64)", M, K, Nc);
    SNF_REQUIRE((long long)M * K < (1LL << 31) && (long long)Nc * K < (1LL << 31) && (long long)M * Nc < (1LL << 31),
                "snf_linear_planes_fwd: M=%d K=%d Nc=%d: operand offsets are 32-bit", M, K, Nc);
    SNF_REQUIRE((c_hi == nullptr) == (c_lo == nullptr), "snf_linear_planes_fwd: c_hi and c_lo go together");
    SNF_REQUIRE(act == SNF_ACT_NONE || act == SNF_ACT_RELU || act == SNF_ACT_GELU, "snf_linear_planes_fwd: act=%d", act);
    SNF_REQUIRE((((uintptr_t)a_hi | (uintptr_t)a_lo | (uintptr_t)w_hi | (uintptr_t)w_lo | (uintptr_t)C | (uintptr_t)c_hi |
                  (uintptr_t)c_lo) % 16) == 0, "snf_linear_planes_fwd: unaligned pointer");
    hipStream_t st = (hipStream_t)stream;
    const bool ct = c_hi != nullptr;  // (fp32-only outputs: the untransposed epilogue, 128-byte runs; measured equal to +-3 %)
    // Tile shape: among the shapes that tile Nc, the smallest estimated time
    //     rounds x (RB NB) x (k chunks + fixed cost of a tile in chunk units) x penalty
    // fitted to the encoder's four GEMMs (tools/bench_gemm_planes.py with every shape forced): rounds = tiles per CU, in
    // steps of 1 / (workgroups that share a CU) -- co-resident workgroups hide each other's prologue, epilogue and barriers, which is
    // why the short-k layers (qkv, lin1: 20 chunks) take the 256 x 64 / 128 x 128 tiles and the long one (lin2: 80 chunks, 1280
    // columns) the 128 x 160 tile that makes exactly 256 workgroups; a launch that leaves CUs idle runs its workgroups a little faster (clock,
    // L2 share): x 0.95.
    struct Shape { int rb, nb, per_cu; float fixed, penalty; };
    static const Shape shapes[] = {{2, 2, 2, 10.f, 1.00f}, {1, 4, 2, 10.f, 1.00f}, {2, 5, 1, 12.f, 1.00f}, {2, 4, 1, 12.f, 1.00f},
                                   {1, 5, 1, 12.f, 1.17f}, {1, 2, 4, 6.5f, 1.35f}};
    int rb = 0, nb = 0;
    float best = 0.f;
    for (const Shape& sh : shapes) {
        if (Nc % (32 * sh.nb)) continue;
        const long long tiles = (long long)((M + 128 * sh.rb - 1) / (128 * sh.rb)) * (Nc / (32 * sh.nb));
        const float rounds = (float)((tiles * sh.per_cu + 255) / 256) / (float)sh.per_cu;
        const float t = rounds * (sh.rb * sh.nb) * ((float)(K / GP_KC) + sh.fixed) * sh.penalty * (tiles < 256 ? 0.95f : 1.f);
        if (rb == 0 || t < best) { best = t; rb = sh.rb; nb = sh.nb; }
    }
    if (force_rb > 0) { rb = force_rb; nb = force_nb; }
    SNF_REQUIRE(rb > 0 && Nc % (32 * nb) == 0, "snf_linear_planes_fwd: no tile shape for Nc=%d (forced shape %d x %d)", Nc, force_rb, force_nb);
    bool launched = false;
#define GP_TRY(R_, N_)                                                                                          \
    if (rb == R_ && nb == N_) {                                                                                 \
        if (ct) gp_launch<R_, N_, true>(st, a_hi, a_lo, w_hi, w_lo, bias, M, K, Nc, act, C, c_hi, c_lo);        \
        else gp_launch<R_, N_, false>(st, a_hi, a_lo, w_hi, w_lo, bias, M, K, Nc, act, C, c_hi, c_lo);          \
        launched = true;                                                                                        \
    }
    GP_TRY(1, 2) GP_TRY(2, 2) GP_TRY(1, 4) GP_TRY(2, 4) GP_TRY(1, 5) GP_TRY(2, 5)
#undef GP_TRY
    SNF_REQUIRE(launched, "snf_linear_planes_fwd: tile shape %d x %d is not built", rb, nb);
    SNF_LAUNCH_CHECK("snf_linear_planes_fwd");
    return SNF_OK;
}

extern "C" int snf_linear_planes_fwd(const uint16_t* a_hi, const uint16_t* a_lo, const uint16_t* w_hi, const uint16_t* w_lo,
                                     const float* bias, int M, int K, int Nc, int act, float* C, uint16_t* c_hi, uint16_t* c_lo,
                                     snf_stream_t stream) {
    return gp_forward(a_hi, a_lo, w_hi, w_lo, bias, M, K, Nc, act, C, c_hi, c_lo, 0, 0, stream);
}

// the same with the tile shape given: (128 rb) rows x (32 nb) columns, (rb, nb) in {1, 2} x {2, 4, 5} -- tools/bench_gemm_planes.py
// times every shape against the choice above
extern "C" int snf_linear_planes_fwd_shape(const uint16_t* a_hi, const uint16_t* a_lo, const uint16_t* w_hi, const uint16_t* w_lo,
                                           const float* bias, int M, int K, int Nc, int act, float* C, uint16_t* c_hi,
                                           uint16_t* c_lo, int rb, int nb, snf_stream_t stream) {
    SNF_REQUIRE((rb == 1 || rb == 2) && (nb == 2 || nb == 4 || nb == 5), "snf_linear_planes_fwd_shape: rb=%d nb=%d", rb, nb);
    return gp_forward(a_hi, a_lo, w_hi, w_lo, bias, M, K, Nc, act, C, c_hi, c_lo, rb, nb, stream);
}

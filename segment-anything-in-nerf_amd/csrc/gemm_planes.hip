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

// GP_ABL (tools/ablate_gp.sh): pieces of the k loop compiled out, results WRONG -- 1: no activation-fragment loads, 2: no weight-fragment
// LDS reads, 4: no weight chunk staging (global load, LDS store, barrier), 8: the hi*hi products only (a third of the MFMAs),
// 16: every wave loads the fragments of rows 0 .. 32 RB (cache hits); k_gemm_planes_sh: 32: no LDS copies in the loop, 64: no barriers, 128: no epilogue
// GP_RING: activation-fragment slots in registers (k-steps requested ahead + 1): 4, or 8 (K % 128 == 0)
#ifndef GP_RING
#define GP_RING 4
#endif
#ifndef GP_ABL
#define GP_ABL 0
#endif
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
// WAVES = 8: a 512-thread workgroup, (256 RB) rows -- the same wave program, twice the rows behind one staged weight chunk: the
// bytes a CU pulls through its vector cache per product are 1 / (32 NB) for the fragments + 1 / (32 RB WAVES) for the weights.
template <int RB, int NB, bool CT, int WAVES = 4>
__global__ __launch_bounds__(64 * WAVES, (RB * NB <= 4 && WAVES == 4 ? 2 : 1)) void k_gemm_planes(
    const uint16_t* Ahi, const uint16_t* Alo, const uint16_t* Whi, const uint16_t* Wlo,  // (no __restrict__: see GP_PIN)
    const float* __restrict__ bias, int M, int K, int Nc, int act, float* __restrict__ C,
    uint16_t* __restrict__ Chi, uint16_t* __restrict__ Clo, int row_tiles, int col_tiles) {
    constexpr int BN = 32 * NB, TR = 32 * WAVES * RB;
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
        const int r = (GP_ABL & 16) ? 32 * b + li : min(row0 + 32 * b + li, M - 1);  // rows past M are computed on row M - 1 and dropped
        aoff[b] = (half * M + r) * 8;
    }
    const size_t a_step = (size_t)16 * M;  // elements between the fragments of consecutive k-steps (two k-blocks)
    // a weight chunk = 2 planes x BN rows x 8 pieces of 16 bytes; piece j of thread tid is row (tid >> 3) + 32 (j % NB), k piece
    // tid & 7 of plane j / NB (Nc % BN == 0: no clamp, so the row of piece j is a uniform distance from the row of piece 0)
    // (8 waves: the plane is a thread's own -- bit 3 of tid -- and it copies NB pieces of it, rows (tid >> 4) + 32 j)
    constexpr bool W8 = WAVES == 8;
    constexpr int PIECES = W8 ? NB : 2 * NB;
    const int wrow = W8 ? tid >> 4 : tid >> 3, wplane = W8 ? (tid >> 3) & 1 : 0;
    const int woff = wrow * K + (tid & 7) * 8;                                          // elements, < 2^31 (host)
    const int wl = wplane * BN * GP_PITCH + wrow * GP_PITCH + (tid & 7) * 8;            // LDS element offset of piece 0
    const uint16_t* wtile_h = (wplane ? Wlo : Whi) + (size_t)col0 * K;
    const uint16_t* wtile_l = Wlo + (size_t)col0 * K;
    gp_u32x4 wreg[PIECES];
    gp_u32x4 ra[GP_RING][RB][2];
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
        const uint16_t* ub = (W8 || j < NB ? wtile_h : wtile_l) + (size_t)(32 * (j % NB)) * K + (size_t)(chunk_) * GP_KC; \
        wreg[j] = *reinterpret_cast<const gp_u32x4*>(ub + woff);                                                                   \
    }
#define GP_WSTORE(buf_)                                                                                                         \
    _Pragma("unroll") for (int j = 0; j < PIECES; ++j)                                                                          \
        *reinterpret_cast<gp_u32x4*>(&gp_lds[(buf_) * BUF + (W8 || j < NB ? 0 : BN * GP_PITCH) + 32 * (j % NB) * GP_PITCH + wl]) = wreg[j];
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
    for (int u = 0; u < GP_RING - 1; ++u) GP_ALOAD(min(u, ksteps - 1), u)  // (the last slot is filled by the loop's first step)
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
    if constexpr ((GP_ABL & 8) != 0) {                                                                                          \
        _Pragma("unroll") for (int tt = 0; tt < NB; ++tt)                                                                       \
            _Pragma("unroll") for (int b = 0; b < RB; ++b)                                                                      \
                acc[b][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[b], bh[tt], acc[b][tt], 0, 0, 0);                       \
        asm volatile("" ::"v"(al[0]), "v"(bl[0]));                                                                              \
    } else if constexpr (CT) {                                                                                                         \
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
    constexpr int CPN = GP_RING / 4;  // chunks per trip: the slot of a k-step is a compile-time number
    for (int c0 = 0; c0 < chunks; c0 += CPN)
#pragma unroll
    for (int cp = 0; cp < CPN; ++cp) {
        const int c = c0 + cp;
        const uint16_t* __restrict__ Bb = gp_lds + (c & 1) * BUF;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            GP_PIN();
            gp_bf16x8 ah[RB], al[RB], nh[NB], nl[NB];
#pragma unroll
            for (int b = 0; b < RB; ++b) {
                ah[b] = __builtin_bit_cast(gp_bf16x8, ra[(4 * cp + u) & (GP_RING - 1)][b][0]);
                al[b] = __builtin_bit_cast(gp_bf16x8, ra[(4 * cp + u) & (GP_RING - 1)][b][1]);
            }
            // (all loads unconditional, clamped: loads under branches make the compiler drain every load in flight first)
            if (u == 0 && !(GP_ABL & 4)) GP_WLOAD(min(c + 1, chunks - 1))
            if (u < 3) {
                if constexpr (GP_ABL & 2) {
#pragma unroll
                    for (int tt = 0; tt < NB; ++tt) { nh[tt] = bh[tt]; nl[tt] = bl[tt]; }
                } else {
                    GP_READ_B(Bb, u + 1, nh, nl)
                }
            }
            if (!(GP_ABL & 1)) GP_ALOAD(min(4 * c + u + GP_RING - 1, ksteps - 1), (4 * cp + u + GP_RING - 1) & (GP_RING - 1))
            GP_MFMAS()
            if (u == 3 && !(GP_ABL & 4)) { GP_WSTORE((c + 1) & 1) }  // (that buffer was last read in chunk c - 1, behind a barrier)
            if (u < 3) GP_WEAVE(0x100, 2 * NB)
            GP_WEAVE(0x020, (u == 0 ? PIECES : 0) + 2 * RB)
            if (u == 3) GP_WEAVE(0x200, PIECES)
            if (u < 3) {
#pragma unroll
                for (int tt = 0; tt < NB; ++tt) { bh[tt] = nh[tt]; bl[tt] = nl[tt]; }
            }
        }
        GP_PIN();
        if (!(GP_ABL & 4)) __syncthreads();
        if (!(GP_ABL & 2)) GP_READ_B(gp_lds + ((c + 1) & 1) * BUF, 0, bh, bl)
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

template <int RB, int NB, bool CT, int WAVES = 4>
static void gp_launch(hipStream_t st, const uint16_t* Ahi, const uint16_t* Alo, const uint16_t* Whi, const uint16_t* Wlo,
                      const float* bias, int M, int K, int Nc, int act, float* C, uint16_t* Chi, uint16_t* Clo) {
    constexpr int BN = 32 * NB, TR = 32 * WAVES * RB;
    const int row_tiles = (M + TR - 1) / TR, col_tiles = (Nc + BN - 1) / BN;
    const int total = row_tiles * col_tiles, per = (total + 7) >> 3;
    const size_t lds = (size_t)2 * 2 * BN * GP_PITCH * sizeof(uint16_t);
    auto kern = k_gemm_planes<RB, NB, CT, WAVES>;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL(kern, dim3(8 * per), dim3(64 * WAVES), lds, st, Ahi, Alo, Whi, Wlo, bias, M, K, Nc, act, C, Chi, Clo, row_tiles,
                       col_tiles);
}


// ---- both operands through LDS ---------------------------------------------------------------------------------------------------
// k_gemm_planes above is bound by what a CU pulls through its vector cache (tools/ablate_gp.sh on 4096 x 1280 -> 5120, 256 x 64 tile:
// 150 us; fragment loads compiled out 115; every wave loading the SAME rows -- all cache hits -- still 140; twice the fragments in
// flight: no different): 81 % of the cache's 64 B/clk at the matrix rate, while the LDS (256 B/clk for b128 reads) idles.  Here BOTH
// operands are k-blocked planes ([K/8][M][8], [K/8][Nc][8]: the weights are split once, their layout is free) that arrive in LDS by
// global_load_lds_dwordx4 -- a piece is 64 consecutive rows of one k-block of one plane, 1 KB, LDS image = global image, conflict-free
// for the b128 fragment reads as it lies.  8 waves in a 4 x 2 grid, a wave owns 64 rows x (32 NB) columns, the tile is 256 rows x
// (64 NB) columns: per 16 k a CU moves (8 + 2 NB) x 2 KB through the cache for 6 NB x 8 MFMAs (29 % of its rate at NB = 5) and
// reads 8 x (2 + NB) x 2 KB of fragments (23 % of the LDS's).  4096 x 5120 makes exactly 256 tiles of 256 x 320.
// Two waves per SIMD with 256 registers each, because an LDS copy costs its wave ~100 issue cycles (MI355X_MICROARCH: 60 among bare
// MFMAs, more in a busy phase): a wave that is alone on its SIMD pays them in matrix time (the 4-wave / 512-register variant of this
// kernel with two full fragment sets ran 172 us where this one runs 137), here the SIMD's other wave issues MFMAs meanwhile.
// A stage is 32 k (72 KB at NB = 5), two stage buffers, one barrier at the END of a stage; the copies of stage c + 1 go out one
// behind the first MFMAs of each item of stage c and have landed (vmcnt(0), by hand) before that barrier.
// Measured (tools/bench_gemm_planes.py, k slope from K = 1280 / 2560 / 5120): 104.6 us per 1280 k + 32 us fixed against 119.5 + 25
// for the 256 x 64 tile; without copies and barriers 94.4 per 1280 k -- the matrix pipe at 0.68 of its zero-operand rate.
typedef __attribute__((address_space(1))) const void* gp_gptr;
typedef __attribute__((address_space(3))) void* gp_lptr;

template <int NB, bool CT>
__global__ __launch_bounds__(512, 1) void k_gemm_planes_sh(const uint16_t* Ahi, const uint16_t* Alo, const uint16_t* Whi,
                                                            const uint16_t* Wlo, const float* __restrict__ bias, int M, int K, int Nc,
                                                            int act, float* __restrict__ C, uint16_t* __restrict__ Chi,
                                                            uint16_t* __restrict__ Clo, int row_tiles, int col_tiles) {
    constexpr int RB = 2, TR = 256, BN = 64 * NB;
    constexpr int KB = 4;
    constexpr int A_PIECES = 2 * KB * 4, W_PIECES = 2 * KB * NB;
    constexpr int STAGE = (A_PIECES + W_PIECES) * 512;
    extern __shared__ __attribute__((aligned(16))) uint16_t gp_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, half = lane >> 5, wr = wave >> 1, wc = wave & 1;
    const int total = row_tiles * col_tiles, per = (total + 7) >> 3;
    const int slot = (int)blockIdx.x >> 3, t = ((int)blockIdx.x & 7) * per + slot;
    if (slot >= per || t >= total) return;
    int rt, ct;
    {
        const int full = col_tiles >> 3, in_full = full * 8 * row_tiles;
        if (t < in_full) {
            const int g = t / (8 * row_tiles), r = t - g * 8 * row_tiles;
            rt = r >> 3;
            ct = g * 8 + (r & 7);
        } else {
            const int w = col_tiles - full * 8, r = t - in_full;
            rt = r / w;
            ct = full * 8 + (r - rt * w);
        }
    }
    // copies of this wave per stage: activations, pieces wave + 8 j (j < 4): row group wave & 3, (plane, k-block) = (wave >> 2) + 2 j;
    // weights: plane wave >> 2, k-block wave & 3, the NB column groups
    const uint32_t off_a = (uint32_t)min(rt * TR + 64 * (wave & 3) + lane, M - 1) * 16u;
    const uint32_t off_w = (uint32_t)((wave & 3) * Nc + ct * BN + lane) * 16u;
    const size_t adv_a = (size_t)KB * M * 16, adv_w = (size_t)KB * Nc * 16;
    const char* wsel = reinterpret_cast<const char*>(wave >> 2 ? Wlo : Whi);
    gp_f32x16 acc[RB][NB];
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
        for (int tt = 0; tt < NB; ++tt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[b][tt][i] = 0.f;
    const int stages = K / (8 * KB);  // even (host)
    const uint16_t* fa = gp_lds + (half * TR + wr * 32 * RB + li) * 8;
    const uint16_t* fw = gp_lds + A_PIECES * 512 + (half * BN + wc * 32 * NB + li) * 8;
    // one k-step: fragments, then per 32-column block its 3 RB products (lo*hi, hi*lo, hi*hi per accumulator, the order of
    // k_gemm_planes) -- the first MFMAs need the first column block only
    // A stage = 2 NB items (k-step s, 32-column block tt): 2 weight-fragment reads + 3 RB MFMAs each (lo*hi, hi*lo, hi*hi per
    // accumulator, the order of k_gemm_planes).  The weight fragments run through a ring of 3 register slots, requested two items
    // ahead; the activation fragments of k-step 1 are requested during k-step 0 -- 216 registers instead of the 272 two full
    // fragment sets would take, and no MFMA waits on a read issued right in front of it (one fragment set: the loop without copies
    // and barriers still ran 133 of 143 us).  One copy of the next stage goes out behind the first MFMAs of an item.
    gp_bf16x8 xa[2][RB][2], xw[3][2];
#define GS_LDA(set_, buf_, s_)                                                                                                  \
    _Pragma("unroll") for (int b = 0; b < RB; ++b) {                                                                            \
        xa[set_][b][0] = *reinterpret_cast<const gp_bf16x8*>(fa + (buf_) * STAGE + ((2 * (s_)) * TR + 32 * b) * 8);             \
        xa[set_][b][1] = *reinterpret_cast<const gp_bf16x8*>(fa + (buf_) * STAGE + ((KB + 2 * (s_)) * TR + 32 * b) * 8);        \
    }
#define GS_LDW(slot_, buf_, s_, tt_)                                                                                            \
    {                                                                                                                           \
        xw[slot_][0] = *reinterpret_cast<const gp_bf16x8*>(fw + (buf_) * STAGE + ((2 * (s_)) * BN + 32 * (tt_)) * 8);           \
        xw[slot_][1] = *reinterpret_cast<const gp_bf16x8*>(fw + (buf_) * STAGE + ((KB + 2 * (s_)) * BN + 32 * (tt_)) * 8);      \
    }
#define GS_COPY1(stage_, buf_, j_)                                                                                              \
    if ((j_) < 4) {                                                                                                             \
        const int pq_ = (wave >> 2) + 2 * (j_);                                                                                 \
        const char* a_ = reinterpret_cast<const char*>(pq_ >= KB ? Alo : Ahi) + (size_t)(stage_) * adv_a;                      \
        __builtin_amdgcn_global_load_lds((gp_gptr)(a_ + (size_t)(pq_ % KB) * M * 16 + off_a),                                   \
                                         (gp_lptr)(gp_lds + (buf_) * STAGE + (8 * (j_) + wave) * 512), 16, 0, 0);               \
    } else {                                                                                                                    \
        __builtin_amdgcn_global_load_lds((gp_gptr)(wsel + (size_t)(stage_) * adv_w + ((j_) - 4) * 1024 + off_w),                \
                                         (gp_lptr)(gp_lds + (buf_) * STAGE + (A_PIECES + wave * NB + (j_) - 4) * 512), 16, 0,   \
                                         0);                                                                                    \
    }
#define GS_ITEM_MMA(set_, slot_, tt_)                                                                                           \
    if constexpr (CT) {                                                                                                         \
        _Pragma("unroll") for (int b = 0; b < RB; ++b)                                                                          \
            acc[b][tt_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xw[slot_][0], xa[set_][b][1], acc[b][tt_], 0, 0, 0);          \
        _Pragma("unroll") for (int b = 0; b < RB; ++b)                                                                          \
            acc[b][tt_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xw[slot_][1], xa[set_][b][0], acc[b][tt_], 0, 0, 0);          \
        _Pragma("unroll") for (int b = 0; b < RB; ++b)                                                                          \
            acc[b][tt_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xw[slot_][0], xa[set_][b][0], acc[b][tt_], 0, 0, 0);          \
    } else {                                                                                                                    \
        _Pragma("unroll") for (int b = 0; b < RB; ++b)                                                                          \
            acc[b][tt_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[set_][b][1], xw[slot_][0], acc[b][tt_], 0, 0, 0);          \
        _Pragma("unroll") for (int b = 0; b < RB; ++b)                                                                          \
            acc[b][tt_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[set_][b][0], xw[slot_][1], acc[b][tt_], 0, 0, 0);          \
        _Pragma("unroll") for (int b = 0; b < RB; ++b)                                                                          \
            acc[b][tt_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[set_][b][0], xw[slot_][0], acc[b][tt_], 0, 0, 0);          \
    }
    // (sched_group_barrier masks: 0x008 MFMA, 0x020 VMEM read, 0x100 DS read)
#define GS_ITEM(buf_, next_stage_, it)                                                                                          \
    if constexpr ((it) < 2 * NB) {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                                      \
        if constexpr ((it) + 2 < 2 * NB) GS_LDW(((it) + 2) % 3, buf_, ((it) + 2) / NB, ((it) + 2) % NB)                         \
        if constexpr ((it) == 1) { GS_LDA(1, buf_, 1) }                                                                         \
        if constexpr ((it) < 4 + NB && !(GP_ABL & 32)) { GS_COPY1(next_stage_, (buf_) ^ 1, it) }                                \
        GS_ITEM_MMA((it) / NB, (it) % 3, (it) % NB)                                                                             \
        __builtin_amdgcn_sched_group_barrier(0x100, ((it) + 2 < 2 * NB ? 2 : 0) + ((it) == 1 ? 2 * RB : 0), 0);                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                                      \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                                      \
        __builtin_amdgcn_sched_group_barrier(0x008, 3 * RB - 2, 0);                                                             \
    }
#define GS_STAGE(buf_, next_stage_)                                                                                             \
    GS_LDA(0, buf_, 0)                                                                                                          \
    GS_LDW(0, buf_, 0, 0)                                                                                                       \
    GS_LDW(1, buf_, 0, 1)                                                                                                       \
    GS_ITEM(buf_, next_stage_, 0) GS_ITEM(buf_, next_stage_, 1) GS_ITEM(buf_, next_stage_, 2) GS_ITEM(buf_, next_stage_, 3)     \
    GS_ITEM(buf_, next_stage_, 4) GS_ITEM(buf_, next_stage_, 5) GS_ITEM(buf_, next_stage_, 6) GS_ITEM(buf_, next_stage_, 7)     \
    GS_ITEM(buf_, next_stage_, 8) GS_ITEM(buf_, next_stage_, 9)
#define GS_SYNC()                                         \
    __builtin_amdgcn_sched_barrier(0);                    \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      \
    if (!(GP_ABL & 64)) __syncthreads();                   \
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4 + NB; ++j) { GS_COPY1(0, 0, j) }
    GS_SYNC()
    for (int c = 0; c < stages; c += 2) {
        GS_STAGE(0, c + 1)
        GS_SYNC()
        GS_STAGE(1, min(c + 2, stages - 1))  // (past the end: the last stage again, read by nothing)
        GS_SYNC()
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef GS_LDA
#undef GS_LDW
#undef GS_COPY1
#undef GS_ITEM_MMA
#undef GS_ITEM
#undef GS_STAGE
#undef GS_SYNC
    // ---- epilogue (the two of k_gemm_planes)
    const int row0 = rt * TR + wr * 32 * RB, col0 = ct * BN + wc * 32 * NB;
    if constexpr ((GP_ABL & 128) != 0) {  // (ablation: one store per lane keeps the accumulators alive)
        float sum = 0.f;
#pragma unroll
        for (int b = 0; b < RB; ++b)
#pragma unroll
            for (int tt = 0; tt < NB; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[b][tt][r];
        if (C) C[(size_t)min(row0 + li, M - 1) * Nc + col0 + half] = sum;
        return;
    }
    if constexpr (CT) {
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const int row = row0 + 32 * b + li;
            if (row >= M) continue;
#pragma unroll
            for (int tt = 0; tt < NB; ++tt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c0 = col0 + 32 * tt + 8 * q + 4 * half;
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
#pragma unroll
        for (int b = 0; b < RB; ++b)
#pragma unroll
            for (int tt = 0; tt < NB; ++tt) {
                const int cc = col0 + 32 * tt + li;
                const float bb = bias ? bias[cc] : 0.f;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int row = row0 + 32 * b + (reg & 3) + 8 * (reg >> 2) + 4 * half;
                    if (row < M) C[(size_t)row * Nc + cc] = gp_act(acc[b][tt][reg] + bb, act);
                }
            }
    }
}

template <int NB, bool CT>
static void gp_launch_sh(hipStream_t st, const uint16_t* Ahi, const uint16_t* Alo, const uint16_t* Whi, const uint16_t* Wlo,
                          const float* bias, int M, int K, int Nc, int act, float* C, uint16_t* Chi, uint16_t* Clo) {
    constexpr int BN = 64 * NB, TR = 256;
    const int row_tiles = (M + TR - 1) / TR, col_tiles = Nc / BN;
    const int total = row_tiles * col_tiles, per = (total + 7) >> 3;
    const size_t lds = (size_t)2 * 8 * (4 + NB) * 1024;
    auto kern = k_gemm_planes_sh<NB, CT>;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL(kern, dim3(8 * per), dim3(512), lds, st, Ahi, Alo, Whi, Wlo, bias, M, K, Nc, act, C, Chi, Clo, row_tiles,
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
    // (the 8-wave 256 x 128 tile: one workgroup per CU, its two waves per SIMD share the staged weights -- the 4900 x 1280 -> 1280
    // projection in one round of 200 workgroups, 54 us against 58)
    struct Shape { int rb, nb, waves, per_cu; float fixed, penalty; };
    static const Shape shapes[] = {{2, 2, 4, 2, 10.f, 1.00f}, {1, 4, 4, 2, 10.f, 1.00f}, {2, 5, 4, 1, 12.f, 1.00f},
                                   {2, 4, 4, 1, 12.f, 1.00f}, {1, 5, 4, 1, 12.f, 1.17f}, {1, 2, 4, 4, 6.5f, 1.35f},
                                   {1, 4, 8, 1, 10.f, 1.00f}};
    int rb = 0, nb = 0, waves = 4;
    float best = 0.f;
    for (const Shape& sh : shapes) {
        if (Nc % (32 * sh.nb)) continue;
        const int tr = 32 * sh.waves * sh.rb;
        const long long tiles = (long long)((M + tr - 1) / tr) * (Nc / (32 * sh.nb));
        const float rounds = (float)((tiles * sh.per_cu + 255) / 256) / (float)sh.per_cu;
        const float t = rounds * (sh.rb * sh.nb * sh.waves / 4) * ((float)(K / GP_KC) + sh.fixed) * sh.penalty * (tiles < 256 ? 0.95f : 1.f);
        if (rb == 0 || t < best) { best = t; rb = sh.rb; nb = sh.nb; waves = sh.waves; }
    }
    if (force_rb != 0) { rb = force_rb < 0 ? -force_rb : force_rb; nb = force_nb; waves = force_rb < 0 ? 8 : 4; }
    SNF_REQUIRE(rb > 0 && Nc % (32 * nb) == 0, "snf_linear_planes_fwd: no tile shape for Nc=%d (forced shape %d x %d)", Nc, force_rb, force_nb);
    bool launched = false;
#define GP_TRY(R_, N_, W_)                                                                                          \
    if (rb == R_ && nb == N_ && waves == W_) {                                                                      \
        if (ct) gp_launch<R_, N_, true, W_>(st, a_hi, a_lo, w_hi, w_lo, bias, M, K, Nc, act, C, c_hi, c_lo);        \
        else gp_launch<R_, N_, false, W_>(st, a_hi, a_lo, w_hi, w_lo, bias, M, K, Nc, act, C, c_hi, c_lo);          \
        launched = true;                                                                                            \
    }
    GP_TRY(1, 2, 4) GP_TRY(2, 2, 4) GP_TRY(1, 4, 4) GP_TRY(2, 4, 4) GP_TRY(1, 5, 4) GP_TRY(2, 5, 4) GP_TRY(1, 4, 8) GP_TRY(1, 5, 8)
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

// the same with the tile shape given: (128 rb) rows x (32 nb) columns, (rb, nb) in {1, 2} x {2, 4, 5}; rb = -1: the 8-wave workgroup,
// 256 rows x (32 nb) columns, nb in {4, 5} -- tools/bench_gemm_planes.py
// times every shape against the choice above
extern "C" int snf_linear_planes_fwd_shape(const uint16_t* a_hi, const uint16_t* a_lo, const uint16_t* w_hi, const uint16_t* w_lo,
                                           const float* bias, int M, int K, int Nc, int act, float* C, uint16_t* c_hi,
                                           uint16_t* c_lo, int rb, int nb, snf_stream_t stream) {
    SNF_REQUIRE(((rb == 1 || rb == 2) && (nb == 2 || nb == 4 || nb == 5)) || (rb == -1 && (nb == 4 || nb == 5)),
                "snf_linear_planes_fwd_shape: rb=%d nb=%d", rb, nb);
    return gp_forward(a_hi, a_lo, w_hi, w_lo, bias, M, K, Nc, act, C, c_hi, c_lo, rb, nb, stream);
}

// the same product with the WEIGHTS k-blocked too ([K/8][Nc][8]: snf_split_planes_kb of the [Nc][K] matrix) -- k_gemm_planes_sh, both
// operands through LDS: 256 x 320 tiles (Nc % 320 == 0) or 256 x 256 (Nc % 256 == 0); K % 64 == 0
extern "C" int snf_linear_planes_kb_fwd(const uint16_t* a_hi, const uint16_t* a_lo, const uint16_t* w_hi, const uint16_t* w_lo,
                                        const float* bias, int M, int K, int Nc, int act, float* C, uint16_t* c_hi, uint16_t* c_lo,
                                        snf_stream_t stream) {
    SNF_REQUIRE(a_hi && a_lo && w_hi && w_lo && (C || c_hi), "snf_linear_planes_kb_fwd: null pointer");
    SNF_REQUIRE(M > 0 && K >= 64 && (K % 64) == 0 && Nc > 0 && ((Nc % 320) == 0 || (Nc % 256) == 0),
                "snf_linear_planes_kb_fwd: M=%d K=%d Nc=%d (K must be a multiple of 64, Nc of 320 or 256)", M, K, Nc);
    SNF_REQUIRE((long long)M * K < (1LL << 31) && (long long)Nc * K < (1LL << 31) && (long long)M * Nc < (1LL << 31),
                "snf_linear_planes_kb_fwd: M=%d K=%d Nc=%d: operand offsets are 32-bit", M, K, Nc);
    SNF_REQUIRE((c_hi == nullptr) == (c_lo == nullptr), "snf_linear_planes_kb_fwd: c_hi and c_lo go together");
    SNF_REQUIRE(act == SNF_ACT_NONE || act == SNF_ACT_RELU || act == SNF_ACT_GELU, "snf_linear_planes_kb_fwd: act=%d", act);
    SNF_REQUIRE((((uintptr_t)a_hi | (uintptr_t)a_lo | (uintptr_t)w_hi | (uintptr_t)w_lo | (uintptr_t)C | (uintptr_t)c_hi |
                  (uintptr_t)c_lo) % 16) == 0, "snf_linear_planes_kb_fwd: unaligned pointer");
    hipStream_t st = (hipStream_t)stream;
    const bool ct = c_hi != nullptr;  // (fp32 output through the transposed epilogue -- 32-byte runs per row -- measured 136 -> 142 us)
    if ((Nc % 320) == 0) {
        if (ct) gp_launch_sh<5, true>(st, a_hi, a_lo, w_hi, w_lo, bias, M, K, Nc, act, C, c_hi, c_lo);
        else gp_launch_sh<5, false>(st, a_hi, a_lo, w_hi, w_lo, bias, M, K, Nc, act, C, c_hi, c_lo);
    } else {
        if (ct) gp_launch_sh<4, true>(st, a_hi, a_lo, w_hi, w_lo, bias, M, K, Nc, act, C, c_hi, c_lo);
        else gp_launch_sh<4, false>(st, a_hi, a_lo, w_hi, w_lo, bias, M, K, Nc, act, C, c_hi, c_lo);
    }
    SNF_LAUNCH_CHECK("snf_linear_planes_kb_fwd");
    return SNF_OK;
}

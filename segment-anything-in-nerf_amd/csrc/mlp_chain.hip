// mlp_chain.hip -- the 64-wide "fully fused" tiny MLPs (tcnn FullyFusedMLP role) on the fp32 matrix cores (gfx950).
//
// Nets: base  32 -> 64 -> 16          (nerfstudio/fields/nerfacto_field.py:157-175, density + 15 geo features)
//       head  31 -> 64 -> 64 -> 3     (nerfacto_field.py:228-240, SH16 ++ geo15 -> rgb, sigmoid)
// Layer-by-layer GEMMs move every hidden activation through HBM several times (4.6 GB per step for these two nets at
// 524k samples).  Here a wavefront carries 32 samples through the WHOLE net with activations in registers:
//
//   the product is formed transposed, H^T[neuron][sample] = W * X^T, with the weights as the MFMA A operand (read from
//   LDS, where all layers live) and the activations as the B operand.  v_mfma_f32_32x32x2_f32 leaves D[row][col] in
//   lane (col, half) / register r with row = (r&3) + 8*(r>>2) + 4*half -- and a B operand wants, per k-step, one k from
//   the lower half-wave and one from the upper.  Since the k order of a dot product is free, step r of the next layer
//   simply uses register r as its B operand and reads the weight column k = row(r, half): no transposes, no LDS
//   round trip for activations, no workgroup barriers in the tile loop (waves are independent).
//
//   forward : X -> [H1] -> [H2] -> Y          (hidden activations are stored once, for the backward)
//   backward: dY -> dH2 -> dH1 -> dX          (data-gradient chain; ReLU masks from the stored activations)
// Weight gradients are three tall-skinny GEMMs over the (pre-masked) dH / H pairs (linear.hip).
#include "common.hpp"
#include <stdlib.h>

namespace snf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// bit 0: the forward chain requests the next tile's input row a tile ahead; bit 1 (off): the two-hidden-layer fused backward its input
// row and output gradients -- measured 0.169 -> 0.176 ms: with the input row formed in the loader the row a tile ahead spills 16 registers
#ifndef SNF_CHAIN_PREFETCH
#define SNF_CHAIN_PREFETCH 1
#endif
#ifndef SNF_WG_BF
#define SNF_WG_BF 1  // fused backward of the colour net: the input row by the branch-free loader of the forward (chain_sh_load_bf)
#endif
constexpr int MC_H = 64;        // hidden width
constexpr int MC_IN = 32;       // (padded) input width
constexpr int MC_P0 = 33;       // LDS pitch of W0 [64][32]
constexpr int MC_P1 = 65;       // LDS pitch of W1 [64][64] and Wout [32][64]

__device__ __forceinline__ int krow(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

struct ChainWeights {
    float* w0;    // [64][MC_P0]   W0[o][i], columns >= in_real zero
    float* w1;    // [64][MC_P1]   W1[o][k]           (NH == 2 only)
    float* wo;    // [32][MC_P1]   Wout[o][k], rows >= out zero
};

template <int NH>
__device__ __forceinline__ void load_chain_weights(float* lds, ChainWeights& cw, const float* __restrict__ W0, int in_real,
                                                   const float* __restrict__ W1, const float* __restrict__ Wout,
                                                   int out) {
    cw.w0 = lds;
    cw.w1 = lds + MC_H * MC_P0;
    cw.wo = cw.w1 + (NH == 2 ? MC_H * MC_P1 : 0);
    // (four elements per thread and trip, the loads unconditional and first: see mc_stage)
    const int nt = (int)blockDim.x;
    auto stage = [&](int total, auto addr, auto dst) {
        for (int i0 = threadIdx.x; i0 < total; i0 += 4 * nt) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = addr(min(i0 + u * nt, total - 1));
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u * nt < total) dst(i0 + u * nt, v[u]);
        }
    };
    stage(MC_H * MC_IN,
          [&](int i) { const int o = i / MC_IN, c = i % MC_IN; const bool ok = c < in_real; const float v = W0[ok ? o * in_real + c : 0]; return ok ? v : 0.f; },
          [&](int i, float v) { cw.w0[(i / MC_IN) * MC_P0 + (i % MC_IN)] = v; });
    if constexpr (NH == 2) {
        stage(MC_H * MC_H, [&](int i) { return W1[i]; }, [&](int i, float v) { cw.w1[(i / MC_H) * MC_P1 + (i % MC_H)] = v; });
    }
    stage(32 * MC_H,
          [&](int i) { const int o = i / MC_H, c = i % MC_H; const bool ok = o < out; const float v = Wout[ok ? o * MC_H + c : 0]; return ok ? v : 0.f; },
          [&](int i, float v) { cw.wo[(i / MC_H) * MC_P1 + (i % MC_H)] = v; });
    __syncthreads();
}

constexpr int chain_lds_floats(int NH) { return MC_H * MC_P0 + (NH == 2 ? MC_H * MC_P1 : 0) + 32 * MC_P1; }

// next layer (64 wide) from a 64-wide activation held as two D tiles: out[u] = sum_t sum_r A(W[u*32+li][t*32+krow]) * act[t][r]
__device__ __forceinline__ void layer_64x64(const float* __restrict__ w, int pitch, const f32x16 (&act)[2], f32x16 (&out)[2],
                                            int li, int half) {
    out[0] = zero16();
    out[1] = zero16();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = t * 32 + krow(r, half);
            const float b = act[t][r];
            out[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[li * pitch + k], b, out[0], 0, 0, 0);
            out[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[(32 + li) * pitch + k], b, out[1], 0, 0, 0);
        }
    }
}

// store a 64-wide D-tile pair as rows of a [N,64] row-major matrix (lane (s,half) owns columns t*32 + 8q + 4half + 0..3)
__device__ __forceinline__ void store_h64(float* __restrict__ H, long long s, const f32x16 (&a)[2], int half) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(H + s * MC_H + t * 32 + 8 * q + 4 * half) =
                make_float4(a[t][4 * q], a[t][4 * q + 1], a[t][4 * q + 2], a[t][4 * q + 3]);
}

__device__ __forceinline__ void load_h64(const float* __restrict__ H, long long s, f32x16 (&a)[2], int half) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(H + s * MC_H + t * 32 + 8 * q + 4 * half);
            a[t][4 * q] = v.x; a[t][4 * q + 1] = v.y; a[t][4 * q + 2] = v.z; a[t][4 * q + 3] = v.w;
        }
}

// ------------------------------------------------------------------------------------------------------------------
template <int NH>
__global__ __launch_bounds__(256) void k_mlp_chain_fwd(const float* __restrict__ X, int ldx, const float* __restrict__ W0,
                                                       int in_real, const float* __restrict__ W1,
                                                       const float* __restrict__ Wout, int out, int out_act, long long N,
                                                       float* __restrict__ H1, float* __restrict__ H2,
                                                       float* __restrict__ Y, int ldy) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    ChainWeights cw;
    load_chain_weights<NH>(lds, cw, W0, in_real, W1, Wout, out);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, half = lane >> 5;
    const long long ntiles = (N + 31) / 32;
    for (long long tile = (long long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long long)gridDim.x * 4) {
        const long long s = tile * 32 + li;
        const bool ok = s < N;
        const long long sc = ok ? s : N - 1;
        // this lane's half of the input row: X[s][half*16 .. half*16+15]
        float x[16];
        if (ldx == 0) {
            // level-major ("planar") input [16 levels][N][2] (the hash-grid forward's ld_out = 0 layout): feature k lives at
            // ((k >> 1) * N + s) * 2 + (k & 1); a half-wave reads 32 consecutive samples of a level = 256 contiguous bytes
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float2 v = *reinterpret_cast<const float2*>(X + ((long long)(half * 8 + q) * N + sc) * 2);
                x[2 * q] = v.x; x[2 * q + 1] = v.y;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(X + sc * ldx + half * 16 + 4 * q);
                x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (half * 16 + i >= in_real) x[i] = 0.f;  // pad columns may hold anything (select, not multiply)
        // layer 0: H1^T[o][s], k pairing (step, 16 + step)
        f32x16 h1[2] = {zero16(), zero16()};
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            const int k = half * 16 + st;
            h1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cw.w0[li * MC_P0 + k], x[st], h1[0], 0, 0, 0);
            h1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cw.w0[(32 + li) * MC_P0 + k], x[st], h1[1], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) h1[t][r] = fmaxf(h1[t][r], 0.f);
        if (H1 != nullptr && ok) store_h64(H1, s, h1, half);
        f32x16 last[2];
        if constexpr (NH == 2) {
            layer_64x64(cw.w1, MC_P1, h1, last, li, half);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) last[t][r] = fmaxf(last[t][r], 0.f);
            if (H2 != nullptr && ok) store_h64(H2, s, last, half);
        } else {
            last[0] = h1[0];
            last[1] = h1[1];
        }
        // output layer (<= 32 neurons): Y^T[o][s]
        f32x16 y = zero16();
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                y = __builtin_amdgcn_mfma_f32_32x32x2f32(cw.wo[li * MC_P1 + t * 32 + krow(r, half)], last[t][r], y, 0, 0, 0);
        if (ok) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = krow(r, half);
                if (o < out) {
                    float v = y[r];
                    if (out_act == SNF_ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
                    else if (out_act == SNF_ACT_RELU) v = fmaxf(v, 0.f);
                    Y[s * ldy + o] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// dZ[s][o] = dY[s*lddy + dy_col_off + o]  (o >= 1, or all o when dY0 == NULL) ; dZ[s][0] = dY0[s] when dY0 != NULL.
// With out_act == SIGMOID the derivative y(1-y) of the stored output Y is applied.
template <int NH>
__global__ __launch_bounds__(256) void k_mlp_chain_bwd(const float* __restrict__ dY, int lddy, int dy_col_off,
                                                       const float* __restrict__ dY0, const float* __restrict__ Yout,
                                                       int ldy, const float* __restrict__ W0, int in_real,
                                                       const float* __restrict__ W1, const float* __restrict__ Wout,
                                                       int out, int out_act, long long N, const float* __restrict__ H1,
                                                       const float* __restrict__ H2, float* __restrict__ dH1,
                                                       float* __restrict__ dH2, float* __restrict__ dZout, int lddz,
                                                       float* __restrict__ dX, int lddx) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    ChainWeights cw;
    load_chain_weights<NH>(lds, cw, W0, in_real, W1, Wout, out);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, half = lane >> 5;
    const int outp = (out + 1) & ~1;  // padded to even: k pairing (step, outp/2 + step)
    const int hsteps = outp >> 1;
    const long long ntiles = (N + 31) / 32;
    for (long long tile = (long long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long long)gridDim.x * 4) {
        const long long s = tile * 32 + li;
        const bool ok = s < N;
        const long long sc = ok ? s : N - 1;
        // ---- dLast^T[k][s] = sum_o Wout[o][k] dZ^T[o][s]
        f32x16 dl[2] = {zero16(), zero16()};
        for (int st = 0; st < hsteps; ++st) {
            const int o = half * hsteps + st;
            float dz = 0.f;
            if (o < out) {
                dz = (o == 0 && dY0 != nullptr) ? dY0[sc] : dY[sc * lddy + dy_col_off + o];
                if (out_act == SNF_ACT_SIGMOID) {
                    const float yv = Yout[sc * ldy + o];
                    dz *= yv * (1.f - yv);
                }
                if (dZout != nullptr && ok) dZout[s * lddz + o] = dz;  // pre-activation output gradient, for the wgrad GEMM
            }
            const int oc = o < 32 ? o : 31;
            dl[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cw.wo[oc * MC_P1 + li], dz, dl[0], 0, 0, 0);
            dl[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cw.wo[oc * MC_P1 + 32 + li], dz, dl[1], 0, 0, 0);
        }
        f32x16 hh[2];
        if constexpr (NH == 2) {
            load_h64(H2, sc, hh, half);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) dl[t][r] = hh[t][r] > 0.f ? dl[t][r] : 0.f;
            if (ok) store_h64(dH2, s, dl, half);
            // dH1^T[k1][s] = sum_k2 W1[k2][k1] dH2^T[k2][s]
            f32x16 d1[2] = {zero16(), zero16()};
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int k2 = u * 32 + krow(r, half);
                    const float b = dl[u][r];
                    d1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cw.w1[k2 * MC_P1 + li], b, d1[0], 0, 0, 0);
                    d1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cw.w1[k2 * MC_P1 + 32 + li], b, d1[1], 0, 0, 0);
                }
            dl[0] = d1[0];
            dl[1] = d1[1];
        }
        load_h64(H1, sc, hh, half);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) dl[t][r] = hh[t][r] > 0.f ? dl[t][r] : 0.f;
        if (ok) store_h64(dH1, s, dl, half);
        // ---- dX^T[i][s] = sum_k1 W0[k1][i] dH1^T[k1][s]
        if (dX != nullptr) {
            f32x16 dx = zero16();
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    dx = __builtin_amdgcn_mfma_f32_32x32x2f32(cw.w0[(t * 32 + krow(r, half)) * MC_P0 + li], dl[t][r], dx, 0,
                                                              0, 0);
            if (ok) {
                if (lddx == 0) {
                    // level-major dX [16][N][2]: registers 4q .. 4q+3 are features 8q + 4*half + {0..3} = two levels; this is
                    // the staged-gradient layout of the hash-grid backward (snf_hashgrid_bwd_presorted with ld_out = 0)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const long long lv = 4 * q + 2 * half;
                        *reinterpret_cast<float2*>(dX + (lv * N + s) * 2) = make_float2(dx[4 * q], dx[4 * q + 1]);
                        *reinterpret_cast<float2*>(dX + ((lv + 1) * N + s) * 2) = make_float2(dx[4 * q + 2], dx[4 * q + 3]);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<float4*>(dX + s * lddx + 8 * q + 4 * half) =
                            make_float4(dx[4 * q], dx[4 * q + 1], dx[4 * q + 2], dx[4 * q + 3]);
                }
            }
        }
    }
}

// ==================================================================================================================
// The same chains on the bf16 matrix cores with the 3-term split (hi*hi + hi*lo + lo*hi, fp32 accumulate; linear_b3.hip has
// the error analysis): a 64 -> 64 layer is 24 v_mfma_f32_32x32x16_bf16 (768 cycles) instead of 64 v_mfma_f32_32x32x2_f32
// (4096 cycles) -- the fp32 chains ran at 22-31 % of the fp32 matrix peak and were bound by exactly that.
//
// The register trick carries over.  A 32x32x16 MFMA wants from lane (n, half) the 8 k-values 8*half .. 8*half+7 of its k-step
// as one bf16x8 B operand.  The accumulator of the previous layer holds, in lane (n, half), rows
// row(r, half) = (r&3) + 8*(r>>2) + 4*half for r = 0..15: registers 8s .. 8s+7 are 8 rows, and over the two halves they are
// 16 DISTINCT rows -- a valid k-step.  So k-step s of the next layer uses registers 8s .. 8s+7 (split to hi / lo in
// registers) as its B operand, and the weight planes in LDS are stored with their k axis permuted to match
// (slot (t, s, half, j) <-> input t*32 + row(8s + j, half)): still no transposes, no LDS round trip, no barriers.
// Opt-in: snf_set_gemm_mode(2) (see chain_b3_on below); modes 0 and 1 run the exact-fp32 chains above.
typedef __bf16 mc_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 mc_bf16x2 __attribute__((ext_vector_type(2)));
typedef float mc_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t mc_cvt_pk(float a, float b) {
    const mc_f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, mc_bf16x2));
}

__device__ __forceinline__ void mc_split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    hi = mc_cvt_pk(x0, x1);
    lo = mc_cvt_pk(x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xFFFF0000u));
}

// three bf16 pieces (24 mantissa bits): hi + mid + lo reproduces an fp32 value to its last bit or two
__device__ __forceinline__ void mc_split3(float x0, float x1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    hi = mc_cvt_pk(x0, x1);
    const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xFFFF0000u);
    mid = mc_cvt_pk(r0, r1);
    lo = mc_cvt_pk(r0 - __uint_as_float(mid << 16), r1 - __uint_as_float(mid & 0xFFFF0000u));
}

constexpr int MC_BP32 = 40;  // bf16 elements per plane row with 32 k-slots (80 B: conflict-free b128 reads)
constexpr int MC_BP64 = 72;  // ... with 64 k-slots (144 B)

// k-slot -> input index.  LIN: the input is 16 floats per half-wave lane (feature half*16 + i): slot = s*16 + half*8 + j.
// otherwise the input is an accumulator tile set: slot = t*32 + s*16 + half*8 + j -> t*32 + row(8s + j, half).
template <bool LIN>
__device__ __forceinline__ int mc_slot_k(int slot) {
    const int j = slot & 7, half = (slot >> 3) & 1, s = (slot >> 4) & 1, t = slot >> 5;
    if constexpr (LIN) return half * 16 + 8 * s + j;
    return t * 32 + krow(8 * s + j, half);
}

// guarded weight element for the staging below: the load is UNCONDITIONAL (element 0 when the guard fails), the guard a select behind
// it -- a load under a branch is a basic block with its own wait
__device__ __forceinline__ float mc_ld(const float* __restrict__ W, int idx, bool ok) {
    const float v = W[ok ? idx : 0];
    return ok ? v : 0.f;
}

// planes[m][slot] = split(get(m, k(slot))) for m < rows, slot < slots (slots even).  Four pairs per thread and trip, all eight loads
// first: one pair per trip with its split right behind the loads was one memory latency per trip -- 4 + 8 + 4 (+ 4 + 8 for the
// recomputing backward) trips at the top of a kernel that runs 40 - 160 us.
template <bool LIN, class G>
__device__ __forceinline__ void mc_stage(uint16_t* __restrict__ Ph, uint16_t* __restrict__ Pl, int pitch, int rows, int slots,
                                         G get) {
    const int pairs = slots >> 1, total = rows * pairs, nt = (int)blockDim.x;
    for (int i0 = threadIdx.x; i0 < total; i0 += 4 * nt) {
        float a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = min(i0 + u * nt, total - 1);
            const int m = i / pairs, sl = (i - m * pairs) * 2;
            a[u] = get(m, mc_slot_k<LIN>(sl));
            b[u] = get(m, mc_slot_k<LIN>(sl + 1));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * nt;
            if (i < total) {
                const int m = i / pairs, sl = (i - m * pairs) * 2;
                uint32_t h, l;
                mc_split2(a[u], b[u], h, l);
                *reinterpret_cast<uint32_t*>(&Ph[m * pitch + sl]) = h;
                *reinterpret_cast<uint32_t*>(&Pl[m * pitch + sl]) = l;
            }
        }
    }
}

// the same with three planes (hi / mid / lo)
template <bool LIN, class G>
__device__ __forceinline__ void mc_stage3(uint16_t* __restrict__ Ph, uint16_t* __restrict__ Pm, uint16_t* __restrict__ Pl, int pitch,
                                          int rows, int slots, G get) {
    const int pairs = slots >> 1, total = rows * pairs, nt = (int)blockDim.x;
    for (int i0 = threadIdx.x; i0 < total; i0 += 4 * nt) {
        float a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = min(i0 + u * nt, total - 1);
            const int m = i / pairs, sl = (i - m * pairs) * 2;
            a[u] = get(m, mc_slot_k<LIN>(sl));
            b[u] = get(m, mc_slot_k<LIN>(sl + 1));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * nt;
            if (i < total) {
                const int m = i / pairs, sl = (i - m * pairs) * 2;
                uint32_t h, md, l;
                mc_split3(a[u], b[u], h, md, l);
                *reinterpret_cast<uint32_t*>(&Ph[m * pitch + sl]) = h;
                *reinterpret_cast<uint32_t*>(&Pm[m * pitch + sl]) = md;
                *reinterpret_cast<uint32_t*>(&Pl[m * pitch + sl]) = l;
            }
        }
    }
}

// Six-product variant of mc_layer_b3: both operands as hi + mid + lo, products hh + hm + mh + mm + hl + lh (the dropped ml, lm,
// ll are <= 2^-24 of the product): fp32-level accuracy (1e-7) at 6 x 32 matrix cycles per 16 k against 8 x 64 for the fp32 MFMA.
template <int NT, int NU, bool LIN>
__device__ __forceinline__ void mc_layer_b6(const uint16_t* __restrict__ Ph, const uint16_t* __restrict__ Pm,
                                            const uint16_t* __restrict__ Pl, int pitch, const f32x16 (&act)[NT], f32x16 (&out)[NU],
                                            int li, int half) {
#pragma unroll
    for (int u = 0; u < NU; ++u) out[u] = zero16();
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint32_t h[4], md[4], l[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) mc_split3(act[t][8 * s + 2 * p], act[t][8 * s + 2 * p + 1], h[p], md[p], l[p]);
            const mc_bf16x8 bh = __builtin_bit_cast(mc_bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
            const mc_bf16x8 bm = __builtin_bit_cast(mc_bf16x8, make_uint4(md[0], md[1], md[2], md[3]));
            const mc_bf16x8 bl = __builtin_bit_cast(mc_bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
            const int slot0 = t * 32 + s * 16 + half * 8;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const mc_bf16x8 ah = *reinterpret_cast<const mc_bf16x8*>(&Ph[(u * 32 + li) * pitch + slot0]);
                const mc_bf16x8 am = *reinterpret_cast<const mc_bf16x8*>(&Pm[(u * 32 + li) * pitch + slot0]);
                const mc_bf16x8 al = *reinterpret_cast<const mc_bf16x8*>(&Pl[(u * 32 + li) * pitch + slot0]);
                out[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, out[u], 0, 0, 0);  // smallest terms first
                out[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, out[u], 0, 0, 0);
                out[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, out[u], 0, 0, 0);
                out[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, out[u], 0, 0, 0);
                out[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, out[u], 0, 0, 0);
                out[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, out[u], 0, 0, 0);
            }
        }
    }
}

// out[u] (u < NU tiles of 32 output rows) = W * act, act = NT tiles of 32 inputs in accumulator layout (or, LIN, one
// register set of 16 floats per half)
template <int NT, int NU, bool LIN>
__device__ __forceinline__ void mc_layer_b3(const uint16_t* __restrict__ Ph, const uint16_t* __restrict__ Pl, int pitch,
                                            const f32x16 (&act)[NT], f32x16 (&out)[NU], int li, int half) {
#pragma unroll
    for (int u = 0; u < NU; ++u) out[u] = zero16();
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint32_t h[4], l[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) mc_split2(act[t][8 * s + 2 * p], act[t][8 * s + 2 * p + 1], h[p], l[p]);
            const mc_bf16x8 bh = __builtin_bit_cast(mc_bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
            const mc_bf16x8 bl = __builtin_bit_cast(mc_bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
            const int slot0 = t * 32 + s * 16 + half * 8;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const mc_bf16x8 ah = *reinterpret_cast<const mc_bf16x8*>(&Ph[(u * 32 + li) * pitch + slot0]);
                const mc_bf16x8 al = *reinterpret_cast<const mc_bf16x8*>(&Pl[(u * 32 + li) * pitch + slot0]);
                out[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, out[u], 0, 0, 0);
                out[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, out[u], 0, 0, 0);
                out[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, out[u], 0, 0, 0);
            }
        }
    }
}

constexpr int chain_b3_lds_elems(int NH, int planes = 2) {  // bf16 elements, all planes of every matrix
    return planes * (MC_H * MC_BP32 + (NH == 2 ? MC_H * MC_BP64 : 0) + 32 * MC_BP64);
}

// SH: the colour net's input row is FORMED in the loader instead of read -- features 0..15 the degree-4 spherical harmonics of the
// sample's ray direction, features 16..16+n_geo-1 columns 1.. of the base net's output row (nerfacto_field.py:336-343: `h = cat(d, geo)`)
// -- so the [N, 32] tensor snf_head_input writes (67 MB per train step, 537 MB per 32 768-ray render chunk: 8 % of a render's kernel
// time for the write alone) never exists.  X / ldx are then the base net's output and its row length.
struct ChainSh {
    const float* dirs;  // [R, 3]
    int S;              // samples per ray: sample n belongs to ray n / S
    int log2S;          // S = 2^log2S, or -1 (a 64-bit division per lane and tile costs ~100 instructions: 128 samples per ray is a shift)
    // the base net's 16-output epilogue (MC_EPI_LIN16) also writes density[n] = exp(output 0) * selector[n] -- k_trunc_exp_fwd's
    // arithmetic on the value it would read back at a 64-byte stride (snf_mlp64_fwd_density)
    const uint8_t* dens_sel;
    float* dens;
};
static ChainSh chain_sh(const float* dirs, int S) {
    int l = -1;
    if (S > 0 && (S & (S - 1)) == 0) for (l = 0; (1 << l) < S; ++l) {}
    return ChainSh{dirs, S, l, nullptr, nullptr};
}

// this lane's half of a formed input row: half 0 = the 16 harmonics of the ray's direction, half 1 = columns 1 .. 15 of the base
// net's output row (16 floats, read as four aligned 16-byte loads) and a zero
// In two halves so that a kernel can request a row a tile ahead and form it when it is used: `raw` = the 3 direction components
// (half 0) or the 16 floats of the base net's output row (half 1).
__device__ __forceinline__ void chain_sh_load(const float* __restrict__ Hb, int ldh, const ChainSh& sh, long long s, int half,
                                              f32x16& raw) {
    if (half == 0) {
        const long long r = sh.log2S >= 0 ? (s >> sh.log2S) : s / sh.S;
        raw[0] = sh.dirs[r * 3 + 0]; raw[1] = sh.dirs[r * 3 + 1]; raw[2] = sh.dirs[r * 3 + 2];
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(Hb + s * ldh + 4 * q);
            raw[4 * q] = v.x; raw[4 * q + 1] = v.y; raw[4 * q + 2] = v.z; raw[4 * q + 3] = v.w;
        }
    }
}
__device__ __forceinline__ void chain_sh_form(const f32x16& raw, int half, f32x16& xo) {
    if (half == 0) {
        float o[16];
        sh16_of(raw[0], raw[1], raw[2], o);
#pragma unroll
        for (int i = 0; i < 16; ++i) xo[i] = o[i];
    } else {
#pragma unroll
        for (int i = 0; i < 15; ++i) xo[i] = raw[i + 1];
        xo[15] = 0.f;
    }
}
// this lane's half of a formed input row: half 0 = the 16 harmonics of the ray's direction, half 1 = columns 1 .. 15 of the base
// net's output row (16 floats, read as four aligned 16-byte loads) and a zero
__device__ __forceinline__ void chain_sh_row(const float* __restrict__ Hb, int ldh, const ChainSh& sh, long long s, int half,
                                             f32x16& xo) {
    f32x16 raw;
    chain_sh_load(Hb, ldh, sh, s, half, raw);
    chain_sh_form(raw, half, xo);
}

// The same row with ONE set of loads for all lanes (round 5; "branch-free vector memory" below): half-wave 1 reads the base net's row as
// four 16-byte loads, half-wave 0 its ray's direction as a 16-byte WINDOW of dirs (one float early for the last ray, so that the window
// stays inside the array) by the first load and the same address again by the other three (same line, no new traffic) -- no register is
// written under two exec masks, so nothing forces a wait between the request and its use a tile later.  A single ray has no 16-byte
// window inside its 12-byte direction: `od` then carries it (three scalar loads by the caller) and half-wave 0 re-reads the base net's row.
struct ChainShBf {
    long long n_rays;
    bool one_ray;
    float od[3];
};
__device__ __forceinline__ ChainShBf chain_sh_bf_init(const ChainSh& sh, long long N) {
    ChainShBf b;
    b.n_rays = sh.log2S >= 0 ? (N >> sh.log2S) : N / sh.S;
    b.one_ray = b.n_rays <= 1;
    b.od[0] = b.od[1] = b.od[2] = 0.f;
    if (b.one_ray) { b.od[0] = sh.dirs[0]; b.od[1] = sh.dirs[1]; b.od[2] = sh.dirs[2]; }
    return b;
}
__device__ __forceinline__ void chain_sh_load_bf(const float* __restrict__ Hb, int ldh, const ChainSh& sh, const ChainShBf& bf,
                                                 long long s, int half, f32x16& xo) {
    const long long r = sh.log2S >= 0 ? (s >> sh.log2S) : s / sh.S;
    const float* pr = Hb + s * ldh;
    const float* pd = bf.one_ray ? pr : sh.dirs + r * 3 - ((r == bf.n_rays - 1) ? 1 : 0);
    // (the direction window starts at a 12-byte stride: a 16-byte load of a 4-byte-aligned address -- legal for global_load_dwordx4 on
    //  gfx9, and typed as such so that the compiler does not assume 16-byte alignment; N = R * S at both entry points keeps it in bounds)
    typedef float mc_f4a4 __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float* pq = half ? pr + 4 * q : pd;
        const mc_f4a4 v = *reinterpret_cast<const mc_f4a4*>(pq);
        xo[4 * q] = v.x; xo[4 * q + 1] = v.y; xo[4 * q + 2] = v.z; xo[4 * q + 3] = v.w;
    }
}
__device__ __forceinline__ void chain_sh_form_bf(const f32x16& raw, const ChainSh& sh, const ChainShBf& bf, long long s, int half,
                                                 f32x16& xo) {
    const long long r = sh.log2S >= 0 ? (s >> sh.log2S) : s / sh.S;
    const bool shifted = r == bf.n_rays - 1;
    f32x16 rw = raw;
    if (half == 0) {  // (selects, no loads: the direction sits in [0..2] or, for the last ray, in [1..3])
        rw[0] = bf.one_ray ? bf.od[0] : shifted ? raw[1] : raw[0];
        rw[1] = bf.one_ray ? bf.od[1] : shifted ? raw[2] : raw[1];
        rw[2] = bf.one_ray ? bf.od[2] : shifted ? raw[3] : raw[2];
    }
    chain_sh_form(rw, half, xo);
}

// ---- branch-free vector memory for the tile loops (round 5) ------------------------------------------------------------------------
// A wave is alone on its SIMD in these kernels, so a memory wait is dead time, and on gfx950 loads and stores share ONE counter (vmcnt,
// in issue order).  Two things turned the "row requested a tile ahead" into a wait per tile (34 % of the wave cycles parked, round-4
// counters): (i) a load whose destination registers are also written under another exec mask (half-wave 0 = direction, half-wave 1 =
// the base net's row) makes the compiler wait for the load before the masked writes; (ii) stores under `if (ok)` / `if (o < out)` have a
// count the compiler cannot know, so the first use of the prefetched row waits for vmcnt(0) -- i.e. for the stores issued just before it.
// Here every lane issues the same loads and the same number of stores (buffer addressing: a lane with nothing to store points outside
// the descriptor's range and the hardware drops it), and the prefetched row is consumed at the BOTTOM of the iteration, behind the
// stores, where the compiler can count them: s_waitcnt vmcnt(<stores>) instead of vmcnt(0).
typedef __amdgpu_buffer_rsrc_t mc_rsrc_t;
typedef unsigned int mc_u4 __attribute__((ext_vector_type(4)));
typedef unsigned int mc_u3 __attribute__((ext_vector_type(3)));
#ifndef SNF_CHAIN_FWD_T
#define SNF_CHAIN_FWD_T 512  // threads per workgroup of the forward chains that keep their weight fragments in LDS (256 | 512)
#endif
constexpr uint32_t MC_OOR = 0x80000000u;  // a byte offset no descriptor of < 2 GB covers

__device__ __forceinline__ mc_rsrc_t mc_rsrc(const void* p, long long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(bytes > 0x7FFFFFFFLL ? 0x7FFFFFFFLL : bytes), 0x00020000);
}

// epilogue kinds of the forward chain (compile time, so that the stores of an iteration have a static count)
constexpr int MC_EPI_GENERIC = 0;   // any out / activation: per-output conditional stores (the old path)
constexpr int MC_EPI_RGB = 1;       // out == 3, sigmoid: one 12-byte store per sample (the colour net)
constexpr int MC_EPI_LIN16 = 2;     // out == 16, no activation, 16-byte aligned rows: two 16-byte stores per lane (the base net)

// LM: 0 = the input layout is a run-time value (ldx == 0: level-major, else row-major), 1 = level-major, 2 = row-major at compile time --
// two run-time paths loading into the same registers leave the compiler guessing which loads are pending at the top of the loop
// THREADS: 256, or 512 for the instantiations that read their weight fragments from LDS inside the tile loop (<= 128 registers): eight
// waves share one LDS image of the weights, two workgroups per CU = four waves per SIMD to hide each other's LDS and memory waits
template <int NH, int PLANES = 2, bool SH = false, bool HS = true, int EPI = MC_EPI_GENERIC, int LM = 0, int THREADS = 256>
// SNF_CHAIN_FWD_WAVES=2 (<= 256 registers): the two-hidden-layer six-product chain then spills 156 B per lane -- alone 0.079 ->
// 0.073 ms, but +53 MB of scratch traffic per step (PMC, r02k) in a step that is pinned by its HBM-bound kernels: not the default
#define SNF_CHAIN_FWD_WAVES 1
__global__ __launch_bounds__(THREADS, THREADS == 512 ? 4 : SNF_CHAIN_FWD_WAVES) void k_mlp_chain_fwd_b3(const float* __restrict__ X, int ldx, const float* __restrict__ W0,
                                                          int in_real, const float* __restrict__ W1,
                                                          const float* __restrict__ Wout, int out, int out_act, long long N,
                                                          float* __restrict__ H1, float* __restrict__ H2,
                                                          float* __restrict__ Y, int ldy, ChainSh sh = ChainSh{nullptr, 1, 0}) {
    extern __shared__ __attribute__((aligned(16))) uint16_t ldsb[];
    constexpr int SZ0 = MC_H * MC_BP32, SZ1 = (NH == 2 ? MC_H * MC_BP64 : 0), SZO = 32 * MC_BP64;
    uint16_t* p0h = ldsb;                          // W0  [64][MC_BP32]  LIN slots over the 32 inputs
    uint16_t* p0l = p0h + SZ0;
    uint16_t* p1h = p0l + SZ0;                     // W1  [64][MC_BP64]  (NH == 2)
    uint16_t* p1l = p1h + SZ1;
    uint16_t* poh = p1l + SZ1;                     // Wout [32][MC_BP64], rows >= out zero
    uint16_t* pol = poh + SZO;
    uint16_t* p0m = pol + SZO;                     // the mid planes (PLANES == 3) behind the hi / lo ones
    uint16_t* p1m = p0m + SZ0;
    uint16_t* pom = p1m + SZ1;
    if constexpr (PLANES == 3) {
        mc_stage3<true>(p0h, p0m, p0l, MC_BP32, MC_H, MC_IN, [&](int m, int k) { return mc_ld(W0, m * in_real + k, k < in_real); });
        if constexpr (NH == 2) mc_stage3<false>(p1h, p1m, p1l, MC_BP64, MC_H, MC_H, [&](int m, int k) { return W1[m * MC_H + k]; });
        mc_stage3<false>(poh, pom, pol, MC_BP64, 32, MC_H, [&](int m, int k) { return mc_ld(Wout, m * MC_H + k, m < out); });
    } else {
        mc_stage<true>(p0h, p0l, MC_BP32, MC_H, MC_IN, [&](int m, int k) { return mc_ld(W0, m * in_real + k, k < in_real); });
        if constexpr (NH == 2) mc_stage<false>(p1h, p1l, MC_BP64, MC_H, MC_H, [&](int m, int k) { return W1[m * MC_H + k]; });
        mc_stage<false>(poh, pol, MC_BP64, 32, MC_H, [&](int m, int k) { return mc_ld(Wout, m * MC_H + k, m < out); });
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, half = lane >> 5;
    const long long ntiles = (N + 31) / 32;
    // The NEXT tile's input row is requested before this tile's layers run and consumed (formed, masked) behind this tile's stores:
    // see "branch-free vector memory" above.  Every lane issues the same loads: rows past the batch are clamped to the last one.
    ChainShBf bf{};
    if constexpr (SH) bf = chain_sh_bf_init(sh, N);
    auto load_x = [&](long long tile_, f32x16& xo) {
        const long long s_ = tile_ * 32 + li;
        const long long sc_ = s_ < N ? s_ : N - 1;
        if constexpr (SH) {
            chain_sh_load_bf(X, ldx, sh, bf, sc_, half, xo);
        } else if (LM == 1 || (LM == 0 && ldx == 0)) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float2 v = *reinterpret_cast<const float2*>(X + ((long long)(half * 8 + q) * N + sc_) * 2);
                xo[2 * q] = v.x; xo[2 * q + 1] = v.y;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(X + sc_ * ldx + half * 16 + 4 * q);
                xo[4 * q] = v.x; xo[4 * q + 1] = v.y; xo[4 * q + 2] = v.z; xo[4 * q + 3] = v.w;
            }
        }
    };
    // raw row -> the layer's input: harmonics / geo columns (SH), pad columns zeroed
    auto form_x = [&](long long tile_, const f32x16& raw, f32x16& xo) {
        if constexpr (SH) {
            const long long s_ = tile_ * 32 + li;
            const long long sc_ = s_ < N ? s_ : N - 1;
            chain_sh_form_bf(raw, sh, bf, sc_, half, xo);
        } else {
            xo = raw;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (half * 16 + i >= in_real) xo[i] = 0.f;  // pad columns may hold anything (select, not multiply)
    };
    const long long tstride = (long long)gridDim.x * (THREADS / 64);
    const long long tile0 = (long long)blockIdx.x * (THREADS / 64) + wave;
    if (tile0 >= ntiles) return;
    const mc_rsrc_t ry = mc_rsrc(Y, N * (long long)ldy * 4);
    f32x16 x[1];
    {
        f32x16 raw0;
        load_x(tile0, raw0);
        form_x(tile0, raw0, x[0]);
    }
    for (long long tile = tile0; tile < ntiles; tile += tstride) {
        const long long s = tile * 32 + li;
        const bool ok = s < N;
        // (the last tile of a wave requests its own row again: one redundant load instead of a branch around loads)
        const long long tnext = tile + tstride < ntiles ? tile + tstride : tile;
        f32x16 rawn;
        load_x(tnext, rawn);
        // (the requests stay HERE: left to itself the scheduler sinks them to their use at the bottom of the tile to save 16 registers)
        if constexpr (EPI != MC_EPI_GENERIC) __builtin_amdgcn_sched_barrier(0);
        f32x16 h1[2];
        if constexpr (PLANES == 3) mc_layer_b6<1, 2, true>(p0h, p0m, p0l, MC_BP32, x, h1, li, half);
        else mc_layer_b3<1, 2, true>(p0h, p0l, MC_BP32, x, h1, li, half);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) h1[t][r] = fmaxf(h1[t][r], 0.f);
        if constexpr (HS) {
            if (H1 != nullptr && ok) store_h64(H1, s, h1, half);
        }
        f32x16 last[2];
        if constexpr (NH == 2) {
            if constexpr (PLANES == 3) mc_layer_b6<2, 2, false>(p1h, p1m, p1l, MC_BP64, h1, last, li, half);
            else mc_layer_b3<2, 2, false>(p1h, p1l, MC_BP64, h1, last, li, half);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) last[t][r] = fmaxf(last[t][r], 0.f);
            if constexpr (HS) {
                if (H2 != nullptr && ok) store_h64(H2, s, last, half);
            }
        } else {
            last[0] = h1[0];
            last[1] = h1[1];
        }
        f32x16 y[1];
        if constexpr (PLANES == 3) mc_layer_b6<2, 1, false>(poh, pom, pol, MC_BP64, last, y, li, half);
        else mc_layer_b3<2, 1, false>(poh, pol, MC_BP64, last, y, li, half);
        if constexpr (EPI == MC_EPI_RGB) {
            // outputs 0 .. 2 of a sample are accumulator registers 0 .. 2 of its half-wave-0 lane: one 12-byte store, dropped by the
            // address check for half-wave 1 and for rows past the batch
            float v3[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) v3[r] = 1.f / (1.f + expf(-y[0][r]));
            const mc_u3 pk = {__float_as_uint(v3[0]), __float_as_uint(v3[1]), __float_as_uint(v3[2])};
            const uint32_t off = (ok && half == 0) ? (uint32_t)(s * ldy) * 4u : MC_OOR;
            __builtin_amdgcn_raw_buffer_store_b96(pk, ry, off, 0, 0);
        } else if constexpr (EPI == MC_EPI_LIN16) {
            // accumulator registers 4 q .. 4 q + 3 of a lane are the four CONSECUTIVE outputs 8 q + 4 half .. + 3 of its sample
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const mc_u4 pk = {__float_as_uint(y[0][4 * q]), __float_as_uint(y[0][4 * q + 1]), __float_as_uint(y[0][4 * q + 2]),
                                  __float_as_uint(y[0][4 * q + 3])};
                const uint32_t off = ok ? ((uint32_t)(s * ldy) + 8u * q + 4u * half) * 4u : MC_OOR;
                __builtin_amdgcn_raw_buffer_store_b128(pk, ry, off, 0, 0);
            }
            if (sh.dens != nullptr && ok && half == 0) {  // (output 0 of a sample: register 0 of its half-wave-0 lane)
                float d = expf(y[0][0]);
                if (sh.dens_sel) d *= (float)sh.dens_sel[s];
                sh.dens[s] = d;
            }
        } else if (ok) {
            if (out_act == SNF_ACT_NONE && (out & 3) == 0 && (ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(Y) & 15) == 0) {
                // accumulator registers 4 q .. 4 q + 3 of a lane are the four CONSECUTIVE outputs 8 q + 4 half .. + 3 of its sample: one
                // 16-byte store per group instead of four scattered 4-byte ones (the base net's [N, 16] output: 2 stores per lane, not 8)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int o = 8 * q + 4 * half;
                    if (o < out)
                        *reinterpret_cast<float4*>(Y + s * ldy + o) = make_float4(y[0][4 * q], y[0][4 * q + 1], y[0][4 * q + 2], y[0][4 * q + 3]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = krow(r, half);
                    if (o < out) {
                        float v = y[0][r];
                        if (out_act == SNF_ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
                        else if (out_act == SNF_ACT_RELU) v = fmaxf(v, 0.f);
                        Y[s * ldy + o] = v;
                    }
                }
            }
        }
        form_x(tnext, rawn, x[0]);  // first use of the prefetched row: behind this tile's stores
    }
}

template <int NH>
__global__ __launch_bounds__(256) void k_mlp_chain_bwd_b3(const float* __restrict__ dY, int lddy, int dy_col_off,
                                                          const float* __restrict__ dY0, const float* __restrict__ Yout,
                                                          int ldy, const float* __restrict__ W0, int in_real,
                                                          const float* __restrict__ W1, const float* __restrict__ Wout,
                                                          int out, int out_act, long long N, const float* __restrict__ H1,
                                                          const float* __restrict__ H2, float* __restrict__ dH1,
                                                          float* __restrict__ dH2, float* __restrict__ dZout, int lddz,
                                                          float* __restrict__ dX, int lddx) {
    extern __shared__ __attribute__((aligned(16))) uint16_t ldsb[];
    uint16_t* poh = ldsb;                          // Wout^T [64 hidden][MC_BP32]  LIN slots over the (<= 32) outputs
    uint16_t* pol = poh + MC_H * MC_BP32;
    uint16_t* p1h = pol + MC_H * MC_BP32;          // W1^T   [64][MC_BP64]  (NH == 2)
    uint16_t* p1l = p1h + (NH == 2 ? MC_H * MC_BP64 : 0);
    uint16_t* p0h = p1l + (NH == 2 ? MC_H * MC_BP64 : 0);  // W0^T [32 inputs][MC_BP64], rows >= in_real zero
    uint16_t* p0l = p0h + 32 * MC_BP64;
    mc_stage<true>(poh, pol, MC_BP32, MC_H, 32, [&](int k, int o) { return mc_ld(Wout, o * MC_H + k, o < out); });
    if constexpr (NH == 2) mc_stage<false>(p1h, p1l, MC_BP64, MC_H, MC_H, [&](int k1, int k2) { return W1[k2 * MC_H + k1]; });
    mc_stage<false>(p0h, p0l, MC_BP64, 32, MC_H, [&](int i, int k1) { return mc_ld(W0, k1 * in_real + i, i < in_real); });
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, half = lane >> 5;
    const long long ntiles = (N + 31) / 32;
    for (long long tile = (long long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long long)gridDim.x * 4) {
        const long long s = tile * 32 + li;
        const bool ok = s < N;
        const long long sc = ok ? s : N - 1;
        // ---- dZ of this lane's 16 outputs (o = half*16 + i), then dLast^T[k][s] = sum_o Wout[o][k] dZ^T[o][s]
        f32x16 dzv[1];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int o = half * 16 + i;
            float dz = 0.f;
            if (o < out) {
                dz = (o == 0 && dY0 != nullptr) ? dY0[sc] : dY[sc * lddy + dy_col_off + o];
                if (out_act == SNF_ACT_SIGMOID) {
                    const float yv = Yout[sc * ldy + o];
                    dz *= yv * (1.f - yv);
                }
                if (dZout != nullptr && ok) dZout[s * lddz + o] = dz;  // pre-activation output gradient, for the wgrad GEMM
            }
            dzv[0][i] = dz;
        }
        f32x16 dl[2];
        mc_layer_b3<1, 2, true>(poh, pol, MC_BP32, dzv, dl, li, half);
        f32x16 hh[2];
        if constexpr (NH == 2) {
            load_h64(H2, sc, hh, half);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) dl[t][r] = hh[t][r] > 0.f ? dl[t][r] : 0.f;
            if (ok) store_h64(dH2, s, dl, half);
            f32x16 d1[2];
            mc_layer_b3<2, 2, false>(p1h, p1l, MC_BP64, dl, d1, li, half);
            dl[0] = d1[0];
            dl[1] = d1[1];
        }
        load_h64(H1, sc, hh, half);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) dl[t][r] = hh[t][r] > 0.f ? dl[t][r] : 0.f;
        if (ok) store_h64(dH1, s, dl, half);
        if (dX != nullptr) {
            f32x16 dxv[1];
            mc_layer_b3<2, 1, false>(p0h, p0l, MC_BP64, dl, dxv, li, half);
            const f32x16 dx = dxv[0];
            if (ok) {
                if (lddx == 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const long long lv = 4 * q + 2 * half;
                        *reinterpret_cast<float2*>(dX + (lv * N + s) * 2) = make_float2(dx[4 * q], dx[4 * q + 1]);
                        *reinterpret_cast<float2*>(dX + ((lv + 1) * N + s) * 2) = make_float2(dx[4 * q + 2], dx[4 * q + 3]);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<float4*>(dX + s * lddx + 8 * q + 4 * half) =
                            make_float4(dx[4 * q], dx[4 * q + 1], dx[4 * q + 2], dx[4 * q + 3]);
                }
            }
        }
    }
}

// ==================================================================================================================
// Data-gradient chain WITH the weight gradients (snf_mlp64_bwd_fused).
//
// k_mlp_chain_bwd leaves dW to three tall-skinny GEMMs per net that re-read what it has just written: per step and for the
// two nets 1.2 GB of hidden activations and their gradients go out to HBM and come back (rocprofv3 FETCH/WRITE_SIZE, r02f:
// chain backward 1.3 GB, 64-wide weight-gradient launches 0.74 GB + part of the 2.3 GB of the bf16x3 kernel).  Here the
// wave that holds dH^T (and has H in registers for the ReLU mask anyway) forms the products itself:
//     dW[o][i] = sum_s dA[s][o] * B[s][i]      (dA in {dZ, dH2, dH1},  B in {H2 | H1, H1, X})
// on v_mfma_f32_32x32x16_bf16 with the 3-term split (the arithmetic of k_gemm_wgrad_b3), contraction over the wave's 32
// samples.  Both operands need the SAMPLE index along k, i.e. per lane 8 consecutive samples of one neuron, while the chain
// keeps one sample per lane: each matrix takes one trip through a per-wave LDS buffer, written transposed T[neuron][sample]
// (ds_write_b32, conflict-free) and read back as two ds_read_b128 per fragment (pitch 36 floats: 16 lanes hit 16 distinct
// bank quads).  A wave's LDS traffic is ordered, so no barrier is needed; the A-side fragments are held in registers while
// the B-side matrix reuses the buffer.  The running sums (8 accumulator tiles for the colour net) stay in registers over the
// wave's whole tile loop; at the end the 8 waves of a workgroup fold them through LDS and the workgroup writes ONE partial
// [Wout | W1 | W0] to the workspace; k_chain_wgrad_reduce adds the partials to the gradient arena (no float atomics).
// 256 threads, one wave per SIMD (the 8 accumulator tiles on top of the chain's working set need the whole 512-entry register
// file: at two waves per SIMD the compiler spills 170 registers per lane), one workgroup per CU; LDS: weights 33 KB + 4 x
// (9 KB staging + 4.5 KB dZ^T).
constexpr int WG_TP = 36;            // floats per transposed row: 32 samples + 4 pad
constexpr int WG_T = 256;
constexpr int WG_WAVES = WG_T / 64;
constexpr int WG_STAGE = 64 * WG_TP;  // one staged 64-neuron matrix
constexpr int WG_DZ = 32 * WG_TP;     // the (zero-padded) output-gradient matrix

__device__ __forceinline__ void wg_frag(const float* __restrict__ T, int row, int ks, int half, mc_bf16x8& hi, mc_bf16x8& lo) {
    const float4 v0 = *reinterpret_cast<const float4*>(&T[row * WG_TP + 16 * ks + 8 * half]);
    const float4 v1 = *reinterpret_cast<const float4*>(&T[row * WG_TP + 16 * ks + 8 * half + 4]);
    uint32_t h[4], l[4];
    mc_split2(v0.x, v0.y, h[0], l[0]);
    mc_split2(v0.z, v0.w, h[1], l[1]);
    mc_split2(v1.x, v1.y, h[2], l[2]);
    mc_split2(v1.z, v1.w, h[3], l[3]);
    hi = __builtin_bit_cast(mc_bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
    lo = __builtin_bit_cast(mc_bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
}

// a 64-wide accumulator-layout tile pair (lane = sample, registers = neurons) -> T[neuron][sample]
__device__ __forceinline__ void wg_put64(float* __restrict__ T, const f32x16 (&a)[2], int li, int half) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) T[(t * 32 + krow(r, half)) * WG_TP + li] = a[t][r];
}

__device__ __forceinline__ f32x16 wg_mma3(const mc_bf16x8& ah, const mc_bf16x8& al, const mc_bf16x8& bh, const mc_bf16x8& bl,
                                          f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
}

#define WG_LDS_ORDER() asm volatile("" ::: "memory")  // compiler fence between the staging and the fragment reads of a wave

// partial layout per workgroup (floats): Wout [32][64] | W1 [64][64] (NH == 2) | W0 [64][32]
constexpr int wg_partial_floats(int NH) { return 32 * 64 + (NH == 2 ? 64 * 64 : 0) + 64 * 32; }

// RC ("recompute"): H1 / H2 are NOT read -- the hidden activations are formed again from X with the forward's own arithmetic
// (mc_layer_b6 on the same three-plane weights: bit-identical to what k_mlp_chain_fwd_b3<NH, 3> computed, so the ReLU masks and
// the weight-gradient operands are the forward's), and the forward does not have to write them: 0.4 GB less written and 0.4 GB
// less read per step for the two field nets.  The two-hidden-layer net forms H1 twice (once on the way to H2, once when it is
// needed itself) rather than keeping 32 more registers alive.
#ifndef SNF_WG_DGRAD_B3
#define SNF_WG_DGRAD_B3 1
#endif
#ifndef SNF_WG_DGRAD_B3_NH1
#define SNF_WG_DGRAD_B3_NH1 0
#endif
// weights of the data-gradient chain in LDS (floats): transposed bf16 hi / lo planes (DG3) or the fp32 matrices
constexpr int wg_weight_lds_floats(int NH) {
    return (SNF_WG_DGRAD_B3 && (NH == 2 || SNF_WG_DGRAD_B3_NH1)) ? (MC_H * MC_BP32 + (NH == 2 ? MC_H * MC_BP64 : 0) + 32 * MC_BP64)
                                                                 : chain_lds_floats(NH);
}
constexpr int wg_rc_lds_elems(int NH) { return 3 * (MC_H * MC_BP32 + (NH == 2 ? MC_H * MC_BP64 : 0)); }

template <int NH, bool RC = false, bool SH = false>
// one hidden layer: 2 waves per SIMD = 252 registers instead of 280, no spills -- a workgroup then fits beside one workgroup of
// the table reduce on a CU (co-residency); alone 0.129 -> 0.126 ms
#ifndef SNF_WG_WAVES_NH1
#define SNF_WG_WAVES_NH1 2
#endif
__global__ __launch_bounds__(WG_T) __attribute__((amdgpu_waves_per_eu(NH == 1 ? SNF_WG_WAVES_NH1 : 1, NH == 1 ? SNF_WG_WAVES_NH1 : 1))) void k_mlp_chain_bwd_wg(const float* __restrict__ dY, int lddy, int dy_col_off,
                                                           const float* __restrict__ dY0, const float* __restrict__ Yout,
                                                           int ldy, const float* __restrict__ X, int ldx,
                                                           const float* __restrict__ W0, int in_real,
                                                           const float* __restrict__ W1, const float* __restrict__ Wout,
                                                           int out, int out_act, long long N, const float* __restrict__ H1,
                                                           const float* __restrict__ H2, float* __restrict__ dX, int lddx,
                                                           float* __restrict__ P, ChainSh sh = ChainSh{nullptr, 1, 0}) {
    static_assert(!SH || RC, "the formed input row (SH) goes with the recomputing backward");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // Data-gradient chain on the bf16 matrix cores with the 3-term split (SNF_WG_DGRAD_B3, round 4): the transposed weights as hi / lo
    // planes with the accumulator layout's k permutation (as k_mlp_chain_bwd_b3 stages them), 48 v_mfma_f32_32x32x16_bf16 per tile
    // instead of 100 v_mfma_f32_32x32x2_f32 (1536 against 6400 matrix cycles for the colour net) and two ds_read_b128 per k-step
    // instead of one ds_read_b32 per instruction.  The gradients' consumers (the weight gradients below, the table backward) already
    // work on 3-term products.  0: the fp32 chain.
    // (two hidden layers only: with one hidden layer and up to 16 outputs the fp32 chain is 48 short instructions, and the split's
    //  registers cost the kernel its second wave per SIMD -- measured 0.123 -> 0.134 ms for the base net, 0.224 -> 0.168 for the colour net)
    constexpr bool DG3 = SNF_WG_DGRAD_B3 != 0 && (NH == 2 || SNF_WG_DGRAD_B3_NH1);
    ChainWeights cw;
    uint16_t* poh = reinterpret_cast<uint16_t*>(lds);      // Wout^T [64 hidden][MC_BP32]  LIN slots over the (<= 32) outputs
    uint16_t* pol = poh + MC_H * MC_BP32;
    uint16_t* p1h = pol + MC_H * MC_BP32;                  // W1^T   [64][MC_BP64]  (NH == 2)
    uint16_t* p1l = p1h + (NH == 2 ? MC_H * MC_BP64 : 0);
    uint16_t* p0h = p1l + (NH == 2 ? MC_H * MC_BP64 : 0);  // W0^T   [32 inputs][MC_BP64], rows >= in_real zero
    uint16_t* p0l = p0h + 32 * MC_BP64;
    if constexpr (DG3) {
        mc_stage<true>(poh, pol, MC_BP32, MC_H, 32, [&](int k_, int o) { return mc_ld(Wout, o * MC_H + k_, o < out); });
        if constexpr (NH == 2) mc_stage<false>(p1h, p1l, MC_BP64, MC_H, MC_H, [&](int k1, int k2) { return W1[k2 * MC_H + k1]; });
        mc_stage<false>(p0h, p0l, MC_BP64, 32, MC_H, [&](int i, int k1) { return mc_ld(W0, k1 * in_real + i, i < in_real); });
        __syncthreads();
    } else {
        load_chain_weights<NH>(lds, cw, W0, in_real, W1, Wout, out);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, half = lane >> 5;
    float* __restrict__ stage_all = lds + wg_weight_lds_floats(NH);       // [WG_WAVES][WG_STAGE]
    float* __restrict__ T = stage_all + wave * WG_STAGE;
    float* __restrict__ Z = stage_all + WG_WAVES * WG_STAGE + wave * WG_DZ;  // dZ^T, rows >= out stay zero
    for (int i = lane; i < WG_DZ; i += 64) Z[i] = 0.f;
    // RC: W0 (LIN slots) and W1 as hi / mid / lo bf16 planes behind the staging area, exactly as the forward kernel stages them
    constexpr int RS0 = MC_H * MC_BP32, RS1 = NH == 2 ? MC_H * MC_BP64 : 0;
    uint16_t* __restrict__ r0h = reinterpret_cast<uint16_t*>(stage_all + WG_WAVES * (WG_STAGE + WG_DZ));
    uint16_t* __restrict__ r0m = r0h + RS0;
    uint16_t* __restrict__ r0l = r0m + RS0;
    uint16_t* __restrict__ r1h = r0l + RS0;
    uint16_t* __restrict__ r1m = r1h + RS1;
    uint16_t* __restrict__ r1l = r1m + RS1;
    if constexpr (RC) {
        mc_stage3<true>(r0h, r0m, r0l, MC_BP32, MC_H, MC_IN, [&](int m, int k) { return mc_ld(W0, m * in_real + k, k < in_real); });
        if constexpr (NH == 2) mc_stage3<false>(r1h, r1m, r1l, MC_BP64, MC_H, MC_H, [&](int m, int k) { return W1[m * MC_H + k]; });
        __syncthreads();
    }
    const int outp = (out + 1) & ~1;
    const int hsteps = outp >> 1;
    constexpr int NT = NH == 2 ? 8 : 4;
    f32x16 acc[NT];  // NH == 2: [0,1] Wout (k tiles), [2..5] W1 (u*2 + t), [6,7] W0 (u) ; NH == 1: [0,1] Wout, [2,3] W0
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = zero16();
    const long long ntiles = (N + 31) / 32;
    // (SNF_CHAIN_PREFETCH, RC only: the next tile's input row and output gradients are requested while this tile is worked on --
    //  the wave is alone on its SIMD and would otherwise sit out both latencies at the top of every tile)
    ChainShBf bfb{};
    if constexpr (SH && SNF_WG_BF != 0) bfb = chain_sh_bf_init(sh, N);
    auto load_x = [&](long long tile_, f32x16& xo) {
        const long long s_ = tile_ * 32 + li;
        const long long sc_ = s_ < N ? s_ : N - 1;
        if constexpr (SH) {
            if constexpr (SNF_WG_BF != 0) chain_sh_load_bf(X, ldx, sh, bfb, sc_, half, xo);  // (raw; formed below)
            else chain_sh_load(X, ldx, sh, sc_, half, xo);
        } else if (ldx == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float2 v = *reinterpret_cast<const float2*>(X + ((long long)(half * 8 + q) * N + sc_) * 2);
                xo[2 * q] = v.x; xo[2 * q + 1] = v.y;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(X + sc_ * ldx + half * 16 + 4 * q);
                xo[4 * q] = v.x; xo[4 * q + 1] = v.y; xo[4 * q + 2] = v.z; xo[4 * q + 3] = v.w;
            }
        }
    };
    const long long tstride = (long long)gridDim.x * WG_WAVES;
    const long long tile0 = (long long)blockIdx.x * WG_WAVES + wave;
    // The wave is alone on its SIMD: the NEXT tile's input row and (nets with <= 4 outputs: the colour net) output gradients are
    // requested while this tile is worked on.  Affordable since the data-gradient chain left the fp32 matrix instruction (428 of 512
    // registers instead of 491; before, the 16-register row spilled 48).
    constexpr bool PFB = RC && NH == 2 && DG3 && (SNF_CHAIN_PREFETCH & 2) != 0;
    // Pre-activation output gradients of sample (tile_, li) for the outputs o_of(0 .. cnt-1) of this lane (cnt wave-uniform, <= 16): ALL
    // loads first -- unconditional, the output index clamped into the row, the density column's separate tensor by a pointer select --
    // then the sigmoid derivative and the bounds.  One guarded load per output with its use behind it compiled to one basic block and one
    // s_waitcnt vmcnt(0) per output: 8 (colour net: 4 gradients + 4 activations) to 16 (base net) memory latencies in a row at the top of
    // every 32-sample tile, in a wave that is alone on its SIMD.
    // (GB outputs per call: with sixteen in flight the one-hidden-layer instances -- two waves per SIMD, 256 registers -- spilled 30-70
    //  registers, with eight still 11-41; four there, eight with two hidden layers)
    constexpr int GB = NH == 1 ? 4 : 8;
    auto dz_gather = [&](long long tile_, int i0, int cnt, auto o_of, float (&dz)[GB]) {
        const long long s_ = tile_ * 32 + li;
        const bool ok_ = s_ < N;
        const long long sc_ = ok_ ? s_ : N - 1;
        float yr[GB];
#pragma unroll
        for (int j = 0; j < GB; ++j) {
            dz[j] = 0.f;
            yr[j] = 0.f;
            if (i0 + j < cnt) {  // wave-uniform
                const int o = o_of(i0 + j), oc = o < out ? o : out - 1;
                const float* __restrict__ src = (o == 0 && dY0 != nullptr) ? dY0 + sc_ : dY + (sc_ * lddy + dy_col_off + oc);
                dz[j] = *src;
            }
        }
        if (out_act == SNF_ACT_SIGMOID) {
#pragma unroll
            for (int j = 0; j < GB; ++j) {
                if (i0 + j < cnt) {
                    const int o = o_of(i0 + j), oc = o < out ? o : out - 1;
                    yr[j] = Yout[sc_ * ldy + oc];
                }
            }
#pragma unroll
            for (int j = 0; j < GB; ++j) dz[j] *= yr[j] * (1.f - yr[j]);
        }
#pragma unroll
        for (int j = 0; j < GB; ++j) {
            const bool live = i0 + j < cnt && o_of(i0 + j) < out && ok_;  // rows beyond N contribute nothing to the weight gradients
            dz[j] = live ? dz[j] : 0.f;
        }
    };
    f32x16 xn;
    if (PFB && tile0 < ntiles) load_x(tile0, xn);
    for (long long tile = tile0; tile < ntiles; tile += tstride) {
        const long long s = tile * 32 + li;
        const bool ok = s < N;
        const long long sc = ok ? s : N - 1;
        f32x16 xr[1];  // RC: this lane's half of the input row (features half*16 .. +15), pad columns zero as in the forward
        if constexpr (RC) {
            if (PFB) {
                xr[0] = xn;
                if (tile + tstride < ntiles) load_x(tile + tstride, xn);
            } else {
                load_x(tile, xr[0]);
            }
            if constexpr (SH) {
                const f32x16 raw = xr[0];
                if constexpr (SNF_WG_BF != 0) chain_sh_form_bf(raw, sh, bfb, sc, half, xr[0]);
                else chain_sh_form(raw, half, xr[0]);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (half * 16 + i >= in_real) xr[0][i] = 0.f;
        }
        auto hidden1 = [&](f32x16 (&h)[2]) {  // H1^T tile pair of this wave's 32 samples, ReLU applied
            mc_layer_b6<1, 2, true>(r0h, r0m, r0l, MC_BP32, xr, h, li, half);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) h[t][r] = fmaxf(h[t][r], 0.f);
        };
        // ---- dZ, dLast^T[k][s] = sum_o Wout[o][k] dZ^T[o][s]; dZ^T goes to its LDS matrix on the way
        f32x16 dl[2] = {zero16(), zero16()};
        if constexpr (DG3) {
            f32x16 dzv[1];  // this lane's 16 outputs o = half * 16 + i (LIN k slots)
            const int cnt = outp < 16 ? outp : 16;
#pragma unroll
            for (int g = 0; g < 16 / GB; ++g) {
                float dzr[GB];
                if (GB * g < cnt) {  // wave-uniform
                    dz_gather(tile, GB * g, cnt, [&](int i) { return half * 16 + i; }, dzr);
                } else {
#pragma unroll
                    for (int j = 0; j < GB; ++j) dzr[j] = 0.f;
                }
#pragma unroll
                for (int j = 0; j < GB; ++j) {
                    const int o = half * 16 + GB * g + j;
                    dzv[0][GB * g + j] = dzr[j];
                    if (o < outp) Z[o * WG_TP + li] = dzr[j];
                }
            }
            mc_layer_b3<1, 2, true>(poh, pol, MC_BP32, dzv, dl, li, half);
        } else {
#pragma unroll 1
            for (int g = 0; g < 16 / GB; ++g) {  // (a real loop: unrolled, the sixteen steps' weight reads were hoisted and spilled)
                if (GB * g < hsteps) {  // wave-uniform
                    float dzr[GB];
                    dz_gather(tile, GB * g, hsteps, [&](int st) { return half * hsteps + st; }, dzr);
#pragma unroll
                    for (int j = 0; j < GB; ++j) {
                        const int st = GB * g + j;
                        if (st < hsteps) {  // wave-uniform
                            const int o = half * hsteps + st;
                            const float dz = dzr[j];
                            if (o < 32) Z[o * WG_TP + li] = dz;
                            const int oc = o < 32 ? o : 31;
                            dl[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cw.wo[oc * MC_P1 + li], dz, dl[0], 0, 0, 0);
                            dl[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cw.wo[oc * MC_P1 + 32 + li], dz, dl[1], 0, 0, 0);
                        }
                    }
                }
            }
        }
        WG_LDS_ORDER();
        mc_bf16x8 zh[2], zl[2];  // A side of dWout: dZ^T rows 0..31, k-steps 0, 1
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wg_frag(Z, li, ks, half, zh[ks], zl[ks]);
        f32x16 hh[2];
        // ---- last hidden layer: its activations are the B side of dWout and the ReLU mask of dLast
        if constexpr (RC) {
            if constexpr (NH == 2) {
                f32x16 h1t[2];
                hidden1(h1t);
                mc_layer_b6<2, 2, false>(r1h, r1m, r1l, MC_BP64, h1t, hh, li, half);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) hh[t][r] = fmaxf(hh[t][r], 0.f);
            } else {
                hidden1(hh);
            }
        } else {
            load_h64(NH == 2 ? H2 : H1, sc, hh, half);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) dl[t][r] = hh[t][r] > 0.f ? dl[t][r] : 0.f;
        wg_put64(T, hh, li, half);  // (the activations are dead after this: registers free for the products below)
        WG_LDS_ORDER();
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                mc_bf16x8 bh, bl;
                wg_frag(T, 32 * t + li, ks, half, bh, bl);
                acc[t] = wg_mma3(zh[ks], zl[ks], bh, bl, acc[t]);
            }
        if constexpr (NH == 2) {
            // dH1^T[k1][s] = sum_k2 W1[k2][k1] dH2^T[k2][s]
            f32x16 d1[2] = {zero16(), zero16()};
            if constexpr (DG3) {
                mc_layer_b3<2, 2, false>(p1h, p1l, MC_BP64, dl, d1, li, half);
            } else {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int k2 = u * 32 + krow(r, half);
                        const float b = dl[u][r];
                        d1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cw.w1[k2 * MC_P1 + li], b, d1[0], 0, 0, 0);
                        d1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cw.w1[k2 * MC_P1 + 32 + li], b, d1[1], 0, 0, 0);
                    }
            }
            // dW1 = dH2^T (A side, staged now) x H1 (B side, staged after the A fragments are in registers)
            WG_LDS_ORDER();
            wg_put64(T, dl, li, half);
            WG_LDS_ORDER();
            mc_bf16x8 ah[2][2], al[2][2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) wg_frag(T, 32 * u + li, ks, half, ah[u][ks], al[u][ks]);
            if constexpr (RC) hidden1(hh);
            else load_h64(H1, sc, hh, half);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) dl[t][r] = hh[t][r] > 0.f ? d1[t][r] : 0.f;  // dl is dH1^T from here on
            WG_LDS_ORDER();
            wg_put64(T, hh, li, half);
            WG_LDS_ORDER();
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    mc_bf16x8 bh, bl;
                    wg_frag(T, 32 * t + li, ks, half, bh, bl);
#pragma unroll
                    for (int u = 0; u < 2; ++u) acc[2 + u * 2 + t] = wg_mma3(ah[u][ks], al[u][ks], bh, bl, acc[2 + u * 2 + t]);
                }
        }
        // ---- dl is dH1^T now.  dX^T[i][s] = sum_k1 W0[k1][i] dH1^T[k1][s]
        if (dX != nullptr) {
            f32x16 dx = zero16();
            if constexpr (DG3) {
                f32x16 dxv[1];
                mc_layer_b3<2, 1, false>(p0h, p0l, MC_BP64, dl, dxv, li, half);
                dx = dxv[0];
            } else {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        dx = __builtin_amdgcn_mfma_f32_32x32x2f32(cw.w0[(t * 32 + krow(r, half)) * MC_P0 + li], dl[t][r], dx, 0,
                                                                  0, 0);
            }
            if (ok) {
                if constexpr (SH) {
                    // only the base net's share of the input has a consumer (the harmonics depend on the ray direction alone): the
                    // gradient of features 16 .. 31 goes out as a compact [N, lddx >= 16] row, column j = feature 16 + j
#pragma unroll
                    for (int q = 2; q < 4; ++q)
                        *reinterpret_cast<float4*>(dX + s * lddx + 8 * q + 4 * half - 16) =
                            make_float4(dx[4 * q], dx[4 * q + 1], dx[4 * q + 2], dx[4 * q + 3]);
                } else if (lddx == 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const long long lv = 4 * q + 2 * half;
                        *reinterpret_cast<float2*>(dX + (lv * N + s) * 2) = make_float2(dx[4 * q], dx[4 * q + 1]);
                        *reinterpret_cast<float2*>(dX + ((lv + 1) * N + s) * 2) = make_float2(dx[4 * q + 2], dx[4 * q + 3]);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<float4*>(dX + s * lddx + 8 * q + 4 * half) =
                            make_float4(dx[4 * q], dx[4 * q + 1], dx[4 * q + 2], dx[4 * q + 3]);
                }
            }
        }
        // ---- dW0 = dH1^T (A side) x X (B side: this lane's half of the input row, features half*16 .. +15)
        WG_LDS_ORDER();
        wg_put64(T, dl, li, half);
        WG_LDS_ORDER();
        mc_bf16x8 a0h[2][2], a0l[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) wg_frag(T, 32 * u + li, ks, half, a0h[u][ks], a0l[u][ks]);
        float x[16];
        if constexpr (RC) {
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = xr[0][i];
        } else if (ldx == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float2 v = *reinterpret_cast<const float2*>(X + ((long long)(half * 8 + q) * N + sc) * 2);
                x[2 * q] = v.x; x[2 * q + 1] = v.y;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(X + sc * ldx + half * 16 + 4 * q);
                x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
            }
        }
        WG_LDS_ORDER();
#pragma unroll
        for (int i = 0; i < 16; ++i) T[(half * 16 + i) * WG_TP + li] = (half * 16 + i < in_real) ? x[i] : 0.f;
        WG_LDS_ORDER();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            mc_bf16x8 bh, bl;
            wg_frag(T, li, ks, half, bh, bl);
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[NT - 2 + u] = wg_mma3(a0h[u][ks], a0l[u][ks], bh, bl, acc[NT - 2 + u]);
        }
        WG_LDS_ORDER();
    }
    // ---- fold the 8 waves' sums through LDS (two tiles per round: 7 x 8 KB in the staging area), wave 0 collects
    __syncthreads();
    float* __restrict__ R = stage_all;  // [wave - 1][2][16][64]
#pragma unroll
    for (int q = 0; q < NT / 2; ++q) {
        if (wave > 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) R[((wave - 1) * 2 + j) * 1024 + r * 64 + lane] = acc[2 * q + j][r];
        }
        __syncthreads();
        if (wave == 0) {
            for (int w2 = 0; w2 < WG_WAVES - 1; ++w2)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[2 * q + j][r] += R[(w2 * 2 + j) * 1024 + r * 64 + lane];
        }
        __syncthreads();
    }
    if (wave == 0) {
        // accumulator tile (m-tile u, n-tile t): element (row = 32u + krow(r, half), col = 32t + li)
        float* __restrict__ Pw = P + (size_t)blockIdx.x * wg_partial_floats(NH);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) Pw[krow(r, half) * 64 + 32 * t + li] = acc[t][r];  // Wout [32][64]
        float* __restrict__ P1 = Pw + 32 * 64;
        if constexpr (NH == 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) P1[(32 * u + krow(r, half)) * 64 + 32 * t + li] = acc[2 + u * 2 + t][r];
            P1 += 64 * 64;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) P1[(32 * u + krow(r, half)) * 32 + li] = acc[NT - 2 + u][r];  // W0 [64][32]
    }
}

// dWout[o][k] (o < out), dW1[k2][k1], dW0[k1][i] (i < in_real) += sum over the workgroup partials.
// 32 outputs x 8 slices of the partial list per workgroup (a thread sums nwg/8 partials, the 8 slices meet in LDS): 256
// workgroups with 8 independent load chains each instead of 32 workgroups walking all 256 partials in sequence (49 us
// under load for 8 MB); the summation tree is fixed, so the result is deterministic.
__global__ __launch_bounds__(256) void k_chain_wgrad_reduce(const float* __restrict__ P, int nwg, int NH, int out, int in_real,
                                                            float* __restrict__ dWout, float* __restrict__ dW1,
                                                            float* __restrict__ dW0) {
    __shared__ float part[8][32];
    const int per = 32 * 64 + (NH == 2 ? 64 * 64 : 0) + 64 * 32;
    const int ol = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + ol;  // per is a multiple of 32
    const int w0 = (int)((long long)nwg * slice / 8), w1 = (int)((long long)nwg * (slice + 1) / 8);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int w = w0;
    for (; w + 4 <= w1; w += 4) {
        s0 += P[(size_t)w * per + e];
        s1 += P[(size_t)(w + 1) * per + e];
        s2 += P[(size_t)(w + 2) * per + e];
        s3 += P[(size_t)(w + 3) * per + e];
    }
    for (; w < w1; ++w) s0 += P[(size_t)w * per + e];
    part[slice][ol] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (slice != 0) return;
    const float sum = ((part[0][ol] + part[1][ol]) + (part[2][ol] + part[3][ol])) +
                      ((part[4][ol] + part[5][ol]) + (part[6][ol] + part[7][ol]));
    if (e < 32 * 64) {
        const int o = e / 64, k = e % 64;
        if (o < out) dWout[o * 64 + k] += sum;
        return;
    }
    int r = e - 32 * 64;
    if (NH == 2) {
        if (r < 64 * 64) {
            dW1[r] += sum;
            return;
        }
        r -= 64 * 64;
    }
    const int k1 = r / 32, i = r % 32;
    if (i < in_real) dW0[k1 * in_real + i] += sum;
}

// opt-in (snf_set_gemm_mode(2)): measured on the train step's two nets against the fp32 chains -- colour net forward
// 0.131 -> 0.087 ms, base net forward 0.104 -> 0.090, colour net backward 0.189 -> 0.168, base net backward 0.130 -> 0.161
// (its chain starts from <= 16 output gradients padded to a 32-wide k-step), step time -1 %: with the matrix work cut 5x the
// chains are bound by their hidden-activation traffic, and the 6x larger round-off is not worth 1 %.
static bool chain_b3_on() { return snf_get_gemm_mode() == 2; }

}  // namespace snf

using namespace snf;

static int chain_common_checks(const char* who, int in_real, int n_hidden, int out, long long N) {
    SNF_REQUIRE(in_real >= 1 && in_real <= MC_IN, "%s: input width %d not in [1,32]", who, in_real);
    SNF_REQUIRE(n_hidden == 1 || n_hidden == 2, "%s: 1 or 2 hidden layers of width 64 (got %d)", who, n_hidden);
    SNF_REQUIRE(out >= 1 && out <= 32, "%s: output width %d not in [1,32]", who, out);
    SNF_REQUIRE(N > 0, "%s: empty batch", who);
    return SNF_OK;
}

// the colour net with its input row formed in the loader (ChainSh): X = the base net's output [N, ldx >= 16], in_real = 16 + n_geo
static int chain_fwd_sh(const float* dirs, int R, int S, const float* Hb, int ldh, int n_geo, const float* W0, const float* W1,
                        const float* Wout, int n_hidden, int out, int out_act, float* H1, float* H2, float* Y, int ldy,
                        snf_stream_t stream) {
    const long long N = (long long)R * S;
    int rc = chain_common_checks("snf_mlp64_fwd_sh", 16 + n_geo, n_hidden, out, N);
    if (rc) return rc;
    SNF_REQUIRE(dirs && Hb && W0 && Wout && Y && (n_hidden == 1 || W1), "snf_mlp64_fwd_sh: null pointer");
    SNF_REQUIRE(R > 0 && S > 0 && n_geo == 15 && ldh >= 16 && ldh % 4 == 0 && ((uintptr_t)Hb % 16) == 0,
                "snf_mlp64_fwd_sh: the base net's output must be [N, ldh >= 16] (ldh %% 4 == 0, 16-byte aligned) with the 15 geo "
                "features in columns 1 .. 15 (got n_geo=%d ldh=%d)", n_geo, ldh);
    SNF_REQUIRE(snf_get_gemm_mode() == 1, "snf_mlp64_fwd_sh: gemm mode 1 only (the six-product forward chain)");
    SNF_REQUIRE(ldy >= out, "snf_mlp64_fwd_sh: ldy < out");
    SNF_REQUIRE((!H1 || ((uintptr_t)H1 % 16) == 0) && (!H2 || ((uintptr_t)H2 % 16) == 0), "snf_mlp64_fwd_sh: unaligned H");
    const long long ntiles = (N + 31) / 32;
    long long blocks = (ntiles + 3) / 4;
    if (blocks > 256 * 4) blocks = 256 * 4;
    const ChainSh sh = chain_sh(dirs, S);
    // the colour net as the step and the render run it (no stored activations, rgb = sigmoid of 3 outputs): the instantiation whose
    // stores have a static count (k_mlp_chain_fwd_b3, "branch-free vector memory"); everything else takes the general epilogue
    const bool rgb = n_hidden == 2 && !H1 && !H2 && out == 3 && out_act == SNF_ACT_SIGMOID && N * (long long)ldy * 4 < 0x7FFFFFFFLL;
    if (rgb) {
        constexpr int T = SNF_CHAIN_FWD_T;
        long long b2 = (ntiles + T / 64 - 1) / (T / 64);
        const long long cap = 256LL * (T == 512 ? 2 : 4);
        if (b2 > cap) b2 = cap;
        hipLaunchKernelGGL((k_mlp_chain_fwd_b3<2, 3, true, false, MC_EPI_RGB, 0, T>), dim3((unsigned)b2), dim3(T),
                           chain_b3_lds_elems(2, 3) * sizeof(uint16_t), (hipStream_t)stream, Hb, ldh, W0, 16 + n_geo, W1, Wout, out, out_act, N,
                           H1, H2, Y, ldy, sh);
    }
    else if (n_hidden == 2)
        hipLaunchKernelGGL((k_mlp_chain_fwd_b3<2, 3, true>), dim3((unsigned)blocks), dim3(256), chain_b3_lds_elems(2, 3) * sizeof(uint16_t),
                           (hipStream_t)stream, Hb, ldh, W0, 16 + n_geo, W1, Wout, out, out_act, N, H1, H2, Y, ldy, sh);
    else
        hipLaunchKernelGGL((k_mlp_chain_fwd_b3<1, 3, true>), dim3((unsigned)blocks), dim3(256), chain_b3_lds_elems(1, 3) * sizeof(uint16_t),
                           (hipStream_t)stream, Hb, ldh, W0, 16 + n_geo, W1, Wout, out, out_act, N, H1, H2, Y, ldy, sh);
    SNF_LAUNCH_CHECK("snf_mlp64_fwd_sh");
    return SNF_OK;
}

extern "C" int snf_mlp64_fwd_sh(const float* dirs, int R, int S, const float* base_out, int ld_base, int n_geo, const float* W0,
                                const float* W1, const float* Wout, int n_hidden, int out, int out_act, float* H1, float* H2,
                                float* Y, int ldy, snf_stream_t stream) {
    return chain_fwd_sh(dirs, R, S, base_out, ld_base, n_geo, W0, W1, Wout, n_hidden, out, out_act, H1, H2, Y, ldy, stream);
}

extern "C" int snf_trunc_exp_fwd(const float* raw, int raw_stride, const uint8_t* selector, int64_t N, float* density,
                                 snf_stream_t stream);

// dens != nullptr: density[n] = exp(Y[n][0]) * selector[n] as well -- inside the base net's epilogue where that one exists, by
// snf_trunc_exp_fwd behind the chain otherwise
static int chain_fwd_plain(const float* X, int ldx, const float* W0, int in_real, const float* W1, const float* Wout,
                           int n_hidden, int out, int out_act, int64_t N, float* H1, float* H2, float* Y, int ldy,
                           const uint8_t* dens_sel, float* dens, snf_stream_t stream) {
    int rc = chain_common_checks("snf_mlp64_fwd", in_real, n_hidden, out, N);
    if (rc) return rc;
    SNF_REQUIRE(X && W0 && Wout && Y && (n_hidden == 1 || W1), "snf_mlp64_fwd: null pointer");
    SNF_REQUIRE((ldx == 0 || (ldx >= MC_IN && ldx % 4 == 0)) && ((uintptr_t)X % 16) == 0,
                "snf_mlp64_fwd: X must be [N, ldx>=32] with ldx %% 4 == 0 and 16-byte aligned rows (pad columns readable)");
    SNF_REQUIRE(ldy >= out, "snf_mlp64_fwd: ldy < out");
    SNF_REQUIRE((!H1 || ((uintptr_t)H1 % 16) == 0) && (!H2 || ((uintptr_t)H2 % 16) == 0), "snf_mlp64_fwd: unaligned H");
    bool dens_done = false;
    const long long ntiles = (N + 31) / 32;
    long long blocks = (ntiles + 3) / 4;
    if (blocks > 256 * 4) blocks = 256 * 4;  // persistent: <= 4 workgroups per CU
    // gemm mode 1 (default): the forward chains on the six-product bf16 split (fp32-level accuracy, 2.7x less matrix time than the
    // fp32 MFMA); SNF_CHAIN_FWD_X6=0 keeps the fp32 MFMA there.  Mode 0: fp32 MFMA.  Mode 2: the three-product split.
    static const int x6 = 1;
    if (chain_b3_on()) {
        if (n_hidden == 2)
            hipLaunchKernelGGL(k_mlp_chain_fwd_b3<2>, dim3((unsigned)blocks), dim3(256), chain_b3_lds_elems(2) * sizeof(uint16_t),
                               (hipStream_t)stream, X, ldx, W0, in_real, W1, Wout, out, out_act, (long long)N, H1, H2, Y, ldy, ChainSh{nullptr, 1, 0});
        else
            hipLaunchKernelGGL(k_mlp_chain_fwd_b3<1>, dim3((unsigned)blocks), dim3(256), chain_b3_lds_elems(1) * sizeof(uint16_t),
                               (hipStream_t)stream, X, ldx, W0, in_real, W1, Wout, out, out_act, (long long)N, H1, H2, Y, ldy, ChainSh{nullptr, 1, 0});
    } else if (x6 && snf_get_gemm_mode() == 1) {
        // (the base net as the step and the render run it: 16 linear outputs, nothing stored but them)
        const bool lin16 = n_hidden == 1 && !H1 && !H2 && out == 16 && out_act == SNF_ACT_NONE && (ldy & 3) == 0 &&
                           ((uintptr_t)Y & 15) == 0 && N * (long long)ldy * 4 < 0x7FFFFFFFLL;
        constexpr int T = SNF_CHAIN_FWD_T;
        long long b2 = (ntiles + T / 64 - 1) / (T / 64);
        const long long cap = 256LL * (T == 512 ? 2 : 4);
        if (b2 > cap) b2 = cap;
        dens_done = lin16;
        if (lin16 && ldx == 0)
            hipLaunchKernelGGL((k_mlp_chain_fwd_b3<1, 3, false, false, MC_EPI_LIN16, 1, T>), dim3((unsigned)b2), dim3(T),
                               chain_b3_lds_elems(1, 3) * sizeof(uint16_t), (hipStream_t)stream, X, ldx, W0, in_real, W1, Wout, out, out_act,
                               (long long)N, H1, H2, Y, ldy, ChainSh{nullptr, 1, 0, dens_sel, dens});
        else if (lin16)
            hipLaunchKernelGGL((k_mlp_chain_fwd_b3<1, 3, false, false, MC_EPI_LIN16, 2, T>), dim3((unsigned)b2), dim3(T),
                               chain_b3_lds_elems(1, 3) * sizeof(uint16_t), (hipStream_t)stream, X, ldx, W0, in_real, W1, Wout, out, out_act,
                               (long long)N, H1, H2, Y, ldy, ChainSh{nullptr, 1, 0, dens_sel, dens});
        else if (n_hidden == 2)
            hipLaunchKernelGGL((k_mlp_chain_fwd_b3<2, 3>), dim3((unsigned)blocks), dim3(256), chain_b3_lds_elems(2, 3) * sizeof(uint16_t),
                               (hipStream_t)stream, X, ldx, W0, in_real, W1, Wout, out, out_act, (long long)N, H1, H2, Y, ldy, ChainSh{nullptr, 1, 0});
        else
            hipLaunchKernelGGL((k_mlp_chain_fwd_b3<1, 3>), dim3((unsigned)blocks), dim3(256), chain_b3_lds_elems(1, 3) * sizeof(uint16_t),
                               (hipStream_t)stream, X, ldx, W0, in_real, W1, Wout, out, out_act, (long long)N, H1, H2, Y, ldy, ChainSh{nullptr, 1, 0});
    } else if (n_hidden == 2)
        hipLaunchKernelGGL(k_mlp_chain_fwd<2>, dim3((unsigned)blocks), dim3(256), chain_lds_floats(2) * sizeof(float),
                           (hipStream_t)stream, X, ldx, W0, in_real, W1, Wout, out, out_act, (long long)N, H1, H2, Y, ldy);
    else
        hipLaunchKernelGGL(k_mlp_chain_fwd<1>, dim3((unsigned)blocks), dim3(256), chain_lds_floats(1) * sizeof(float),
                           (hipStream_t)stream, X, ldx, W0, in_real, W1, Wout, out, out_act, (long long)N, H1, H2, Y, ldy);
    SNF_LAUNCH_CHECK("snf_mlp64_fwd");
    if (dens != nullptr && !dens_done) return snf_trunc_exp_fwd(Y, ldy, dens_sel, N, dens, stream);
    return SNF_OK;
}

extern "C" int snf_mlp64_fwd(const float* X, int ldx, const float* W0, int in_real, const float* W1, const float* Wout,
                             int n_hidden, int out, int out_act, int64_t N, float* H1, float* H2, float* Y, int ldy,
                             snf_stream_t stream) {
    return chain_fwd_plain(X, ldx, W0, in_real, W1, Wout, n_hidden, out, out_act, N, H1, H2, Y, ldy, nullptr, nullptr, stream);
}

// snf_mlp64_fwd followed by snf_trunc_exp_fwd(Y, ldy, selector, N, density) (trunc_exp of output 0: nerfacto_field.py:244-252 after
// the base MLP), the density written from the chain's epilogue when it is the base net's (one hidden layer, 16 linear outputs,
// nothing else stored) instead of re-read at a 64-byte stride -- identical values
extern "C" int snf_mlp64_fwd_density(const float* X, int ldx, const float* W0, int in_real, const float* W1, const float* Wout,
                                     int n_hidden, int out, int out_act, int64_t N, float* H1, float* H2, float* Y, int ldy,
                                     const uint8_t* selector, float* density, snf_stream_t stream) {
    SNF_REQUIRE(density != nullptr, "snf_mlp64_fwd_density: null density");
    return chain_fwd_plain(X, ldx, W0, in_real, W1, Wout, n_hidden, out, out_act, N, H1, H2, Y, ldy, selector, density, stream);
}

extern "C" int snf_mlp64_bwd_data(const float* dY, int lddy, int dy_col_off, const float* dY0, const float* Y, int ldy,
                                  const float* W0, int in_real, const float* W1, const float* Wout, int n_hidden, int out,
                                  int out_act, int64_t N, const float* H1, const float* H2, float* dH1, float* dH2,
                                  float* dZ, int lddz, float* dX, int lddx, snf_stream_t stream) {
    int rc = chain_common_checks("snf_mlp64_bwd_data", in_real, n_hidden, out, N);
    if (rc) return rc;
    SNF_REQUIRE(dY && W0 && Wout && H1 && dH1 && (n_hidden == 1 || (W1 && H2 && dH2)), "snf_mlp64_bwd_data: null pointer");
    SNF_REQUIRE(out_act != SNF_ACT_SIGMOID || Y, "snf_mlp64_bwd_data: Y required for the sigmoid derivative");
    SNF_REQUIRE(out_act != SNF_ACT_RELU, "snf_mlp64_bwd_data: ReLU output activation is not supported");
    SNF_REQUIRE(!dZ || lddz >= out, "snf_mlp64_bwd_data: lddz < out");
    SNF_REQUIRE(!dX || ((lddx == 0 || (lddx >= MC_IN && lddx % 4 == 0)) && ((uintptr_t)dX % 16) == 0),
                "snf_mlp64_bwd_data: bad dX layout");
    SNF_REQUIRE(((uintptr_t)H1 % 16) == 0 && ((uintptr_t)dH1 % 16) == 0, "snf_mlp64_bwd_data: unaligned H1/dH1");
    const long long ntiles = (N + 31) / 32;
    long long blocks = (ntiles + 3) / 4;
    if (blocks > 256 * 4) blocks = 256 * 4;
    if (chain_b3_on()) {
        if (n_hidden == 2)
            hipLaunchKernelGGL(k_mlp_chain_bwd_b3<2>, dim3((unsigned)blocks), dim3(256), chain_b3_lds_elems(2) * sizeof(uint16_t),
                               (hipStream_t)stream, dY, lddy, dy_col_off, dY0, Y, ldy, W0, in_real, W1, Wout, out, out_act,
                               (long long)N, H1, H2, dH1, dH2, dZ, lddz, dX, lddx);
        else
            hipLaunchKernelGGL(k_mlp_chain_bwd_b3<1>, dim3((unsigned)blocks), dim3(256), chain_b3_lds_elems(1) * sizeof(uint16_t),
                               (hipStream_t)stream, dY, lddy, dy_col_off, dY0, Y, ldy, W0, in_real, W1, Wout, out, out_act,
                               (long long)N, H1, H2, dH1, dH2, dZ, lddz, dX, lddx);
    } else if (n_hidden == 2)
        hipLaunchKernelGGL(k_mlp_chain_bwd<2>, dim3((unsigned)blocks), dim3(256), chain_lds_floats(2) * sizeof(float),
                           (hipStream_t)stream, dY, lddy, dy_col_off, dY0, Y, ldy, W0, in_real, W1, Wout, out, out_act,
                           (long long)N, H1, H2, dH1, dH2, dZ, lddz, dX, lddx);
    else
        hipLaunchKernelGGL(k_mlp_chain_bwd<1>, dim3((unsigned)blocks), dim3(256), chain_lds_floats(1) * sizeof(float),
                           (hipStream_t)stream, dY, lddy, dy_col_off, dY0, Y, ldy, W0, in_real, W1, Wout, out, out_act,
                           (long long)N, H1, H2, dH1, dH2, dZ, lddz, dX, lddx);
    SNF_LAUNCH_CHECK("snf_mlp64_bwd_data");
    return SNF_OK;
}

// ---- data-gradient chain + weight gradients in one pass (k_mlp_chain_bwd_wg) -------------------------------------------------
extern "C" int64_t snf_mlp64_bwd_fused_workspace_bytes(int n_hidden) {
    if (n_hidden != 1 && n_hidden != 2) return 0;
    return (int64_t)256 * wg_partial_floats(n_hidden) * (int64_t)sizeof(float);
}

static int chain_bwd_fused(const float* dY, int lddy, int dy_col_off, const float* dY0, const float* Y, int ldy,
                           const float* X, int ldx, const float* W0, int in_real, const float* W1, const float* Wout,
                           int n_hidden, int out, int out_act, int64_t N, const float* H1, const float* H2, float* dX,
                           int lddx, float* dW0, float* dW1, float* dWout, void* workspace, int64_t workspace_bytes,
                           snf_stream_t stream, const float* sh_dirs, int sh_S);

extern "C" int snf_mlp64_bwd_fused(const float* dY, int lddy, int dy_col_off, const float* dY0, const float* Y, int ldy,
                                   const float* X, int ldx, const float* W0, int in_real, const float* W1, const float* Wout,
                                   int n_hidden, int out, int out_act, int64_t N, const float* H1, const float* H2, float* dX,
                                   int lddx, float* dW0, float* dW1, float* dWout, void* workspace, int64_t workspace_bytes,
                                   snf_stream_t stream) {
    return chain_bwd_fused(dY, lddy, dy_col_off, dY0, Y, ldy, X, ldx, W0, in_real, W1, Wout, n_hidden, out, out_act, N, H1, H2, dX, lddx,
                           dW0, dW1, dWout, workspace, workspace_bytes, stream, nullptr, 1);
}

extern "C" int snf_mlp64_bwd_fused_sh(const float* dY, int lddy, const float* Y, int ldy, const float* dirs, int R, int S,
                                      const float* base_out, int ld_base, int n_geo, const float* W0, const float* W1,
                                      const float* Wout, int n_hidden, int out, int out_act, float* d_geo, int ld_dgeo, float* dW0,
                                      float* dW1, float* dWout, void* workspace, int64_t workspace_bytes, snf_stream_t stream) {
    SNF_REQUIRE(dirs && base_out && R > 0 && S > 0 && n_geo == 15 && ld_base >= 16 && ld_base % 4 == 0 && ((uintptr_t)base_out % 16) == 0,
                "snf_mlp64_bwd_fused_sh: the base net's output must be [N, ld >= 16] (ld %% 4 == 0, aligned), geo in columns 1 .. 15");
    SNF_REQUIRE(!d_geo || (ld_dgeo >= 16 && ld_dgeo % 4 == 0 && ((uintptr_t)d_geo % 16) == 0),
                "snf_mlp64_bwd_fused_sh: d_geo must be [N, ld >= 16] (ld %% 4 == 0, 16-byte aligned)");
    // (X = the base net's output with its own stride, dX = the compact geo gradient: `sh_dirs != NULL` tells the shared body that the
    //  row-major [N, >= 32] layout checks of the plain entry do not apply -- both layouts were checked above)
    return chain_bwd_fused(dY, lddy, 0, nullptr, Y, ldy, base_out, ld_base, W0, 16 + n_geo, W1, Wout, n_hidden, out, out_act,
                           (int64_t)R * S, nullptr, nullptr, d_geo, ld_dgeo, dW0, dW1, dWout, workspace, workspace_bytes, stream, dirs, S);
}

static int chain_bwd_fused(const float* dY, int lddy, int dy_col_off, const float* dY0, const float* Y, int ldy,
                           const float* X, int ldx, const float* W0, int in_real, const float* W1, const float* Wout,
                           int n_hidden, int out, int out_act, int64_t N, const float* H1, const float* H2, float* dX,
                           int lddx, float* dW0, float* dW1, float* dWout, void* workspace, int64_t workspace_bytes,
                           snf_stream_t stream, const float* sh_dirs, int sh_S) {
    int rc = chain_common_checks("snf_mlp64_bwd_fused", in_real, n_hidden, out, N);
    if (rc) return rc;
    // dZ[s][o] is read at dY[s * lddy + dy_col_off + o]: column 0 may come from dY0 instead, and only then may the offset be -1
    // (the base net below the colour net: d(geo) is [N, 16] holding output columns 1 .. 15 in its columns 0 .. 14)
    SNF_REQUIRE(dy_col_off >= 0 || (dy_col_off == -1 && dY0 != nullptr),
                "snf_mlp64_bwd_fused: dy_col_off must be >= 0 (or -1 with dY0 supplying output column 0)");
    SNF_REQUIRE(lddy >= dy_col_off + out, "snf_mlp64_bwd_fused: lddy < dy_col_off + out (the gradient row does not hold the output columns)");
    const bool recompute = H1 == nullptr;  // the hidden activations are formed again from X (six-product forward arithmetic)
    SNF_REQUIRE(dY && X && W0 && Wout && dW0 && dWout && workspace && (n_hidden == 1 || (W1 && dW1 && (recompute || H2))),
                "snf_mlp64_bwd_fused: null pointer");
    SNF_REQUIRE(!recompute || (H2 == nullptr && snf_get_gemm_mode() == 1),
                "snf_mlp64_bwd_fused: H1 = NULL (recompute) needs H2 = NULL and gemm mode 1 (the forward whose arithmetic it repeats)");
    SNF_REQUIRE(out_act != SNF_ACT_SIGMOID || Y, "snf_mlp64_bwd_fused: Y required for the sigmoid derivative");
    SNF_REQUIRE(out_act != SNF_ACT_RELU, "snf_mlp64_bwd_fused: ReLU output activation is not supported");
    SNF_REQUIRE(sh_dirs || ((ldx == 0 || (ldx >= MC_IN && ldx % 4 == 0)) && ((uintptr_t)X % 16) == 0),
                "snf_mlp64_bwd_fused: X must be [N, ldx>=32] (ldx %% 4 == 0) or level-major (ldx = 0), 16-byte aligned");
    SNF_REQUIRE(sh_dirs || !dX || ((lddx == 0 || (lddx >= MC_IN && lddx % 4 == 0)) && ((uintptr_t)dX % 16) == 0),
                "snf_mlp64_bwd_fused: bad dX layout");
    SNF_REQUIRE((!H1 || ((uintptr_t)H1 % 16) == 0) && (!H2 || ((uintptr_t)H2 % 16) == 0) && ((uintptr_t)workspace % 16) == 0,
                "snf_mlp64_bwd_fused: unaligned H1 / H2 / workspace");
    SNF_REQUIRE(workspace_bytes >= snf_mlp64_bwd_fused_workspace_bytes(n_hidden), "snf_mlp64_bwd_fused: workspace too small");
    const long long ntiles = (N + 31) / 32;
    long long blocks = (ntiles + WG_WAVES - 1) / WG_WAVES;
    if (blocks > 256) blocks = 256;  // persistent: one workgroup per CU (LDS)
    float* P = (float*)workspace;
    const size_t lds = (size_t)(wg_weight_lds_floats(n_hidden) + WG_WAVES * (WG_STAGE + WG_DZ)) * sizeof(float) +
                       (recompute ? (size_t)wg_rc_lds_elems(n_hidden) * sizeof(uint16_t) : 0);
    hipStream_t st = (hipStream_t)stream;
    auto launch = [&](auto kern, bool& attr) {
        if (!attr) {
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr = true;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(WG_T), lds, st, dY, lddy, dy_col_off, dY0, Y, ldy, X, ldx, W0, in_real,
                           W1, Wout, out, out_act, (long long)N, H1, H2, dX, lddx, P, ChainSh{nullptr, 1, 0});
    };
    static bool a20 = false, a21 = false, a10 = false, a11 = false;
    if (sh_dirs != nullptr) {
        // the input row formed in the loader (recompute only): X = the base net's output, dX = the compact gradient of its geo columns
        const ChainSh sh = chain_sh(sh_dirs, sh_S);
        static bool s2 = false, s1 = false;
        auto launch_sh = [&](auto kern, bool& attr) {
            if (!attr) {
                hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                attr = true;
            }
            hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(WG_T), lds, st, dY, lddy, dy_col_off, dY0, Y, ldy, X, ldx, W0, in_real,
                               W1, Wout, out, out_act, (long long)N, H1, H2, dX, lddx, P, sh);
        };
        if (n_hidden == 2) launch_sh(k_mlp_chain_bwd_wg<2, true, true>, s2);
        else launch_sh(k_mlp_chain_bwd_wg<1, true, true>, s1);
    } else
    if (n_hidden == 2 && recompute) launch(k_mlp_chain_bwd_wg<2, true>, a21);
    else if (n_hidden == 2) launch(k_mlp_chain_bwd_wg<2, false>, a20);
    else if (recompute) launch(k_mlp_chain_bwd_wg<1, true>, a11);
    else launch(k_mlp_chain_bwd_wg<1, false>, a10);
    hipLaunchKernelGGL(k_chain_wgrad_reduce, dim3(wg_partial_floats(n_hidden) / 32), dim3(256), 0, st, P, (int)blocks,
                       n_hidden, out, in_real, dWout, dW1, dW0);
    SNF_LAUNCH_CHECK("snf_mlp64_bwd_fused");
    return SNF_OK;
}

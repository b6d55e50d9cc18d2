// linear.hip -- the dense layers of the tiny MLPs on the fp32 matrix cores (gfx950).
//
// All three GEMMs of a layer are "tall-skinny": one dimension is the number of samples (10^5..10^6), the
// other two are <= 256.  They run on v_mfma_f32_32x32x2_f32 (exact fp32: bitwise an fmaf chain), which is
// what keeps rendered RGB / feature tensors within 1e-4 of the fp32 oracle.
//
//   k_gemm_rows<BT,DERIV> : C[M,Nc] = op(A)[M,K] * B      (forward: BT=1, B = W[O,I] read transposed;
//                                                          data-grad: BT=0, B = W[O,I] read as [K=O][Nc=I])
//   k_gemm_wgrad          : dW[O,I] += op(dY)^T[O,rows] * X[rows,I]   split over row chunks, fp32 atomics
//
// Workgroup = 256 threads = 4 waves.  k_gemm_rows: tile 128 rows x 64 cols x 32 k; wave w owns rows
// [32w,32w+32) and two 32x32 accumulators.  Operands go through LDS k-major ([k][row]) so that an MFMA
// operand read is one conflict-free ds_read_b32 per lane; the row pitch (129 / 65 floats) makes the
// transposing ds_write_b32 pattern conflict-free as well.
#include "common.hpp"
#include <stdlib.h>

namespace snf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 64, BK = 32;
constexpr int LDA_S = BM + 1;  // 129: 4*129 mod 32 == 4 -> transposing stores hit 32 distinct banks
constexpr int LDB_S = BN + 1;  // 65

__device__ __forceinline__ float act_deriv(float y, int act) {
    if (act == SNF_ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (act == SNF_ACT_SIGMOID) return y * (1.f - y);
    return 1.f;
}

__device__ __forceinline__ float act_apply(float x, int act) {
    if (act == SNF_ACT_RELU) return fmaxf(x, 0.f);
    if (act == SNF_ACT_SIGMOID) return 1.f / (1.f + expf(-x));
    if (act == SNF_ACT_GELU) return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));  // nn.GELU() (erf form)
    return x;
}

// load 4 consecutive k-elements of row `row` starting at k (zero outside the matrix)
template <bool DERIV>
__device__ __forceinline__ float4 load_a4(const float* __restrict__ A, const float* __restrict__ Aux, int row, int k,
                                          int M, int K, int lda, int ldaux, int act, bool vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row >= M || k >= K) return v;
    const float* p = A + (size_t)row * lda + k;
    if (vec) {
        v = *reinterpret_cast<const float4*>(p);
    } else {
        v.x = p[0];
        if (k + 1 < K) v.y = p[1];
        if (k + 2 < K) v.z = p[2];
        if (k + 3 < K) v.w = p[3];
    }
    if constexpr (DERIV) {
        if (act != SNF_ACT_NONE) {
            const float* q = Aux + (size_t)row * ldaux + k;
            float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
            if (vec) {
                y = *reinterpret_cast<const float4*>(q);
            } else {
                y.x = q[0];
                if (k + 1 < K) y.y = q[1];
                if (k + 2 < K) y.z = q[2];
                if (k + 3 < K) y.w = q[3];
            }
            v.x *= act_deriv(y.x, act);
            v.y *= act_deriv(y.y, act);
            v.z *= act_deriv(y.z, act);
            v.w *= act_deriv(y.w, act);
        }
    }
    return v;
}

// Branch-free 16-byte variants (VEC kernels: every leading dimension and K / Nc a multiple of 4, 16-B aligned bases):
// out-of-range rows / k-quads are clamped to a valid address and zeroed by a select, so all loads of a tile issue
// back to back without exec-mask branches in between.
template <bool DERIV>
__device__ __forceinline__ float4 load_a4_vec(const float* __restrict__ A, const float* __restrict__ Aux, int row, int k,
                                              int M, int K, int lda, int ldaux, int act) {
    const bool ok = (row < M) && (k < K);
    const int rc = row < M ? row : M - 1;
    const int kc = k < K ? k : 0;  // k is a multiple of 4 and lda % 4 == 0: kc + 3 stays inside the (padded) row
    float4 v = *reinterpret_cast<const float4*>(A + (size_t)rc * lda + kc);
    if constexpr (DERIV) {
        if (act != SNF_ACT_NONE) {
            const float4 y = *reinterpret_cast<const float4*>(Aux + (size_t)rc * ldaux + kc);
            v.x *= act_deriv(y.x, act);
            v.y *= act_deriv(y.y, act);
            v.z *= act_deriv(y.z, act);
            v.w *= act_deriv(y.w, act);
        }
    }
    // pad columns (k+i >= K) of a padded row hold arbitrary bytes: select, never multiply
    v.x = ok ? v.x : 0.f;
    v.y = (ok && k + 1 < K) ? v.y : 0.f;
    v.z = (ok && k + 2 < K) ? v.z : 0.f;
    v.w = (ok && k + 3 < K) ? v.w : 0.f;
    return v;
}

// level-major ("planar") matrix [C/2 levels][R_total][2] read as if it were row-major [R_total, C] at (r, c), c % 4 == 0
__device__ __forceinline__ float4 load_b4_planar(const float* __restrict__ B, int r, int c, long long r_total) {
    const float2 a = *reinterpret_cast<const float2*>(B + ((long long)(c >> 1) * r_total + r) * 2);
    const float2 b = *reinterpret_cast<const float2*>(B + ((long long)((c >> 1) + 1) * r_total + r) * 2);
    return make_float4(a.x, a.y, b.x, b.y);
}

__device__ __forceinline__ float4 load_b4_vec(const float* __restrict__ B, int r, int c, int Rn, int Cn, int ldb,
                                              long long r_total = 0) {
    const bool ok = (r < Rn) && (c < Cn);
    const int rc = r < Rn ? r : Rn - 1;
    const int cc = c < Cn ? c : 0;
    float4 v = ldb ? *reinterpret_cast<const float4*>(B + (size_t)rc * ldb + cc) : load_b4_planar(B, rc, cc, r_total);
    v.x = ok ? v.x : 0.f;
    v.y = (ok && c + 1 < Cn) ? v.y : 0.f;
    v.z = (ok && c + 2 < Cn) ? v.z : 0.f;
    v.w = (ok && c + 3 < Cn) ? v.w : 0.f;
    return v;
}

// generic 4-wide load along the contiguous dimension of a row-major matrix [R, C] at (r, c)
__device__ __forceinline__ float4 load_b4(const float* __restrict__ B, int r, int c, int Rn, int Cn, int ldb, bool vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r >= Rn || c >= Cn) return v;
    const float* p = B + (size_t)r * ldb + c;
    if (vec) {
        v = *reinterpret_cast<const float4*>(p);
    } else {
        v.x = p[0];
        if (c + 1 < Cn) v.y = p[1];
        if (c + 2 < Cn) v.z = p[2];
        if (c + 3 < Cn) v.w = p[3];
    }
    return v;
}

template <bool BT, bool DERIV, bool VEC>
__global__ __launch_bounds__(256) void k_gemm_rows(const float* __restrict__ A, const float* __restrict__ Aux,
                                                   const float* __restrict__ B, const float* __restrict__ bias, int M,
                                                   int K, int Nc, int lda, int ldaux, int ldb, int ldc, int act_in,
                                                   int act_out, int vecA, int vecB, float* __restrict__ C, int ksplit = 0,
                                                   long long c_split_stride = 0) {
    __shared__ float As[BK * LDA_S];
    __shared__ float Bs[BK * LDB_S];
    if constexpr (BT && !DERIV) {
        if (ksplit > 0) {  // split-K slice z: a GEMM on the k-range [z*ksplit, ...) into the z-th partial buffer
            const int kb = blockIdx.z * ksplit;
            A += kb; B += kb; K = min(ksplit, K - kb);
            C += (size_t)blockIdx.z * c_split_stride;
        }
    }
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int row0 = blockIdx.x * BM;
    const int col0 = blockIdx.y * BN;
    const bool two = (col0 + 32) < Nc;  // second 32-column accumulator needed?
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }

    const int a_kq = tid & 7, a_r = tid >> 3;  // A staging: 8 k-quads x 32 rows, 4 passes
    const int b_jq = tid & 15, b_k = tid >> 4;  // B staging when B is [K, Nc]: 16 col-quads x 16 k, 2 passes
    float4 av[4], bv[2];
    // global -> registers for the tile starting at k0 (software pipeline: issued one tile ahead of its use)
    auto fetch = [&](int k0) {
        if constexpr (VEC) {
#pragma unroll
            for (int p = 0; p < 4; ++p)
                av[p] = load_a4_vec<DERIV>(A, Aux, row0 + a_r + 32 * p, k0 + a_kq * 4, M, K, lda, ldaux, act_in);
            if constexpr (BT) {
#pragma unroll
                for (int p = 0; p < 2; ++p) bv[p] = load_b4_vec(B, col0 + a_r + 32 * p, k0 + a_kq * 4, Nc, K, ldb);
            } else {
#pragma unroll
                for (int p = 0; p < 2; ++p) bv[p] = load_b4_vec(B, k0 + b_k + 16 * p, col0 + b_jq * 4, K, Nc, ldb);
            }
        } else {
#pragma unroll
            for (int p = 0; p < 4; ++p)
                av[p] = load_a4<DERIV>(A, Aux, row0 + a_r + 32 * p, k0 + a_kq * 4, M, K, lda, ldaux, act_in, vecA);
            if constexpr (BT) {
#pragma unroll
                for (int p = 0; p < 2; ++p) bv[p] = load_b4(B, col0 + a_r + 32 * p, k0 + a_kq * 4, Nc, K, ldb, vecB);
            } else {
#pragma unroll
                for (int p = 0; p < 2; ++p) bv[p] = load_b4(B, k0 + b_k + 16 * p, col0 + b_jq * 4, K, Nc, ldb, vecB);
            }
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        __syncthreads();  // previous tile fully consumed
        // ---- registers -> LDS (A transposed into [k][row])
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int r = a_r + 32 * p;
            As[(a_kq * 4 + 0) * LDA_S + r] = av[p].x;
            As[(a_kq * 4 + 1) * LDA_S + r] = av[p].y;
            As[(a_kq * 4 + 2) * LDA_S + r] = av[p].z;
            As[(a_kq * 4 + 3) * LDA_S + r] = av[p].w;
        }
        if constexpr (BT) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int j = a_r + 32 * p;
                Bs[(a_kq * 4 + 0) * LDB_S + j] = bv[p].x;
                Bs[(a_kq * 4 + 1) * LDB_S + j] = bv[p].y;
                Bs[(a_kq * 4 + 2) * LDB_S + j] = bv[p].z;
                Bs[(a_kq * 4 + 3) * LDB_S + j] = bv[p].w;
            }
        } else {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float* d = &Bs[(b_k + 16 * p) * LDB_S + b_jq * 4];
                d[0] = bv[p].x; d[1] = bv[p].y; d[2] = bv[p].z; d[3] = bv[p].w;
            }
        }
        __syncthreads();
        if (k0 + BK < K) fetch(k0 + BK);  // next tile's HBM/L2 latency hides under this tile's MFMAs
        // ---- 16 k-pairs on the matrix core
        const int kh = lane >> 5, li = lane & 31;
        const float* ap = &As[kh * LDA_S + wave * 32 + li];
        const float* bp = &Bs[kh * LDB_S + li];
        if (two) {
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                const float a = ap[kk * 2 * LDA_S];
                const float b0 = bp[kk * 2 * LDB_S];
                const float b1 = bp[kk * 2 * LDB_S + 32];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                const float a = ap[kk * 2 * LDA_S];
                const float b0 = bp[kk * 2 * LDB_S];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
            }
        }
    }
    // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int li = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = row0 + wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kh;
        if (row < M) {
            const int c0 = col0 + li;
            if (c0 < Nc) {
                float v = acc0[reg];
                if (bias) v += bias[c0];
                C[(size_t)row * ldc + c0] = act_apply(v, act_out);
            }
            const int c1 = c0 + 32;
            if (two && c1 < Nc) {
                float v = acc1[reg];
                if (bias) v += bias[c1];
                C[(size_t)row * ldc + c1] = act_apply(v, act_out);
            }
        }
    }
}

// dW[O,I] += sum over this block's rows of dZ[n,o] * X[n,i];   tile 64(o) x 64(i), waves 2x2.
constexpr int WG_LD = 65;
template <bool VEC>
__global__ __launch_bounds__(256) void k_gemm_wgrad(const float* __restrict__ dY, const float* __restrict__ Y,
                                                    const float* __restrict__ X, int N, int I, int O, int lddy, int ldy,
                                                    int ldx, int act, int rows_per_block, int vecA, int vecB,
                                                    float* __restrict__ dW, float* __restrict__ dbias) {
    __shared__ float As[BK * WG_LD];  // [n][o]
    __shared__ float Bs[BK * WG_LD];  // [n][i]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int o0 = blockIdx.x * 64, i0 = blockIdx.y * 64;
    const int n_begin = blockIdx.z * rows_per_block;
    const int n_end = min(N, n_begin + rows_per_block);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float bsum = 0.f;
    const int q = tid & 15, kr = tid >> 4;  // 16 quads x 16 rows, 2 passes
    float4 av[2], bv[2];
    auto fetch = [&](int n0) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int n = n0 + kr + 16 * p;
            const bool ok = n < n_end;
            // A: dZ[n][o0 + q*4 ..]  (activation derivative folded in)
            if constexpr (VEC) {
                av[p] = load_a4_vec<true>(dY, Y, n, o0 + q * 4, n_end, O, lddy, ldy, act);
                bv[p] = load_b4_vec(X, n, i0 + q * 4, n_end, I, ldx, N);
            } else {
                av[p] = ok ? load_a4<true>(dY, Y, n, o0 + q * 4, N, O, lddy, ldy, act, vecA) : make_float4(0, 0, 0, 0);
                bv[p] = ok ? load_b4(X, n, i0 + q * 4, N, I, ldx, vecB) : make_float4(0, 0, 0, 0);
            }
        }
    };
    fetch(n_begin);
    for (int n0 = n_begin; n0 < n_end; n0 += BK) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float* da = &As[(kr + 16 * p) * WG_LD + q * 4];
            da[0] = av[p].x; da[1] = av[p].y; da[2] = av[p].z; da[3] = av[p].w;
            float* db = &Bs[(kr + 16 * p) * WG_LD + q * 4];
            db[0] = bv[p].x; db[1] = bv[p].y; db[2] = bv[p].z; db[3] = bv[p].w;
        }
        __syncthreads();
        if (n0 + BK < n_end) fetch(n0 + BK);
        const int kh = lane >> 5, li = lane & 31;
        const float* ap = &As[kh * WG_LD + wm * 32 + li];
        const float* bp = &Bs[kh * WG_LD + wn * 32 + li];
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk * 2 * WG_LD], bp[kk * 2 * WG_LD], acc, 0, 0, 0);
        if (dbias != nullptr && blockIdx.y == 0 && tid < 64) {
#pragma unroll 8
            for (int k = 0; k < BK; ++k) bsum += As[k * WG_LD + tid];
        }
    }
    const int li = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int o = o0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kh;
        const int i = i0 + wn * 32 + li;
        if (o < O && i < I) unsafeAtomicAdd(&dW[(size_t)o * I + i], acc[reg]);
    }
    if (dbias != nullptr && blockIdx.y == 0 && tid < 64 && o0 + tid < O) unsafeAtomicAdd(&dbias[o0 + tid], bsum);
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace snf

using namespace snf;

extern "C" int snf_linear_fwd(const float* X, const float* W, const float* bias, int N, int I, int O, int ldx, int ldy,
                              int act, float* Y, snf_stream_t stream) {
    SNF_REQUIRE(X && W && Y, "snf_linear_fwd: null pointer");
    // ldx == -8: X is level-major [I/8][N][8] (two feature grids' planar output side by side)
    SNF_REQUIRE(N > 0 && I > 0 && O > 0 && (ldx >= I || (ldx == -8 && I % 16 == 0)) && ldy >= O,
                "snf_linear_fwd: bad shape N=%d I=%d O=%d", N, I, O);
    SNF_REQUIRE(act >= 0 && act <= 3, "snf_linear_fwd: bad activation %d", act);
    const int took = b3_try_fwd(X, W, bias, N, I, O, ldx, ldy, act, Y, stream);
    SNF_REQUIRE(took >= 0 && (took > 0 || ldx > 0),
                "snf_linear_fwd: a level-major X (ldx = -8) needs 64 <= I <= 256, I %% 16 == 0, O >= 64, aligned pointers and gemm mode >= 1");
    if (took) {
        SNF_LAUNCH_CHECK("snf_linear_fwd(bf16x3)");
        return SNF_OK;
    }
    const int vecA = aligned16(X) && (ldx % 4 == 0) && (I % 4 == 0);
    const int vecB = aligned16(W) && (I % 4 == 0);
    dim3 grid(ceil_div(N, BM), ceil_div(O, BN));
    if (vecA && vecB && N >= 1)
        hipLaunchKernelGGL((k_gemm_rows<true, false, true>), grid, dim3(256), 0, (hipStream_t)stream, X,
                           (const float*)nullptr, W, bias, N, I, O, ldx, 0, I, ldy, SNF_ACT_NONE, act, vecA, vecB, Y);
    else
        hipLaunchKernelGGL((k_gemm_rows<true, false, false>), grid, dim3(256), 0, (hipStream_t)stream, X,
                           (const float*)nullptr, W, bias, N, I, O, ldx, 0, I, ldy, SNF_ACT_NONE, act, vecA, vecB, Y);
    SNF_LAUNCH_CHECK("snf_linear_fwd");
    return SNF_OK;
}

namespace snf {
// Y[n,o] = act(sum_z P[z][n][o] + bias[o])
__global__ __launch_bounds__(256) void k_splitk_epilogue(const float* __restrict__ P, int splits, long long stride, int M,
                                                         int Nc, const float* __restrict__ bias, int act, float* __restrict__ Y,
                                                         int ldy) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)M * Nc) return;
    const int n = (int)(t / Nc), o = (int)(t - (long long)n * Nc);
    float v = 0.f;
    for (int z = 0; z < splits; ++z) v += P[(size_t)z * stride + t];
    if (bias) v += bias[o];
    Y[(size_t)n * ldy + o] = act_apply(v, act);
}

// split-K plan for a forward GEMM whose 128 x 64 output tiles do not fill the chip: number of k-slices (1 = no split)
static int splitk_plan(int N, int I, int O, int* ksplit) {
    const int tiles = ceil_div(N, BM) * ceil_div(O, BN);
    *ksplit = I;
    if (tiles >= 256 || I < 512 || (I % 4)) return 1;
    int splits = ceil_div(512, tiles);
    if (splits > I / 128) splits = I / 128;
    if (splits < 2) return 1;
    int ks = ceil_div(I, splits);
    ks = ((ks + 31) / 32) * 32;
    *ksplit = ks;
    return ceil_div(I, ks);
}
}  // namespace snf

extern "C" int64_t snf_linear_fwd_workspace_bytes(int N, int I, int O) {
    int ks;
    const int splits = splitk_plan(N, I, O, &ks);
    return splits > 1 ? (int64_t)splits * N * O * (int64_t)sizeof(float) : 0;
}

extern "C" int snf_linear_fwd_ws(const float* X, const float* W, const float* bias, int N, int I, int O, int ldx, int ldy,
                                 int act, float* Y, void* workspace, int64_t workspace_bytes, snf_stream_t stream) {
    int ks;
    const int splits = (X && W && Y && N > 0 && I > 0 && O > 0) ? splitk_plan(N, I, O, &ks) : 1;
    if (splits <= 1 || workspace == nullptr) return snf_linear_fwd(X, W, bias, N, I, O, ldx, ldy, act, Y, stream);
    SNF_REQUIRE(ldx >= I && ldy >= O && act >= 0 && act <= 3, "snf_linear_fwd_ws: bad shape / activation");
    SNF_REQUIRE(workspace_bytes >= snf_linear_fwd_workspace_bytes(N, I, O), "snf_linear_fwd_ws: workspace too small");
    SNF_REQUIRE(aligned16(X) && aligned16(W) && aligned16(workspace) && (ldx % 4 == 0),
                "snf_linear_fwd_ws: X, W, workspace must be 16-byte aligned and ldx a multiple of 4");
    float* P = (float*)workspace;
    const long long stride = (long long)N * O;
    dim3 grid(ceil_div(N, BM), ceil_div(O, BN), splits);
    if (!b3_try_fwd_splitk(X, W, N, I, O, ldx, ks, splits, P, stream))
        hipLaunchKernelGGL((k_gemm_rows<true, false, true>), grid, dim3(256), 0, (hipStream_t)stream, X,
                           (const float*)nullptr, W, (const float*)nullptr, N, I, O, ldx, 0, I, O, SNF_ACT_NONE, SNF_ACT_NONE,
                           1, 1, P, ks, stride);
    const long long total = (long long)N * O;
    hipLaunchKernelGGL(k_splitk_epilogue, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P, splits,
                       stride, N, O, bias, act, Y, ldy);
    SNF_LAUNCH_CHECK("snf_linear_fwd_ws");
    return SNF_OK;
}

extern "C" int snf_linear_bwd_data(const float* dY, const float* Y, const float* W, int N, int I, int O, int lddy,
                                   int ldy, int lddx, int act, float* dX, snf_stream_t stream) {
    SNF_REQUIRE(dY && W && dX, "snf_linear_bwd_data: null pointer");
    SNF_REQUIRE(act == SNF_ACT_NONE || Y, "snf_linear_bwd_data: Y required for activation derivative");
    SNF_REQUIRE(act != SNF_ACT_GELU, "snf_linear_bwd_data: GELU is forward-only (image encoder inference)");
    // lddx == -8: dX is written level-major [I/8][N][8] (the staged-gradient layout of snf_hashgrid_bwd_presorted, ld_out = 0)
    SNF_REQUIRE(N > 0 && I > 0 && O > 0 && lddy >= O && (lddx >= I || (lddx == -8 && I % 8 == 0)), "snf_linear_bwd_data: bad shape");
    const int took = b3_try_bwd_data(dY, Y, W, N, I, O, lddy, ldy, lddx, act, dX, stream);
    SNF_REQUIRE(took >= 0 && (took > 0 || lddx > 0),
                "snf_linear_bwd_data: a level-major dX (lddx = -8) needs 64 <= O <= 256, O %% 16 == 0, I >= 64, aligned pointers and gemm mode >= 1");
    if (took) {
        SNF_LAUNCH_CHECK("snf_linear_bwd_data(bf16x3)");
        return SNF_OK;
    }
    const int vecA = aligned16(dY) && (lddy % 4 == 0) && (O % 4 == 0) &&
                     (act == SNF_ACT_NONE || (aligned16(Y) && ldy % 4 == 0));
    const int vecB = aligned16(W) && (I % 4 == 0);
    dim3 grid(ceil_div(N, BM), ceil_div(I, BN));
    if (vecA && vecB)
        hipLaunchKernelGGL((k_gemm_rows<false, true, true>), grid, dim3(256), 0, (hipStream_t)stream, dY, Y, W,
                           (const float*)nullptr, N, O, I, lddy, ldy, I, lddx, act, SNF_ACT_NONE, vecA, vecB, dX);
    else
        hipLaunchKernelGGL((k_gemm_rows<false, true, false>), grid, dim3(256), 0, (hipStream_t)stream, dY, Y, W,
                           (const float*)nullptr, N, O, I, lddy, ldy, I, lddx, act, SNF_ACT_NONE, vecA, vecB, dX);
    SNF_LAUNCH_CHECK("snf_linear_bwd_data");
    return SNF_OK;
}

extern "C" int snf_linear_bwd_weight(const float* dY, const float* Y, const float* X, int N, int I, int O, int lddy,
                                     int ldy, int ldx, int act, float* dW, float* dbias, snf_stream_t stream) {
    SNF_REQUIRE(dY && X && dW, "snf_linear_bwd_weight: null pointer");
    SNF_REQUIRE(act == SNF_ACT_NONE || Y, "snf_linear_bwd_weight: Y required for activation derivative");
    SNF_REQUIRE(act != SNF_ACT_GELU, "snf_linear_bwd_weight: GELU is forward-only (image encoder inference)");
    // ldx == 0: X is level-major [I/2][N][2] (the hash-grid forward's planar output); needs the vector path
    // ldx == -8: X is level-major [I/8][N][8] (feature grids); only the bf16x3 kernel reads that layout
    SNF_REQUIRE(N > 0 && I > 0 && O > 0 && lddy >= O && (ldx >= I || (ldx == 0 && I % 4 == 0) || (ldx == -8 && I % 8 == 0)),
                "snf_linear_bwd_weight: bad shape");
    if (b3_try_bwd_weight(dY, Y, X, N, I, O, lddy, ldy, ldx, act, dW, dbias, stream)) {
        SNF_LAUNCH_CHECK("snf_linear_bwd_weight(bf16x3)");
        return SNF_OK;
    }
    SNF_REQUIRE(ldx >= 0, "snf_linear_bwd_weight: a level-major X with 8 features per level (ldx = -8) needs I, O >= 64, "
                          "aligned pointers and gemm mode >= 1");
    // activations may be stored with padded leading dimensions (multiple of 4): the vector loaders mask pad columns
    const int vecA = aligned16(dY) && (lddy % 4 == 0) && (act == SNF_ACT_NONE || (aligned16(Y) && ldy % 4 == 0));
    const int vecB = aligned16(X) && (ldx % 4 == 0);
    const int to = ceil_div(O, 64), ti = ceil_div(I, 64);
    // aim for ~1024 workgroups (measured best for the 64-wide nets: 2048 doubles the atomics of the final accumulation, 512
    // leaves the CUs short of loads in flight); every chunk is a multiple of BK rows
    static const int target = 1024;
    int chunks = target / (to * ti);
    if (chunks < 1) chunks = 1;
    int rows = ceil_div(N, chunks);
    rows = ((rows + BK - 1) / BK) * BK;
    if (rows < 4 * BK) rows = 4 * BK;
    chunks = ceil_div(N, rows);
    dim3 grid(to, ti, chunks);
    SNF_REQUIRE(ldx != 0 || (vecA && vecB), "snf_linear_bwd_weight: a level-major X needs 16-byte aligned dY / X");
    if (vecA && vecB)
        hipLaunchKernelGGL(k_gemm_wgrad<true>, grid, dim3(256), 0, (hipStream_t)stream, dY, Y, X, N, I, O, lddy, ldy, ldx,
                           act, rows, vecA, vecB, dW, dbias);
    else
        hipLaunchKernelGGL(k_gemm_wgrad<false>, grid, dim3(256), 0, (hipStream_t)stream, dY, Y, X, N, I, O, lddy, ldy, ldx,
                           act, rows, vecA, vecB, dW, dbias);
    SNF_LAUNCH_CHECK("snf_linear_bwd_weight");
    return SNF_OK;
}

// Weight gradient with a caller-provided scratch buffer: the feature-head shapes (64 <= I, O <= 256, N >= 8192, no bias, gemm
// mode >= 1) run the full-width kernel of linear_b3.hip -- every operand element read once, per-chunk partial sums in the
// scratch buffer, a second small kernel adds them to dW.  Everything else (and a NULL / short workspace) is
// snf_linear_bwd_weight.
extern "C" int64_t snf_linear_bwd_weight_workspace_bytes(int N, int I, int O) {
    if (N <= 0 || I <= 0 || O <= 0) return 0;
    return (int64_t)b3_wgrad_full_workspace_bytes(N, I, O);
}

extern "C" int snf_linear_bwd_weight_ws(const float* dY, const float* Y, const float* X, int N, int I, int O, int lddy, int ldy,
                                        int ldx, int act, float* dW, float* dbias, void* workspace, int64_t workspace_bytes,
                                        snf_stream_t stream) {
    SNF_REQUIRE(dY && X && dW, "snf_linear_bwd_weight_ws: null pointer");
    SNF_REQUIRE(act == SNF_ACT_NONE || Y, "snf_linear_bwd_weight_ws: Y required for activation derivative");
    if (N > 0 && I > 0 && O > 0 && lddy >= O && act != SNF_ACT_GELU &&
        b3_try_bwd_weight_full(dY, Y, X, N, I, O, lddy, ldy, ldx, act, dW, dbias, workspace, workspace_bytes, stream)) {
        SNF_LAUNCH_CHECK("snf_linear_bwd_weight_ws");
        return SNF_OK;
    }
    return snf_linear_bwd_weight(dY, Y, X, N, I, O, lddy, ldy, ldx, act, dW, dbias, stream);
}

// ---- the same two gradients when dY is the gradient of a weighted mean over `group` consecutive rows and is never written out
extern "C" int snf_linear_fwd_mean(const float* X, const float* W, int N, int I, int O, int ldx, const float* row_weight, int group,
                                   float* Hbar, uint8_t* Ymask, float* Y, int ldy, snf_stream_t stream) {
    SNF_REQUIRE(X && W && row_weight && Hbar && Ymask, "snf_linear_fwd_mean: null pointer");
    SNF_REQUIRE(N > 0 && I > 0 && O > 0 && (ldx >= I || (ldx == -8 && I % 16 == 0)) && (Y == nullptr || ldy >= O),
                "snf_linear_fwd_mean: bad shape N=%d I=%d O=%d", N, I, O);
    const int took = b3_try_fwd_mean(X, W, N, I, O, ldx, Y ? ldy : O, Y, row_weight, group, Hbar, Ymask, stream);
    SNF_REQUIRE(took > 0, "snf_linear_fwd_mean: needs the weight-stationary bf16-split kernel (group == 16, N %% 16 == 0, O %% 32 == 0, "
                          "64 <= I <= 256, I %% 16 == 0, N >= 4096, aligned pointers, gemm mode >= 1); use snf_linear_fwd + "
                          "snf_feature_mean_fwd otherwise");
    SNF_LAUNCH_CHECK("snf_linear_fwd_mean");
    return SNF_OK;
}

extern "C" int snf_linear_bwd_data_rows(const float* dYg, const float* row_scale, int group, const float* Y, int y_is_mask,
                                        const float* W, int N, int I, int O, int lddy, int ldy, int lddx, int act, float* dX,
                                        snf_stream_t stream) {
    SNF_REQUIRE(dYg && row_scale && W && dX, "snf_linear_bwd_data_rows: null pointer");
    SNF_REQUIRE(act == SNF_ACT_NONE || Y, "snf_linear_bwd_data_rows: Y required for activation derivative");
    SNF_REQUIRE(N > 0 && I > 0 && O > 0 && group > 0 && lddy >= O && (lddx >= I || (lddx == -8 && I % 8 == 0)) &&
                    act != SNF_ACT_GELU, "snf_linear_bwd_data_rows: bad shape");
    SNF_REQUIRE(!y_is_mask || (act == SNF_ACT_RELU && O % 8 == 0), "snf_linear_bwd_data_rows: a bit mask stands for a ReLU output");
    const int took = b3_try_bwd_data(dYg, Y, W, N, I, O, lddy, ldy, lddx, act, dX, stream, row_scale, group, y_is_mask ? 1 : 0);
    SNF_REQUIRE(took > 0, "snf_linear_bwd_data_rows: needs the weight-stationary bf16-split kernel (64 <= O <= 256, O %% 16 == 0, "
                          "I >= 64, N >= 4096, aligned pointers, gemm mode >= 1); write the broadcast gradient with "
                          "snf_feature_mean_bwd and call snf_linear_bwd_data otherwise");
    SNF_LAUNCH_CHECK("snf_linear_bwd_data_rows");
    return SNF_OK;
}

extern "C" int snf_linear_bwd_weight_rows(const float* dYg, const float* row_scale, int group, const float* Y, int y_is_mask,
                                          const float* X, int N, int I, int O, int lddy, int ldy, int ldx, int act, float* dW,
                                          void* workspace, int64_t workspace_bytes, snf_stream_t stream) {
    SNF_REQUIRE(dYg && row_scale && X && dW, "snf_linear_bwd_weight_rows: null pointer");
    SNF_REQUIRE(act == SNF_ACT_NONE || Y, "snf_linear_bwd_weight_rows: Y required for activation derivative");
    SNF_REQUIRE(N > 0 && I > 0 && O > 0 && group > 0 && lddy >= O && act != SNF_ACT_GELU, "snf_linear_bwd_weight_rows: bad shape");
    SNF_REQUIRE(!y_is_mask || (act == SNF_ACT_RELU && O % 8 == 0), "snf_linear_bwd_weight_rows: a bit mask stands for a ReLU output");
    const int took = b3_try_bwd_weight_full(dYg, Y, X, N, I, O, lddy, ldy, ldx, act, dW, nullptr, workspace, workspace_bytes, stream,
                                            row_scale, group, y_is_mask ? 1 : 0);
    SNF_REQUIRE(took > 0, "snf_linear_bwd_weight_rows: needs the full-width kernel (snf_linear_bwd_weight_workspace_bytes > 0 and a "
                          "workspace of that size); write the broadcast gradient with snf_feature_mean_bwd and call "
                          "snf_linear_bwd_weight_ws otherwise");
    SNF_LAUNCH_CHECK("snf_linear_bwd_weight_rows");
    return SNF_OK;
}


// mlp_tiny.hip -- the proposal networks' density MLP (tcnn FullyFusedMLP with 16 neurons, one hidden layer, one output:
// nerfstudio/fields/density_fields.py:80-97 with the samnerf proposal_net_args hidden_dim = 16; input = the 5-level F=2 hash
// encoding, 10 wide) in one launch per direction (gfx950).
//
// 176 multiply-adds per sample is far below the point where a matrix-core tile pays: one thread per sample on the vector
// ALU, weights broadcast from LDS.  The backward forms the weight gradient of a wave's 64 samples on the matrix cores (see
// k_mlp_tiny_bwd), folds the waves through LDS and issues ONE set of global atomics per workgroup -- the three GEMM launches + two
// wgrad launches this replaces spent most of their time on 10-wide unaligned rows.
#include "common.hpp"
#include "mlp_tiny_device.hpp"

namespace snf {

constexpr int MT_SPT = 4;  // samples per thread in the backward

template <int I, int H>
__global__ __launch_bounds__(256) void k_mlp_tiny_fwd(const float* __restrict__ X, int ldx, const float* __restrict__ W0,
                                                      const float* __restrict__ W1, long long N, float* __restrict__ Hid,
                                                      float* __restrict__ Y) {
    __shared__ float w0[H * I], w1[H];
    for (int i = threadIdx.x; i < H * I; i += 256) w0[i] = W0[i];
    if (threadIdx.x < H) w1[threadIdx.x] = W1[threadIdx.x];
    __syncthreads();
    const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float x[I];
    const float* xp = X + n * ldx;
#pragma unroll
    for (int i = 0; i < I; i += 2) {
        const float2 t = *reinterpret_cast<const float2*>(xp + i);
        x[i] = t.x; x[i + 1] = t.y;
    }
    float h[H];
    const float y = mt_forward<I, H>(x, w0, w1, h);
    if (Hid != nullptr) {
#pragma unroll
        for (int j = 0; j < H; j += 4)
            *reinterpret_cast<float4*>(Hid + n * H + j) = make_float4(h[j], h[j + 1], h[j + 2], h[j + 3]);
    }
    Y[n] = y;
}

// Backward.  A thread owns one sample per trip: dh = relu'(h) dy w1, dX = dh W0 on the vector ALU (weights broadcast from LDS).  The
// weight gradient dW0[j][i] = sum_n dh[n][j] x[n][i] contracts over SAMPLES: round 1 kept all H*I partial sums in every thread's
// registers (460 of the SIMD's 512: one wave per SIMD, and inside the step the kernel could only start on a CU nothing else occupied
// -- 0.32 ms against 0.055 alone, event timeline r04) and reduced them with 176 x 6 wave shuffles.  Here a wave hands its 64 samples'
// dh and x rows through a private LDS tile (pitch 52 floats: conflict-free 16-byte row writes) to v_mfma_f32_16x16x4_f32 -- A = dh^T
// (16 hidden units x 4 samples), B = x (4 samples x 16 padded inputs), 16 instructions per 64 samples, the running 16 x 16 sum in four
// accumulator registers -- exact fp32 like the fmaf chain it replaces.  dW1[j] = sum_n dy h[n][j] stays a 16-register per-thread
// sum.  ~80 registers, 12.6 KB of LDS per wave.
constexpr int MT_PITCH = 52;
typedef float mt_f32x4 __attribute__((ext_vector_type(4)));

template <int I, int H>
__global__ __launch_bounds__(256) void k_mlp_tiny_bwd(const float* __restrict__ dY, const float* __restrict__ X, int ldx,
                                                      const float* __restrict__ Hid, const float* __restrict__ W0,
                                                      const float* __restrict__ W1, long long N, float* __restrict__ dX,
                                                      int lddx, float* __restrict__ dW0, float* __restrict__ dW1) {
    static_assert(H == 16 && I <= 16 && I % 2 == 0, "the MFMA tile below is 16 hidden units x 16 (padded) inputs");
    constexpr int NW = H * I + H;
    __shared__ __attribute__((aligned(16))) float tile[4][64 * MT_PITCH];  // per wave: [sample][dh 0..15 | x 16..31]
    __shared__ float part[4][NW];
    // (the 176 weights are read at wave-uniform addresses: scalar loads into SGPRs, a group of four hidden units at a time -- as
    //  LDS broadcasts the compiler kept all of them in vector registers across the loop)
    const float* __restrict__ w0 = W0;
    const float* __restrict__ w1 = W1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* __restrict__ T = tile[wave];
    {   // the pad inputs I .. 15 of a sample's x row are never written again: zero once
        float* row = T + lane * MT_PITCH + 16;
#pragma unroll
        for (int i = I; i < 16; ++i) row[i] = 0.f;
    }
    __syncthreads();
    float a1[H];
#pragma unroll
    for (int j = 0; j < H; ++j) a1[j] = 0.f;
    mt_f32x4 acc = {0.f, 0.f, 0.f, 0.f};  // lane (i = lane % 16, g = lane / 16): dW0[4 g + r][i], r = 0..3
#pragma unroll 1
    for (int sidx = 0; sidx < MT_SPT; ++sidx) {
        const long long n = ((long long)blockIdx.x * MT_SPT + sidx) * 256 + threadIdx.x;
        const bool ok = n < N;
        const long long nc = ok ? n : N - 1;
        const float dy = ok ? dY[nc] : 0.f;  // (a row beyond N contributes exact zeros)
        float x[I], dx[I], dh[H];
        const float* xp = X + nc * ldx;
#pragma unroll
        for (int i = 0; i < I; i += 2) {
            const float2 t = *reinterpret_cast<const float2*>(xp + i);
            x[i] = t.x; x[i + 1] = t.y;
            dx[i] = 0.f; dx[i + 1] = 0.f;
        }
#pragma unroll
        for (int j4 = 0; j4 < H; j4 += 4) {
            const float4 hv = *reinterpret_cast<const float4*>(Hid + nc * H + j4);
            const float h4[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = j4 + q;
                a1[j] += dy * h4[q];
                dh[j] = h4[q] > 0.f ? dy * w1[j] : 0.f;
#pragma unroll
                for (int i = 0; i < I; ++i) dx[i] += dh[j] * w0[j * I + i];
            }
            // (keeps the 40 weight reads of a group of four hidden units behind the previous group's arithmetic: left alone, the
            //  scheduler fetches all 160 up front and the kernel is back at one wave per SIMD)
            __builtin_amdgcn_sched_barrier(0);
        }
        if (dX != nullptr && ok) {
#pragma unroll
            for (int i = 0; i < I; i += 2) *reinterpret_cast<float2*>(dX + n * lddx + i) = make_float2(dx[i], dx[i + 1]);
        }
        // ---- this wave's 64 (dh, x) rows -> LDS -> 16 matrix instructions (a wave's LDS traffic is ordered: no barrier)
        float* row = T + lane * MT_PITCH;
#pragma unroll
        for (int j = 0; j < H; j += 4) *reinterpret_cast<float4*>(row + j) = make_float4(dh[j], dh[j + 1], dh[j + 2], dh[j + 3]);
#pragma unroll
        for (int i = 0; i < I; i += 2) *reinterpret_cast<float2*>(row + 16 + i) = make_float2(ok ? x[i] : 0.f, ok ? x[i + 1] : 0.f);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int s4 = 0; s4 < 16; ++s4) {
            const float* r = T + (4 * s4 + (lane >> 4)) * MT_PITCH + (lane & 15);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(r[0], r[16], acc, 0, 0, 0);
        }
        asm volatile("" ::: "memory");
    }
    // dW0: the four waves' 16 x 16 sums through LDS; dW1: wave reduction of the 16 per-thread sums; one atomic per value per workgroup
    {
        const int i = lane & 15, g = lane >> 4;
        if (i < I) {
#pragma unroll
            for (int r = 0; r < 4; ++r) part[wave][(4 * g + r) * I + i] = acc[r];
        }
    }
#pragma unroll
    for (int j = 0; j < H; ++j) {
        const float s1 = wave_sum(a1[j]);
        if (lane == 0) part[wave][H * I + j] = s1;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < NW; e += 256) {
        const float s = part[0][e] + part[1][e] + part[2][e] + part[3][e];
        if (e < H * I) unsafeAtomicAdd(&dW0[e], s);
        else unsafeAtomicAdd(&dW1[e - H * I], s);
    }
}

}  // namespace snf

using namespace snf;

extern "C" int snf_mlp_tiny_supported(int I, int H, int O) { return (I == 10 && H == 16 && O == 1) ? 1 : 0; }

extern "C" int snf_mlp_tiny_fwd(const float* X, int ldx, const float* W0, const float* W1, int I, int H, int64_t N,
                                float* Hid, float* Y, snf_stream_t stream) {
    SNF_REQUIRE(X && W0 && W1 && Y && N > 0, "snf_mlp_tiny_fwd: bad argument");
    SNF_REQUIRE(snf_mlp_tiny_supported(I, H, 1), "snf_mlp_tiny_fwd: only the 10 -> 16 -> 1 proposal net is built (I=%d H=%d)", I, H);
    SNF_REQUIRE(ldx >= I && ldx % 2 == 0 && ((uintptr_t)X % 8) == 0 && (!Hid || ((uintptr_t)Hid % 16) == 0),
                "snf_mlp_tiny_fwd: X rows must be 8-byte aligned (ldx even), Hid 16-byte aligned");
    const unsigned blocks = (unsigned)((N + 255) / 256);
    hipLaunchKernelGGL((k_mlp_tiny_fwd<10, 16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, X, ldx, W0, W1, (long long)N,
                       Hid, Y);
    SNF_LAUNCH_CHECK("snf_mlp_tiny_fwd");
    return SNF_OK;
}

extern "C" int snf_mlp_tiny_bwd(const float* dY, const float* X, int ldx, const float* Hid, const float* W0, const float* W1,
                                int I, int H, int64_t N, float* dX, int lddx, float* dW0, float* dW1, snf_stream_t stream) {
    SNF_REQUIRE(dY && X && Hid && W0 && W1 && dW0 && dW1 && N > 0, "snf_mlp_tiny_bwd: bad argument");
    SNF_REQUIRE(snf_mlp_tiny_supported(I, H, 1), "snf_mlp_tiny_bwd: only the 10 -> 16 -> 1 proposal net is built (I=%d H=%d)", I, H);
    SNF_REQUIRE(ldx >= I && ldx % 2 == 0 && ((uintptr_t)X % 8) == 0 && ((uintptr_t)Hid % 16) == 0,
                "snf_mlp_tiny_bwd: X rows must be 8-byte aligned (ldx even), Hid 16-byte aligned");
    SNF_REQUIRE(!dX || (lddx >= I && lddx % 2 == 0 && ((uintptr_t)dX % 8) == 0), "snf_mlp_tiny_bwd: bad dX layout");
    const unsigned blocks = (unsigned)((N + 256 * MT_SPT - 1) / (256 * MT_SPT));
    hipLaunchKernelGGL((k_mlp_tiny_bwd<10, 16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, dY, X, ldx, Hid, W0, W1,
                       (long long)N, dX, lddx, dW0, dW1);
    SNF_LAUNCH_CHECK("snf_mlp_tiny_bwd");
    return SNF_OK;
}

// mlp_tiny.hip -- the proposal networks' density MLP (tcnn FullyFusedMLP with 16 neurons, one hidden layer, one output:
// nerfstudio/fields/density_fields.py:80-97 with the samnerf proposal_net_args hidden_dim = 16; input = the 5-level F=2 hash
// encoding, 10 wide) in one launch per direction (gfx950).
//
// 176 multiply-adds per sample is far below the point where a matrix-core tile pays: one thread per sample on the vector
// ALU, weights broadcast from LDS.  The backward keeps the full weight gradient (H*I + H values) in per-thread registers
// over the samples it visits, reduces them across the wave with shuffles and across the workgroup through LDS, and issues
// ONE set of global atomics per workgroup -- the three GEMM launches + two wgrad launches this replaces spent most of their
// time on 10-wide unaligned rows.
#include "common.hpp"

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
    float y = 0.f;
    float h[H];
#pragma unroll
    for (int j = 0; j < H; ++j) {
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < I; ++i) a += w0[j * I + i] * x[i];
        h[j] = fmaxf(a, 0.f);
        y += w1[j] * h[j];
    }
    if (Hid != nullptr) {
#pragma unroll
        for (int j = 0; j < H; j += 4)
            *reinterpret_cast<float4*>(Hid + n * H + j) = make_float4(h[j], h[j + 1], h[j + 2], h[j + 3]);
    }
    Y[n] = y;
}

template <int I, int H>
__global__ __launch_bounds__(256) void k_mlp_tiny_bwd(const float* __restrict__ dY, const float* __restrict__ X, int ldx,
                                                      const float* __restrict__ Hid, const float* __restrict__ W0,
                                                      const float* __restrict__ W1, long long N, float* __restrict__ dX,
                                                      int lddx, float* __restrict__ dW0, float* __restrict__ dW1) {
    constexpr int NW = H * I + H;
    __shared__ float w0[H * I], w1[H];
    __shared__ float part[4][NW];
    for (int i = threadIdx.x; i < H * I; i += 256) w0[i] = W0[i];
    if (threadIdx.x < H) w1[threadIdx.x] = W1[threadIdx.x];
    __syncthreads();
    float a0[H][I], a1[H];
#pragma unroll
    for (int j = 0; j < H; ++j) {
        a1[j] = 0.f;
#pragma unroll
        for (int i = 0; i < I; ++i) a0[j][i] = 0.f;
    }
    for (int sidx = 0; sidx < MT_SPT; ++sidx) {
        const long long n = ((long long)blockIdx.x * MT_SPT + sidx) * 256 + threadIdx.x;
        if (n < N) {
            const float dy = dY[n];
            float x[I], dx[I];
            const float* xp = X + n * ldx;
#pragma unroll
            for (int i = 0; i < I; i += 2) {
                const float2 t = *reinterpret_cast<const float2*>(xp + i);
                x[i] = t.x; x[i + 1] = t.y;
                dx[i] = 0.f; dx[i + 1] = 0.f;
            }
#pragma unroll
            for (int j4 = 0; j4 < H; j4 += 4) {
                const float4 hv = *reinterpret_cast<const float4*>(Hid + n * H + j4);
                const float h4[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int j = j4 + q;
                    a1[j] += dy * h4[q];
                    const float dh = h4[q] > 0.f ? dy * w1[j] : 0.f;
#pragma unroll
                    for (int i = 0; i < I; ++i) {
                        a0[j][i] += dh * x[i];
                        dx[i] += dh * w0[j * I + i];
                    }
                }
            }
            if (dX != nullptr) {
#pragma unroll
                for (int i = 0; i < I; i += 2) *reinterpret_cast<float2*>(dX + n * lddx + i) = make_float2(dx[i], dx[i + 1]);
            }
        }
    }
    // wave reduction of the NW accumulators, then the four waves through LDS, then one atomic per value per workgroup
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < H; ++j) {
#pragma unroll
        for (int i = 0; i < I; ++i) {
            const float s = wave_sum(a0[j][i]);
            if (lane == 0) part[wave][j * I + i] = s;
        }
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

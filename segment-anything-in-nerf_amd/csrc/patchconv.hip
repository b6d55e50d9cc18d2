// patchconv.hip -- the SAM conv head (samnerf/sam_model.py:196-200,259-264: Conv2d(C,C,k,pad) -> ReLU -> Conv2d -> mean over
// the p x p patch) as GEMMs on the library's own matrix-core kernels (gfx950).
//
// The head sees R/p^2 independent p x p "images" (p = 4).  Features stay CHANNEL-LAST, exactly as MeanRenderer leaves them:
// x[row, c], row = patch*p*p + y*p + x.  With zero padding a k x k convolution of such a patch is one GEMM over the
// unfolded rows,
//     col[row, c*k*k + t] = x[patch, y + dy_t, x + dx_t, c]   (0 outside the patch),   t = ky*k + kx, dy = ky - k/2,
//     conv(x)[row, o] = sum_j col[row, j] * W[o, j] + b[o],    W = Conv2d.weight viewed [O, C*k*k]  (no copy: the torch
// weight layout [O, C, k, k] flattens to exactly this column order).
// The second convolution is followed by the mean over the patch, and both are linear, so the mean moves in front of the
// GEMM: mean_rows(col) is a [R/p^2, C*k*k] matrix whose entry (c, t) is 1/p^2 times the sum of h over the sub-rectangle of
// the patch that tap t can see -- the second GEMM runs on p^2 = 16x fewer rows.
//
//   snf_patch_unfold        x [R, C]            -> col  [R, C*k*k]
//   snf_patch_fold          dcol [R, C*k*k]     -> dx   [R, C]            (adjoint of unfold)
//   snf_patch_unfold_mean   h [R, C]            -> cm   [R/p^2, C*k*k]
//   snf_patch_fold_mean     dcm [R/p^2, C*k*k]  -> dh   [R, C]            (adjoint of unfold_mean)
// All four are pure data movement (HBM/LDS bound); the arithmetic is snf_linear_fwd / _bwd_data / _bwd_weight.
#include "common.hpp"

namespace snf {

constexpr int PC_MAX_KK = 25;  // k <= 5

// All four kernels are templated on the kernel size (KT = 3: the head's; 0: k at run time): with k, k*k and C as run-time divisors every
// element paid two or three integer divisions (~25 VALU instructions each) -- the "pure data movement" was 70-450 instructions per
// element and the four launches took 0.09 ms of the SAM stream's serial chain.  Row / tap loops are nested instead of divided out, the
// channel loop is the lanes', and every sum keeps its order (bit-identical results).

// One workgroup per (patch, image row of the patch): the patch's p*p rows of x [C] go into LDS (pitch C + 1: the transposing reads below
// hit distinct banks), then the C*k*k outputs of the workgroup's p pixels leave as contiguous streams.
template <int KT>
__global__ __launch_bounds__(256) void k_patch_unfold(const float* __restrict__ x, int p, int C, int krt, float* __restrict__ col) {
    extern __shared__ float nb[];  // [p*p][C + 1]
    const int k = KT > 0 ? KT : krt, kk = k * k, h = k / 2, pp = p * p, CP = C + 1;
    const int patch = blockIdx.x;
    const float* __restrict__ xp = x + (size_t)patch * pp * C;
    for (int r = 0; r < pp; ++r)
        for (int c = threadIdx.x; c < C; c += 256) nb[r * CP + c] = xp[(size_t)r * C + c];
    __syncthreads();
    const int n = C * kk;
    // blockIdx.y: the image row of the patch this workgroup writes (p pixels); every workgroup stages the whole patch (L2 hits)
    for (int r = blockIdx.y * p; r < (int)(blockIdx.y + 1) * p; ++r) {
        const int y = r / p, xx = r - y * p;
        float* __restrict__ o = col + ((size_t)patch * pp + r) * n;
        for (int e = threadIdx.x; e < n; e += 256) {
            const int c = e / kk, t = e - c * kk;  // (KT > 0: a multiply and a shift)
            const int ty = t / k, iy = y + ty - h, ix = xx + (t - ty * k) - h;
            o[e] = (iy >= 0 && iy < p && ix >= 0 && ix < p) ? nb[(iy * p + ix) * CP + c] : 0.f;
        }
    }
}

// The same for patches too large for LDS (the image encoder's neck: one 64 x 64 "patch"): one workgroup per output ROW, its k*k
// neighbour rows read coalesced into LDS (pitch C + 1), the C*k*k outputs leave as one contiguous stream.
template <int KT>
__global__ __launch_bounds__(256) void k_patch_unfold_row(const float* __restrict__ x, int p, int C, int krt, float* __restrict__ col) {
    extern __shared__ float nb[];  // [k*k][C + 1]
    const int k = KT > 0 ? KT : krt, kk = k * k, h = k / 2, pp = p * p, CP = C + 1;
    const int row = blockIdx.x;
    const int patch = row / pp, y = (row - patch * pp) / p, xx = row - patch * pp - y * p;
    for (int t = 0; t < kk; ++t) {
        const int ty = t / k, iy = y + ty - h, ix = xx + (t - ty * k) - h;
        const bool in = iy >= 0 && iy < p && ix >= 0 && ix < p;  // workgroup-uniform
        const float* __restrict__ src = x + ((size_t)patch * pp + (in ? iy * p + ix : 0)) * C;
        for (int c = threadIdx.x; c < C; c += 256) nb[t * CP + c] = in ? src[c] : 0.f;
    }
    __syncthreads();
    const int n = C * kk;
    float* __restrict__ o = col + (size_t)row * n;
    for (int e = threadIdx.x; e < n; e += 256) {
        const int c = e / kk, t = e - c * kk;
        o[e] = nb[t * CP + c];
    }
}

// One workgroup per (patch, chunk of CC channels): the chunk's columns of all p*p rows of dcol are read as contiguous
// runs into LDS; each thread then sums the <= k*k taps of its outputs.
constexpr int PC_CC = 32;
template <int KT>
__global__ __launch_bounds__(256) void k_patch_fold(const float* __restrict__ dcol, int p, int C, int krt, float* __restrict__ dx) {
    extern __shared__ float sm[];  // [p*p][CC*k*k]
    const int k = KT > 0 ? KT : krt, kk = k * k, h = k / 2, pp = p * p;
    const int patch = blockIdx.x, c0 = blockIdx.y * PC_CC;
    const int cc = min(PC_CC, C - c0), run = cc * kk;
    for (int r = 0; r < pp; ++r) {
        const float* __restrict__ src = dcol + ((size_t)patch * pp + r) * C * kk + (size_t)c0 * kk;
        for (int j = threadIdx.x; j < run; j += 256) sm[r * (PC_CC * kk) + j] = src[j];
    }
    __syncthreads();
    // thread -> (row r = e / CC, channel c = e % CC): CC = 32 is a constant, the lanes of a half-wave take the 32 channels of one row
    for (int e = threadIdx.x; e < pp * PC_CC; e += 256) {
        const int r = e / PC_CC, c = e - r * PC_CC;
        if (c >= cc) continue;
        const int y = r / p, xx = r - y * p;
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < (KT > 0 ? KT * KT : 1); ++t) {
            if constexpr (KT > 0) {
                // output pixel (oy, ox) read input (y, xx) through tap t  <=>  oy = y - dy_t, ox = xx - dx_t
                const int oy = y - (t / KT - KT / 2), ox = xx - (t % KT - KT / 2);
                if (oy >= 0 && oy < p && ox >= 0 && ox < p) acc += sm[(oy * p + ox) * (PC_CC * kk) + c * kk + t];
            }
        }
        if constexpr (KT == 0) {
            for (int t = 0; t < kk; ++t) {
                const int oy = y - (t / k - h), ox = xx - (t % k - h);
                if (oy >= 0 && oy < p && ox >= 0 && ox < p) acc += sm[(oy * p + ox) * (PC_CC * kk) + c * kk + t];
            }
        }
        dx[((size_t)patch * pp + r) * C + c0 + c] = acc;
    }
}

// One workgroup per patch: h[p*p][C] into LDS, then cm[c*k*k + t] = 1/p^2 * sum of h over the rectangle tap t can see.
template <int KT>
__global__ __launch_bounds__(256) void k_patch_unfold_mean(const float* __restrict__ hh, int p, int C, int krt,
                                                           float* __restrict__ cm) {
    extern __shared__ float sm[];  // [p*p][C]
    const int k = KT > 0 ? KT : krt, kk = k * k, h = k / 2, pp = p * p;
    const int patch = blockIdx.x;
    for (int e = threadIdx.x; e < pp * C; e += 256) sm[e] = hh[(size_t)patch * pp * C + e];
    __syncthreads();
    const float inv = 1.f / (float)pp;
    float* __restrict__ o = cm + (size_t)patch * C * kk;
    for (int e = threadIdx.x; e < C * kk; e += 256) {
        const int c = e / kk, t = e - c * kk;
        const int ty = t / k, dy = ty - h, dx = (t - ty * k) - h;
        // output pixel (y, x) reads input (y + dy, x + dx): inputs iy in [max(0,dy), min(p, p+dy))
        float acc = 0.f;
        for (int iy = max(0, dy); iy < min(p, p + dy); ++iy)
            for (int ix = max(0, dx); ix < min(p, p + dx); ++ix) acc += sm[(iy * p + ix) * C + c];
        o[e] = acc * inv;
    }
}

// One workgroup per patch: dcm[C*k*k] into LDS, then dh[(y,x), c] = 1/p^2 * sum over the taps whose rectangle holds (y,x).
template <int KT>
__global__ __launch_bounds__(256) void k_patch_fold_mean(const float* __restrict__ dcm, int p, int C, int krt,
                                                         float* __restrict__ dh) {
    extern __shared__ float sm[];  // [C*k*k]
    const int k = KT > 0 ? KT : krt, kk = k * k, h = k / 2, pp = p * p;
    const int patch = blockIdx.x;
    for (int e = threadIdx.x; e < C * kk; e += 256) sm[e] = dcm[(size_t)patch * C * kk + e];
    __syncthreads();
    const float inv = 1.f / (float)pp;
    for (int r = 0; r < pp; ++r) {
        const int iy = r / p, ix = r - iy * p;
        for (int c = threadIdx.x; c < C; c += 256) {
            float acc = 0.f;
            if constexpr (KT > 0) {
#pragma unroll
                for (int t = 0; t < KT * KT; ++t) {
                    const int dy = t / KT - KT / 2, dx = t % KT - KT / 2;
                    if (iy >= max(0, dy) && iy < min(p, p + dy) && ix >= max(0, dx) && ix < min(p, p + dx)) acc += sm[c * kk + t];
                }
            } else {
                for (int t = 0; t < kk; ++t) {
                    const int dy = t / k - h, dx = t % k - h;
                    if (iy >= max(0, dy) && iy < min(p, p + dy) && ix >= max(0, dx) && ix < min(p, p + dx)) acc += sm[c * kk + t];
                }
            }
            dh[((size_t)patch * pp + r) * C + c] = acc * inv;
        }
    }
}

}  // namespace snf

using namespace snf;

static int check_patch(const char* who, const void* a, const void* b, int R, int p, int C, int k) {
    SNF_REQUIRE(a && b, "%s: null pointer", who);
    SNF_REQUIRE(p >= 1 && p <= 64 && k >= 1 && k <= 5 && (k & 1) && C >= 1, "%s: bad patch=%d kernel=%d C=%d", who, p, k, C);
    SNF_REQUIRE(R > 0 && R % (p * p) == 0, "%s: R=%d is not a multiple of patch^2=%d", who, R, p * p);
    return SNF_OK;
}

extern "C" int snf_patch_unfold(const float* x, int R, int p, int C, int k, float* col, snf_stream_t stream) {
    int rc = check_patch("snf_patch_unfold", x, col, R, p, C, k);
    if (rc) return rc;
    const size_t lds = (size_t)p * p * (C + 1) * sizeof(float);
    if (lds <= 48 * 1024) {  // the patch fits LDS: one workgroup per (patch, image row)
        if (k == 3) hipLaunchKernelGGL(k_patch_unfold<3>, dim3(R / (p * p), p), dim3(256), lds, (hipStream_t)stream, x, p, C, k, col);
        else hipLaunchKernelGGL(k_patch_unfold<0>, dim3(R / (p * p), p), dim3(256), lds, (hipStream_t)stream, x, p, C, k, col);
    } else {
        const size_t lds_row = (size_t)k * k * (C + 1) * sizeof(float);
        SNF_REQUIRE(lds_row <= 64 * 1024, "snf_patch_unfold: (C + 1)*k*k floats exceed the 64 KB row staging");
        if (k == 3) hipLaunchKernelGGL(k_patch_unfold_row<3>, dim3(R), dim3(256), lds_row, (hipStream_t)stream, x, p, C, k, col);
        else hipLaunchKernelGGL(k_patch_unfold_row<0>, dim3(R), dim3(256), lds_row, (hipStream_t)stream, x, p, C, k, col);
    }
    SNF_LAUNCH_CHECK("snf_patch_unfold");
    return SNF_OK;
}

extern "C" int snf_patch_fold(const float* dcol, int R, int p, int C, int k, float* dx, snf_stream_t stream) {
    int rc = check_patch("snf_patch_fold", dcol, dx, R, p, C, k);
    if (rc) return rc;
    const size_t lds = (size_t)p * p * PC_CC * k * k * sizeof(float);
    SNF_REQUIRE(lds <= 64 * 1024, "snf_patch_fold: patch too large for the fold kernel (p <= 8 at k = 3)");
    const dim3 grid(R / (p * p), (C + PC_CC - 1) / PC_CC);
    if (k == 3) hipLaunchKernelGGL(k_patch_fold<3>, grid, dim3(256), lds, (hipStream_t)stream, dcol, p, C, k, dx);
    else hipLaunchKernelGGL(k_patch_fold<0>, grid, dim3(256), lds, (hipStream_t)stream, dcol, p, C, k, dx);
    SNF_LAUNCH_CHECK("snf_patch_fold");
    return SNF_OK;
}

extern "C" int snf_patch_unfold_mean(const float* h, int R, int p, int C, int k, float* cm, snf_stream_t stream) {
    int rc = check_patch("snf_patch_unfold_mean", h, cm, R, p, C, k);
    if (rc) return rc;
    const size_t lds = (size_t)p * p * C * sizeof(float);
    SNF_REQUIRE(lds <= 64 * 1024, "snf_patch_unfold_mean: patch^2*C too large");
    if (k == 3) hipLaunchKernelGGL(k_patch_unfold_mean<3>, dim3(R / (p * p)), dim3(256), lds, (hipStream_t)stream, h, p, C, k, cm);
    else hipLaunchKernelGGL(k_patch_unfold_mean<0>, dim3(R / (p * p)), dim3(256), lds, (hipStream_t)stream, h, p, C, k, cm);
    SNF_LAUNCH_CHECK("snf_patch_unfold_mean");
    return SNF_OK;
}

extern "C" int snf_patch_fold_mean(const float* dcm, int R, int p, int C, int k, float* dh, snf_stream_t stream) {
    int rc = check_patch("snf_patch_fold_mean", dcm, dh, R, p, C, k);
    if (rc) return rc;
    const size_t lds = (size_t)C * k * k * sizeof(float);
    SNF_REQUIRE(lds <= 64 * 1024, "snf_patch_fold_mean: C*k*k too large");
    if (k == 3) hipLaunchKernelGGL(k_patch_fold_mean<3>, dim3(R / (p * p)), dim3(256), lds, (hipStream_t)stream, dcm, p, C, k, dh);
    else hipLaunchKernelGGL(k_patch_fold_mean<0>, dim3(R / (p * p)), dim3(256), lds, (hipStream_t)stream, dcm, p, C, k, dh);
    SNF_LAUNCH_CHECK("snf_patch_fold_mean");
    return SNF_OK;
}

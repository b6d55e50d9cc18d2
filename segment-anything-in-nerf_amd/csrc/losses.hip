// losses.hip -- the losses of the path (gfx950): the two per-ray regularisers of the nerfacto loss, forward + gradient in one
// pass, and the L2 rendering / distillation losses (snf_rowmse_loss_*).
//   snf_interlevel : interlevel_loss / lossfun_outer / outer   (model_components/losses.py:46-120)
//   snf_distortion : distortion_loss / lossfun_distortion      (model_components/losses.py:124-143)
// One wavefront per ray; bins, weights and the difference array live in LDS.
#include "common.hpp"

namespace snf {

constexpr int RAYS_PER_BLOCK = 4;
constexpr int MAXN = 260;  // S, P <= 256 (+1 bin edge, + slack)

// searchsorted(a[0..n), v, side='right'): number of elements <= v
__device__ __forceinline__ int upper_bound(const float* a, int n, float v) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] <= v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void k_interlevel(const float* __restrict__ sb_f, const float* __restrict__ w_f,
                                                    const float* __restrict__ sb_p, const float* __restrict__ w_p, int R,
                                                    int S, int P, float grad_scale, float* __restrict__ loss_rows,
                                                    float* __restrict__ grad_wp) {
    __shared__ float s_cp[RAYS_PER_BLOCK][MAXN];    // proposal bin edges [P+1]
    __shared__ float s_cy[RAYS_PER_BLOCK][MAXN];    // [0, cumsum(w_p)]  [P+1]
    __shared__ float s_diff[RAYS_PER_BLOCK][MAXN];  // difference array for d loss / d w_p
    const int wv = threadIdx.x >> 6;
    const int r_raw = blockIdx.x * RAYS_PER_BLOCK + wv;
    const bool active = r_raw < R;
    const int r = active ? r_raw : R - 1;
    const int lane = lane_id();
    float* cp = s_cp[wv];
    float* cy = s_cy[wv];
    float* df = s_diff[wv];
    for (int i = lane; i <= P; i += WAVE) { cp[i] = sb_p[(size_t)r * (P + 1) + i]; df[i] = 0.f; }
    float carry = 0.f;
    for (int base = 0; base < P; base += WAVE) {
        const int i = base + lane;
        const float v = (i < P) ? w_p[(size_t)r * P + i] : 0.f;
        const float inc = wave_incl_scan(v) + carry;
        if (i < P) cy[i + 1] = inc;
        carry = __shfl(inc, WAVE - 1, WAVE);
    }
    if (lane == 0) cy[0] = 0.f;
    __syncthreads();
    float part = 0.f;
    for (int s = lane; s < S; s += WAVE) {
        const float t0 = sb_f[(size_t)r * (S + 1) + s];
        const float t1 = sb_f[(size_t)r * (S + 1) + s + 1];
        const float w = w_f[(size_t)r * S + s];
        int lo = upper_bound(cp, P, t0) - 1;      // searchsorted(t1_starts = cp[0..P), t0, right) - 1
        lo = min(max(lo, 0), P - 1);
        int hi = upper_bound(cp + 1, P, t1);      // searchsorted(t1_ends = cp[1..P], t1, right)
        hi = min(max(hi, 0), P - 1);
        const float w_outer = cy[hi + 1] - cy[lo];
        const float d = fmaxf(w - w_outer, 0.f);
        part += d * d / (w + 1e-7f);
        if (grad_wp && d > 0.f) {
            const float g = -2.f * d / (w + 1e-7f) * grad_scale;  // d/d w_outer
            atomicAdd(&df[lo], g);
            atomicAdd(&df[hi + 1], -g);
        }
    }
    part = wave_sum(part);
    if (active && lane == 0) loss_rows[r] = part;
    __syncthreads();
    if (grad_wp) {
        float c2 = 0.f;
        for (int base = 0; base < P; base += WAVE) {
            const int i = base + lane;
            const float v = (i < P) ? df[i] : 0.f;
            const float inc = wave_incl_scan(v) + c2;
            if (active && i < P) grad_wp[(size_t)r * P + i] = inc;
            c2 = __shfl(inc, WAVE - 1, WAVE);
        }
    }
}

__global__ __launch_bounds__(256) void k_distortion(const float* __restrict__ sb, const float* __restrict__ w_in, int R,
                                                    int S, float grad_scale, float* __restrict__ loss_rows,
                                                    float* __restrict__ grad_w) {
    __shared__ float s_u[RAYS_PER_BLOCK][MAXN];
    __shared__ float s_w[RAYS_PER_BLOCK][MAXN];
    const int wv = threadIdx.x >> 6;
    const int r_raw = blockIdx.x * RAYS_PER_BLOCK + wv;
    const bool active = r_raw < R;
    const int r = active ? r_raw : R - 1;
    const int lane = lane_id();
    float* u = s_u[wv];
    float* w = s_w[wv];
    const float* t = sb + (size_t)r * (S + 1);
    for (int i = lane; i < S; i += WAVE) {
        u[i] = (t[i + 1] + t[i]) / 2.f;
        w[i] = w_in[(size_t)r * S + i];
    }
    __syncthreads();
    float part = 0.f;
    for (int i = lane; i < S; i += WAVE) {
        const float ui = u[i], wi = w[i];
        float inner = 0.f;
        for (int j = 0; j < S; ++j) inner += w[j] * fabsf(ui - u[j]);
        const float width = t[i + 1] - t[i];
        part += wi * inner + wi * wi * width / 3.f;
        if (active && grad_w) grad_w[(size_t)r * S + i] = (2.f * inner + 2.f * wi * width / 3.f) * grad_scale;
    }
    part = wave_sum(part);
    if (active && lane == 0) loss_rows[r] = part;
}

// ---- a19: the L2 (distillation) losses.  nn.MSELoss()(image, rgb)              = mean over all elements
//           w * mse_loss(pred, target, 'none').mean(-1).nanmean()      = mean over the rows whose own mean is not NaN
// (samnerf/sam_model.py:316-328, nerfstudio/models/nerfacto.py:326-333).  One wavefront per row; workgroup partials are
// added to two device scalars, and the LAST workgroup to arrive (atomic ticket) turns them into {loss, row count} -- one
// launch, no host involvement.  `acc` is a scratch buffer of SNF_ROWMSE_SCRATCH_WORDS words whose ticket word (index
// 2 * 256) must be zero on entry; it is left zeroed again.
constexpr int RM_MAX_BLOCKS = 256;

__global__ __launch_bounds__(256) void k_rowmse_fwd(const float* __restrict__ pred, const float* __restrict__ target, int R,
                                                    int C, float weight, int nan_skip, float* __restrict__ acc,
                                                    float* __restrict__ out) {
    __shared__ float s_sum[4], s_cnt[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float wsum = 0.f, wcnt = 0.f;
    // four rows per wave and trip: their loads are independent, so one memory latency covers all four
    for (int r0 = (blockIdx.x * 4 + wave) * 4; r0 < R; r0 += gridDim.x * 16) {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = lane; c < C; c += 64) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = min(r0 + q, R - 1);
                const float d = pred[(size_t)r * C + c] - target[(size_t)r * C + c];
                a[q] += d * d;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float m = wave_sum(a[q]) / (float)C;  // the row's mean squared error
            const bool skip = (r0 + q >= R) || (nan_skip && (m != m));
            wsum += skip ? 0.f : m;
            wcnt += skip ? 0.f : 1.f;
        }
    }
    if (lane == 0) { s_sum[wave] = wsum; s_cnt[wave] = wcnt; }
    __syncthreads();
    __shared__ bool last;
    if (threadIdx.x == 0) {
        // per-workgroup partials in their own slots (no same-address float atomics); only the ticket is shared
        acc[blockIdx.x] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
        acc[RM_MAX_BLOCKS + blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        __threadfence();
        const unsigned ticket = atomicAdd(reinterpret_cast<unsigned*>(&acc[2 * RM_MAX_BLOCKS]), 1u);
        last = ticket == gridDim.x - 1;
    }
    __syncthreads();
    if (last && wave == 0) {  // every partial is in: the last workgroup's first wave finishes and re-arms the ticket
        __threadfence();
        float t = 0.f, c = 0.f;
        for (int i = lane; i < (int)gridDim.x; i += 64) {
            t += __builtin_nontemporal_load(&acc[i]);
            c += __builtin_nontemporal_load(&acc[RM_MAX_BLOCKS + i]);
        }
        t = wave_sum(t);
        c = wave_sum(c);
        if (lane == 0) {
            atomicExch(reinterpret_cast<unsigned*>(&acc[2 * RM_MAX_BLOCKS]), 0u);
            out[0] = weight * (t / c);  // 0/0 = NaN when every row is skipped, as torch.nanmean does
            out[1] = c;
        }
    }
}

// dpred[r][c] = gout * weight * 2 (pred - target) / (C * count), zero for the rows nanmean skipped (0 * NaN stays NaN, as in
// autograd).  `gout` and `count` are device scalars.
__global__ __launch_bounds__(256) void k_rowmse_bwd(const float* __restrict__ pred, const float* __restrict__ target, int R,
                                                    int C, float weight, int nan_skip, const float* __restrict__ gout,
                                                    const float* __restrict__ out, float* __restrict__ dpred) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float k = gout[0] * weight * 2.f / ((float)C * out[1]);
    for (int r = blockIdx.x * 4 + wave; r < R; r += gridDim.x * 4) {
        const float* p = pred + (size_t)r * C;
        const float* t = target + (size_t)r * C;
        float* d = dpred + (size_t)r * C;
        float kk = k;
        if (nan_skip) {  // the forward's row test, recomputed
            float a = 0.f;
            for (int c = lane; c < C; c += 64) {
                const float e = p[c] - t[c];
                a += e * e;
            }
            a = wave_sum(a);
            if (a != a) kk = 0.f;
        }
        for (int c = lane; c < C; c += 64) d[c] = kk * (p[c] - t[c]);
    }
}


// y += alpha * x, the product rounded before the sum (what autograd does when it scales a branch's gradient and then adds
// the branches: mul, then add -- no fused multiply-add)
__global__ __launch_bounds__(256) void k_add_scaled(long long n, float alpha, const float* __restrict__ x,
                                                    float* __restrict__ y) {
#pragma clang fp contract(off)
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const float t = alpha * x[i];
        y[i] = y[i] + t;
    }
}

// Scalar tail of NerfactoModel.get_loss_dict / get_metrics_dict (nerfstudio/models/nerfacto.py:316-344) in one launch:
//   interlevel_loss = mult_i * sum(rows_i) / n_i ; distortion = sum(rows_d) / R ; distortion_loss = mult_d * distortion ;
//   psnr = -10 log10(rgb_mse) ; total = rgb_loss + interlevel_loss + distortion_loss
// rows_i may be NULL (eval / no proposal level).  out: {total, rgb_loss, interlevel_loss, distortion_loss, distortion, psnr}.
__global__ __launch_bounds__(1024) void k_nerf_loss_summary(const float* __restrict__ rgb_mse,
                                                            const float* __restrict__ rows_i, float scale_i,
                                                            const float* __restrict__ rows_d, float scale_d, float mult_d,
                                                            int R, float* __restrict__ out) {
    __shared__ float si[16], sd[16];
    float a = 0.f, b = 0.f;
    for (int r = threadIdx.x; r < R; r += 1024) {
        if (rows_i) a += rows_i[r];
        b += rows_d[r];
    }
    a = wave_sum(a);
    b = wave_sum(b);
    if ((threadIdx.x & 63) == 0) { si[threadIdx.x >> 6] = a; sd[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float ta = 0.f, tb = 0.f;
        for (int w = 0; w < 16; ++w) { ta += si[w]; tb += sd[w]; }
        const float rgb = rgb_mse[0];
        const float inter = ta * scale_i, dist = tb * scale_d, dloss = mult_d * dist;
        out[0] = rgb + inter + dloss;
        out[1] = rgb;
        out[2] = inter;
        out[3] = dloss;
        out[4] = dist;
        out[5] = -10.f * log10f(rgb);
    }
}

}  // namespace snf

using namespace snf;

extern "C" int snf_interlevel(const float* sbins_fine, const float* w_fine, const float* sbins_prop, const float* w_prop,
                              int R, int S, int P, float grad_scale, float* loss_rows, float* grad_w_prop,
                              snf_stream_t stream) {
    SNF_REQUIRE(sbins_fine && w_fine && sbins_prop && w_prop && loss_rows, "snf_interlevel: null pointer");
    SNF_REQUIRE(R > 0 && S >= 1 && P >= 1 && S <= 256 && P <= 256, "snf_interlevel: bad shape R=%d S=%d P=%d", R, S, P);
    hipLaunchKernelGGL(k_interlevel, dim3(ceil_div(R, RAYS_PER_BLOCK)), dim3(256), 0, (hipStream_t)stream, sbins_fine,
                       w_fine, sbins_prop, w_prop, R, S, P, grad_scale, loss_rows, grad_w_prop);
    SNF_LAUNCH_CHECK("snf_interlevel");
    return SNF_OK;
}

extern "C" int snf_distortion(const float* sbins, const float* w, int R, int S, float grad_scale, float* loss_rows,
                              float* grad_w, snf_stream_t stream) {
    SNF_REQUIRE(sbins && w && loss_rows, "snf_distortion: null pointer");
    SNF_REQUIRE(R > 0 && S >= 1 && S <= 256, "snf_distortion: bad shape R=%d S=%d", R, S);
    hipLaunchKernelGGL(k_distortion, dim3(ceil_div(R, RAYS_PER_BLOCK)), dim3(256), 0, (hipStream_t)stream, sbins, w, R,
                       S, grad_scale, loss_rows, grad_w);
    SNF_LAUNCH_CHECK("snf_distortion");
    return SNF_OK;
}

extern "C" int snf_rowmse_loss_fwd(const float* pred, const float* target, int R, int C, float weight, int nan_skip,
                                   float* acc, float* out, snf_stream_t stream) {
    SNF_REQUIRE(pred && target && acc && out && R > 0 && C > 0, "snf_rowmse_loss_fwd: bad argument");
    // one trip of four rows per wave where possible (the loss sits on the step's critical chain: latency matters, not bytes)
    int blocks = ceil_div(R, 16);
    if (blocks > RM_MAX_BLOCKS) blocks = RM_MAX_BLOCKS;
    hipLaunchKernelGGL(k_rowmse_fwd, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pred, target, R, C, weight, nan_skip, acc,
                       out);
    SNF_LAUNCH_CHECK("snf_rowmse_loss_fwd");
    return SNF_OK;
}

extern "C" int snf_rowmse_loss_bwd(const float* pred, const float* target, int R, int C, float weight, int nan_skip,
                                   const float* gout, const float* out, float* dpred, snf_stream_t stream) {
    SNF_REQUIRE(pred && target && gout && out && dpred && R > 0 && C > 0, "snf_rowmse_loss_bwd: bad argument");
    int blocks = ceil_div(R, 4);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_rowmse_bwd, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pred, target, R, C, weight, nan_skip, gout,
                       out, dpred);
    SNF_LAUNCH_CHECK("snf_rowmse_loss_bwd");
    return SNF_OK;
}

extern "C" int snf_add_scaled(int64_t n, float alpha, const float* x, float* y, snf_stream_t stream) {
    SNF_REQUIRE(x && y && n > 0, "snf_add_scaled: bad argument");
    hipLaunchKernelGGL(k_add_scaled, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, (long long)n, alpha, x, y);
    SNF_LAUNCH_CHECK("snf_add_scaled");
    return SNF_OK;
}

extern "C" int snf_nerf_loss_summary(const float* rgb_mse, const float* interlevel_rows, float interlevel_scale,
                                     const float* distortion_rows, float distortion_scale, float distortion_mult, int R,
                                     float* out, snf_stream_t stream) {
    SNF_REQUIRE(rgb_mse && distortion_rows && out && R > 0, "snf_nerf_loss_summary: bad argument");
    hipLaunchKernelGGL(k_nerf_loss_summary, dim3(1), dim3(1024), 0, (hipStream_t)stream, rgb_mse, interlevel_rows,
                       interlevel_scale, distortion_rows, distortion_scale, distortion_mult, R, out);
    SNF_LAUNCH_CHECK("snf_nerf_loss_summary");
    return SNF_OK;
}

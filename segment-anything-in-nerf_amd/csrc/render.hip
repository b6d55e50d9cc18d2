// render.hip -- per-ray volumetric rendering kernels (gfx950): one 64-lane wavefront per ray, transmittance as a
// wavefront prefix scan.
//   snf_head_input        a13  SH degree-4 basis ++ geo features        (utils/math.py:27-73, nerfacto_field.py:336-343)
//   snf_weights_fwd/bwd   a8+a9 trunc_exp * selector, get_weights       (activations.py:24-40, cameras/rays.py:141-163)
//   snf_composite_fwd/bwd a14+a15 RGB 'last_sample', accumulation, median depth (renderers.py:97-140,222,260-270)
//   snf_feature_mean_*    a18  MeanRenderer                             (samnerf/sam_model.py:126-137)
#include "common.hpp"

namespace snf {

constexpr int RAYS_PER_BLOCK = 4;
constexpr int MAXC = 4;  // samples per lane: n <= 256

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_head_input(const float* __restrict__ dirs, const float* __restrict__ geo, int R,
                                                    int S, int n_geo, int ld_geo, float* __restrict__ out, int ld_out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)R * S) return;
    const int r = (int)(t / S);
    float sh[16];
    sh16_of(dirs[r * 3 + 0], dirs[r * 3 + 1], dirs[r * 3 + 2], sh);
    float* o = out + (size_t)t * ld_out;
#pragma unroll
    for (int j = 0; j < 16; ++j) o[j] = sh[j];
    const float* g = geo + (size_t)t * ld_geo;
    for (int j = 0; j < n_geo; ++j) o[16 + j] = g[j];
    for (int j = 16 + n_geo; j < ld_out; ++j) o[j] = 0.f;  // pad columns (e.g. 31 -> 32 for the fused MLP)
}

// The same for the fused colour net's input width (ld_out == 32): 8 lanes per sample, lane c writes columns 4c .. 4c+3 as one
// dwordx4, so a wave stores 8 whole 128-byte rows per instruction.  (One thread per sample writes 16 bytes into 64 different
// lines per instruction: rocprofv3 WRITE_SIZE showed 3x the algorithmic bytes for it.)
__global__ __launch_bounds__(256) void k_head_input32(const float* __restrict__ dirs, const float* __restrict__ geo, int R,
                                                      int S, int n_geo, int ld_geo, float* __restrict__ out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long s = t >> 3;
    const int c = (int)(t & 7);
    if (s >= (long long)R * S) return;
    float4 v;
    if (c < 4) {
        const int r = (int)(s / S);
        float sh[16];
        sh16_of(dirs[r * 3 + 0], dirs[r * 3 + 1], dirs[r * 3 + 2], sh);
        v = c == 0 ? make_float4(sh[0], sh[1], sh[2], sh[3]) : c == 1 ? make_float4(sh[4], sh[5], sh[6], sh[7])
            : c == 2 ? make_float4(sh[8], sh[9], sh[10], sh[11]) : make_float4(sh[12], sh[13], sh[14], sh[15]);
    } else {
        const float* g = geo + (size_t)s * ld_geo;
        const int j = (c - 4) * 4;
        v.x = j + 0 < n_geo ? g[j + 0] : 0.f;
        v.y = j + 1 < n_geo ? g[j + 1] : 0.f;
        v.z = j + 2 < n_geo ? g[j + 2] : 0.f;
        v.w = j + 3 < n_geo ? g[j + 3] : 0.f;
    }
    *reinterpret_cast<float4*>(out + (size_t)s * 32 + c * 4) = v;
}

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_weights_fwd(const float* __restrict__ raw, int raw_stride, int is_density,
                                                     const uint8_t* __restrict__ selector,
                                                     const float* __restrict__ ebins, int R, int n,
                                                     float* __restrict__ weights, float* __restrict__ density) {
    const int r = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (r >= R) return;
    const int lane = lane_id();
    const float* eb = ebins + (size_t)r * (n + 1);
    float carry = 0.f;
    for (int base = 0; base < n; base += WAVE) {
        const int i = base + lane;
        float dd = 0.f, sigma = 0.f;
        if (i < n) {
            const size_t s = (size_t)r * n + i;
            sigma = raw[s * raw_stride];
            if (!is_density) sigma = expf(sigma);
            if (selector) sigma *= (float)selector[s];
            dd = (eb[i + 1] - eb[i]) * sigma;
        }
        const float inc = wave_incl_scan(dd);
        const float excl = carry + shift_up1(inc);
        if (i < n) {
            const float alpha = 1.f - expf(-dd);
            const float T = expf(-excl);
            weights[(size_t)r * n + i] = nan_to_num(alpha * T);
            if (density) density[(size_t)r * n + i] = sigma;
        }
        carry += __shfl(inc, WAVE - 1, WAVE);
    }
}

// dL/ddd_k = g_k * T_k * exp(-dd_k) - sum_{i>k} g_i * alpha_i * T_i ;  then through sigma = exp(raw)*sel with the
// truncated-exp backward g * exp(clamp(raw,-15,15)).
__global__ __launch_bounds__(256) void k_weights_bwd(const float* __restrict__ raw, int raw_stride, int is_density,
                                                     const uint8_t* __restrict__ selector,
                                                     const float* __restrict__ ebins, const float* __restrict__ gw, int R,
                                                     int n, float* __restrict__ grad_raw) {
    const int r = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (r >= R) return;
    const int lane = lane_id();
    const float* eb = ebins + (size_t)r * (n + 1);
    float dd[MAXC], T[MAXC], gk[MAXC], delta[MAXC], rawv[MAXC], selv[MAXC], gwt[MAXC];
    float carry = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int i = c * WAVE + lane;
        dd[c] = 0.f; T[c] = 0.f; gk[c] = 0.f; delta[c] = 0.f; rawv[c] = 0.f; selv[c] = 0.f; gwt[c] = 0.f;
        if (c * WAVE < n) {
            if (i < n) {
                const size_t s = (size_t)r * n + i;
                rawv[c] = raw[s * raw_stride];
                selv[c] = selector ? (float)selector[s] : 1.f;
                delta[c] = eb[i + 1] - eb[i];
                dd[c] = delta[c] * ((is_density ? rawv[c] : expf(rawv[c])) * selv[c]);
            }
            const float inc = wave_incl_scan(dd[c]);
            const float excl = carry + shift_up1(inc);
            carry += __shfl(inc, WAVE - 1, WAVE);
            if (i < n) {
                T[c] = expf(-excl);
                const float alpha = 1.f - expf(-dd[c]);
                const float wraw = alpha * T[c];
                // nan_to_num backward: no gradient where the raw weight was nan/inf
                const float g = (wraw == wraw && fabsf(wraw) != INFINITY) ? gw[(size_t)r * n + i] : 0.f;
                gk[c] = g;
                gwt[c] = g * wraw;
            }
        }
    }
    // suffix sums of g_i*w_i over i > k, chunks visited from the far end
    float tail = 0.f;
#pragma unroll
    for (int c = MAXC - 1; c >= 0; --c) {
        if (c * WAVE < n) {
            const int i = c * WAVE + lane;
            const float incl = wave_incl_scan_rev(gwt[c]);
            const float after = tail + shift_down1(incl);
            tail += __shfl(incl, 0, WAVE);
            if (i < n) {
                const float gdd = gk[c] * T[c] * expf(-dd[c]) - after;
                const float gsigma = gdd * delta[c];
                const float xr = fminf(fmaxf(rawv[c], -15.f), 15.f);
                grad_raw[((size_t)r * n + i) * raw_stride] = gsigma * selv[c] * (is_density ? 1.f : expf(xr));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// density = trunc_exp(raw) * selector (Field.get_density tail: nerfacto_field.py:260-265, density_fields.py:120-124)
__global__ __launch_bounds__(256) void k_trunc_exp_fwd(const float* __restrict__ raw, int raw_stride,
                                                       const uint8_t* __restrict__ selector, long long N,
                                                       float* __restrict__ density) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    float d = expf(raw[t * raw_stride]);
    if (selector) d *= (float)selector[t];
    density[t] = d;
}

__global__ __launch_bounds__(256) void k_trunc_exp_bwd(const float* __restrict__ raw, int raw_stride,
                                                       const uint8_t* __restrict__ selector,
                                                       const float* __restrict__ gd, long long N,
                                                       float* __restrict__ grad_raw, int grad_stride) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    const float x = fminf(fmaxf(raw[t * raw_stride], -15.f), 15.f);
    float g = gd[t] * expf(x);
    if (selector) g *= (float)selector[t];
    grad_raw[t * grad_stride] = g;
}

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_composite_fwd(const float* __restrict__ rgb, const float* __restrict__ weights,
                                                       const float* __restrict__ ebins, int R, int S, int training,
                                                       float* __restrict__ out_rgb, float* __restrict__ out_acc,
                                                       float* __restrict__ out_depth) {
    const int r = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (r >= R) return;
    const int lane = lane_id();
    const float* w = weights + (size_t)r * S;
    const float* c = rgb + (size_t)r * S * 3;
    float sr = 0.f, sg = 0.f, sb = 0.f, sw = 0.f;
    float carry = 0.f;
    int med = 0x7fffffff;  // first index with cumsum >= 0.5 (searchsorted side='left')
    for (int base = 0; base < S; base += WAVE) {
        const int i = base + lane;
        float wi = 0.f;
        if (i < S) {
            wi = w[i];
            sw += wi;
            if (out_rgb) {
                float cr = c[i * 3 + 0], cg = c[i * 3 + 1], cb = c[i * 3 + 2];
                if (!training) { cr = nan_to_num(cr); cg = nan_to_num(cg); cb = nan_to_num(cb); }
                sr += wi * cr; sg += wi * cg; sb += wi * cb;
            }
        }
        const float inc = wave_incl_scan(wi) + carry;
        if (i < S && inc >= 0.5f) med = min(med, i);
        carry = __shfl(inc, WAVE - 1, WAVE);
    }
    sr = wave_sum(sr); sg = wave_sum(sg); sb = wave_sum(sb); sw = wave_sum(sw);
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) med = min(med, __shfl_xor(med, d, WAVE));
    if (lane == 0) {
        if (out_rgb) {
            float lr = c[(S - 1) * 3 + 0], lg = c[(S - 1) * 3 + 1], lb = c[(S - 1) * 3 + 2];
            if (!training) { lr = nan_to_num(lr); lg = nan_to_num(lg); lb = nan_to_num(lb); }
            float o0 = sr + lr * (1.f - sw), o1 = sg + lg * (1.f - sw), o2 = sb + lb * (1.f - sw);
            if (!training) {
                o0 = fminf(fmaxf(o0, 0.f), 1.f); o1 = fminf(fmaxf(o1, 0.f), 1.f); o2 = fminf(fmaxf(o2, 0.f), 1.f);
            }
            out_rgb[r * 3 + 0] = o0; out_rgb[r * 3 + 1] = o1; out_rgb[r * 3 + 2] = o2;
        }
        if (out_acc) out_acc[r] = sw;
        if (out_depth) {
            const int idx = min(med, S - 1);  // clamp(searchsorted, 0, S-1)
            const float* eb = ebins + (size_t)r * (S + 1);
            out_depth[r] = (eb[idx] + eb[idx + 1]) / 2.f;
        }
    }
}

// training-mode backward: comp = sum_s w_s c_s + c_last (1 - sum_s w_s)
__global__ __launch_bounds__(256) void k_composite_bwd(const float* __restrict__ rgb, const float* __restrict__ weights,
                                                       const float* __restrict__ gout, int R, int S,
                                                       float* __restrict__ grad_rgb, float* __restrict__ grad_w) {
    const int r = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (r >= R) return;
    const int lane = lane_id();
    const float* w = weights + (size_t)r * S;
    const float* c = rgb + (size_t)r * S * 3;
    const float g0 = gout[r * 3 + 0], g1 = gout[r * 3 + 1], g2 = gout[r * 3 + 2];
    const float l0 = c[(S - 1) * 3 + 0], l1 = c[(S - 1) * 3 + 1], l2 = c[(S - 1) * 3 + 2];
    float part = 0.f;
    for (int i = lane; i < S; i += WAVE) part += w[i];
    const float sw = wave_sum(part);
    for (int i = lane; i < S; i += WAVE) {
        const float wi = w[i];
        float a0 = g0 * wi, a1 = g1 * wi, a2 = g2 * wi;
        if (i == S - 1) { a0 += g0 * (1.f - sw); a1 += g1 * (1.f - sw); a2 += g2 * (1.f - sw); }
        float* gr = grad_rgb + ((size_t)r * S + i) * 3;
        gr[0] = a0; gr[1] = a1; gr[2] = a2;
        grad_w[(size_t)r * S + i] = g0 * (c[i * 3 + 0] - l0) + g1 * (c[i * 3 + 1] - l1) + g2 * (c[i * 3 + 2] - l2);
    }
}

// ------------------------------------------------------------------------------------------
// out[r, c] = sum_k w[r,k] * e[r,k,c] ; thread per (r, c-quad)
__global__ __launch_bounds__(256) void k_feature_mean_fwd(const float* __restrict__ e, const float* __restrict__ w, int R,
                                                          int K, int C, float* __restrict__ out) {
    const int cq = C >> 2;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)R * cq) return;
    const int r = (int)(t / cq), q = (int)(t - (long long)r * cq);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < K; ++k) {
        const float wk = w[(size_t)r * K + k];
        const float4 v = *reinterpret_cast<const float4*>(e + ((size_t)r * K + k) * C + q * 4);
        acc.x += wk * v.x; acc.y += wk * v.y; acc.z += wk * v.z; acc.w += wk * v.w;
    }
    *reinterpret_cast<float4*>(out + (size_t)r * C + q * 4) = acc;
}

__global__ __launch_bounds__(256) void k_feature_mean_bwd(const float* __restrict__ g, const float* __restrict__ w, int R,
                                                          int K, int C, float* __restrict__ ge) {
    const int cq = C >> 2;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)R * K * cq) return;
    const long long rk = t / cq;
    const int q = (int)(t - rk * cq);
    const int r = (int)(rk / K);
    const float wk = w[rk];
    const float4 v = *reinterpret_cast<const float4*>(g + (size_t)r * C + q * 4);
    *reinterpret_cast<float4*>(ge + (size_t)rk * C + q * 4) = make_float4(wk * v.x, wk * v.y, wk * v.z, wk * v.w);
}

}  // namespace snf

using namespace snf;

extern "C" int snf_head_input(const float* dirs, const float* geo, int R, int S, int n_geo, int ld_geo, float* out,
                              int ld_out, snf_stream_t stream) {
    SNF_REQUIRE(dirs && geo && out, "snf_head_input: null pointer");
    SNF_REQUIRE(R > 0 && S > 0 && n_geo >= 0 && ld_out >= 16 + n_geo && ld_geo >= n_geo, "snf_head_input: bad shape");
    if (ld_out == 32 && n_geo <= 16 && ((uintptr_t)out & 15) == 0)
        hipLaunchKernelGGL(k_head_input32, dim3(ceil_div((long long)R * S * 8, 256)), dim3(256), 0, (hipStream_t)stream, dirs,
                           geo, R, S, n_geo, ld_geo, out);
    else
        hipLaunchKernelGGL(k_head_input, dim3(ceil_div((long long)R * S, 256)), dim3(256), 0, (hipStream_t)stream, dirs, geo,
                           R, S, n_geo, ld_geo, out, ld_out);
    SNF_LAUNCH_CHECK("snf_head_input");
    return SNF_OK;
}

extern "C" int snf_weights_fwd(const float* raw, int raw_stride, int is_density, const uint8_t* selector,
                               const float* ebins, int R, int n, float* weights, float* density, snf_stream_t stream) {
    SNF_REQUIRE(raw && ebins && weights, "snf_weights_fwd: null pointer");
    SNF_REQUIRE(R > 0 && n > 0 && n <= MAXC * WAVE && raw_stride >= 1, "snf_weights_fwd: bad shape R=%d n=%d", R, n);
    hipLaunchKernelGGL(k_weights_fwd, dim3(ceil_div(R, RAYS_PER_BLOCK)), dim3(256), 0, (hipStream_t)stream, raw,
                       raw_stride, is_density, selector, ebins, R, n, weights, density);
    SNF_LAUNCH_CHECK("snf_weights_fwd");
    return SNF_OK;
}

extern "C" int snf_weights_bwd(const float* raw, int raw_stride, int is_density, const uint8_t* selector,
                               const float* ebins, const float* grad_weights, int R, int n, float* grad_raw,
                               snf_stream_t stream) {
    SNF_REQUIRE(raw && ebins && grad_weights && grad_raw, "snf_weights_bwd: null pointer");
    SNF_REQUIRE(R > 0 && n > 0 && n <= MAXC * WAVE && raw_stride >= 1, "snf_weights_bwd: bad shape R=%d n=%d", R, n);
    hipLaunchKernelGGL(k_weights_bwd, dim3(ceil_div(R, RAYS_PER_BLOCK)), dim3(256), 0, (hipStream_t)stream, raw,
                       raw_stride, is_density, selector, ebins, grad_weights, R, n, grad_raw);
    SNF_LAUNCH_CHECK("snf_weights_bwd");
    return SNF_OK;
}

extern "C" int snf_trunc_exp_fwd(const float* raw, int raw_stride, const uint8_t* selector, int64_t N, float* density,
                                 snf_stream_t stream) {
    SNF_REQUIRE(raw && density && N > 0 && raw_stride >= 1, "snf_trunc_exp_fwd: bad argument");
    hipLaunchKernelGGL(k_trunc_exp_fwd, dim3(ceil_div(N, 256)), dim3(256), 0, (hipStream_t)stream, raw, raw_stride,
                       selector, (long long)N, density);
    SNF_LAUNCH_CHECK("snf_trunc_exp_fwd");
    return SNF_OK;
}

extern "C" int snf_trunc_exp_bwd(const float* raw, int raw_stride, const uint8_t* selector, const float* grad_density,
                                 int64_t N, float* grad_raw, int grad_stride, snf_stream_t stream) {
    SNF_REQUIRE(raw && grad_density && grad_raw && N > 0 && raw_stride >= 1 && grad_stride >= 1,
                "snf_trunc_exp_bwd: bad argument");
    hipLaunchKernelGGL(k_trunc_exp_bwd, dim3(ceil_div(N, 256)), dim3(256), 0, (hipStream_t)stream, raw, raw_stride,
                       selector, grad_density, (long long)N, grad_raw, grad_stride);
    SNF_LAUNCH_CHECK("snf_trunc_exp_bwd");
    return SNF_OK;
}

extern "C" int snf_composite_fwd(const float* rgb, const float* weights, const float* ebins, int R, int S, int training,
                                 float* out_rgb, float* out_acc, float* out_depth, snf_stream_t stream) {
    SNF_REQUIRE(weights && (rgb || !out_rgb) && (ebins || !out_depth), "snf_composite_fwd: null pointer");
    SNF_REQUIRE(R > 0 && S > 0, "snf_composite_fwd: bad shape");
    // rgb == NULL (with out_rgb == NULL) renders depth / accumulation only, e.g. prop_depth_i
    hipLaunchKernelGGL(k_composite_fwd, dim3(ceil_div(R, RAYS_PER_BLOCK)), dim3(256), 0, (hipStream_t)stream, rgb,
                       weights, ebins, R, S, training, out_rgb, out_acc, out_depth);
    SNF_LAUNCH_CHECK("snf_composite_fwd");
    return SNF_OK;
}

extern "C" int snf_composite_bwd(const float* rgb, const float* weights, const float* grad_out_rgb, int R, int S,
                                 float* grad_rgb, float* grad_weights, snf_stream_t stream) {
    SNF_REQUIRE(rgb && weights && grad_out_rgb && grad_rgb && grad_weights, "snf_composite_bwd: null pointer");
    SNF_REQUIRE(R > 0 && S > 0, "snf_composite_bwd: bad shape");
    hipLaunchKernelGGL(k_composite_bwd, dim3(ceil_div(R, RAYS_PER_BLOCK)), dim3(256), 0, (hipStream_t)stream, rgb,
                       weights, grad_out_rgb, R, S, grad_rgb, grad_weights);
    SNF_LAUNCH_CHECK("snf_composite_bwd");
    return SNF_OK;
}

extern "C" int snf_feature_mean_fwd(const float* embeds, const float* w, int R, int K, int C, float* out,
                                    snf_stream_t stream) {
    SNF_REQUIRE(embeds && w && out, "snf_feature_mean_fwd: null pointer");
    SNF_REQUIRE(R > 0 && K > 0 && C > 0 && C % 4 == 0, "snf_feature_mean_fwd: bad shape (C must be a multiple of 4)");
    hipLaunchKernelGGL(k_feature_mean_fwd, dim3(ceil_div((long long)R * (C / 4), 256)), dim3(256), 0,
                       (hipStream_t)stream, embeds, w, R, K, C, out);
    SNF_LAUNCH_CHECK("snf_feature_mean_fwd");
    return SNF_OK;
}

extern "C" int snf_feature_mean_bwd(const float* grad_out, const float* w, int R, int K, int C, float* grad_embeds,
                                    snf_stream_t stream) {
    SNF_REQUIRE(grad_out && w && grad_embeds, "snf_feature_mean_bwd: null pointer");
    SNF_REQUIRE(R > 0 && K > 0 && C > 0 && C % 4 == 0, "snf_feature_mean_bwd: bad shape (C must be a multiple of 4)");
    hipLaunchKernelGGL(k_feature_mean_bwd, dim3(ceil_div((long long)R * K * (C / 4), 256)), dim3(256), 0,
                       (hipStream_t)stream, grad_out, w, R, K, C, grad_embeds);
    SNF_LAUNCH_CHECK("snf_feature_mean_bwd");
    return SNF_OK;
}

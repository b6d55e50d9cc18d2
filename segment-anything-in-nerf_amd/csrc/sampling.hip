// sampling.hip -- ray-sample generation kernels (gfx950).
//   snf_sample_spacing  : a3  UniformLinDispPiecewiseSampler   (ray_samplers.py:79-126,223-246)
//   snf_positions       : a1+a4 positions + contraction + (x+2)/4 + selector
//   snf_pdf_resample    : a10 PDFSampler                       (ray_samplers.py:298-367)
//   snf_topk_sharpen    : a16 top-K + sharpen + renormalise    (samnerf/sam_model.py:244-248)
// One 64-lane wavefront owns one ray; a 256-thread workgroup carries 4 rays.
#include "common.hpp"

// Bin edges go through ill-conditioned maps (1/(2-2y) near the far plane); evaluate them with separately rounded
// mul/add exactly like the torch reference so that sample positions agree to the last bit.
#pragma clang fp contract(off)

namespace snf {

constexpr int RAYS_PER_BLOCK = 4;
constexpr int MAX_BINS = 512;  // P+1, S+1 <= 512

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sample_spacing(const float* __restrict__ nears,
                                                        const float* __restrict__ fars,
                                                        const float* __restrict__ t_rand, int R, int P,
                                                        float* __restrict__ sbins, float* __restrict__ ebins) {
    const int r = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (r >= R) return;
    const int lane = lane_id();
    const float s_near = spacing_fn(nears[r]);
    const float s_far = spacing_fn(fars[r]);
    const bool jitter = t_rand != nullptr;
    const float t = jitter ? t_rand[r] : 0.f;
    for (int i = lane; i <= P; i += WAVE) {
        float b = linspace_at(0.f, 1.f, P + 1, i);
        if (jitter) {
            // bin_lower/bin_upper of ray_samplers.py:107-109
            const float bm = linspace_at(0.f, 1.f, P + 1, i > 0 ? i - 1 : 0);
            const float bp = linspace_at(0.f, 1.f, P + 1, i < P ? i + 1 : P);
            const float lower = (i == 0) ? b : (b + bm) / 2.f;
            const float upper = (i == P) ? b : (bp + b) / 2.f;
            b = lower + (upper - lower) * t;
        }
        sbins[(size_t)r * (P + 1) + i] = b;
        ebins[(size_t)r * (P + 1) + i] = spacing_fn_inv(b * s_far + (1.f - b) * s_near);
    }
}

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_positions(const float* __restrict__ origins,
                                                   const float* __restrict__ dirs,
                                                   const float* __restrict__ ebins,
                                                   const int32_t* __restrict__ ids, int R, int n, int K,
                                                   int contraction, int use_selector, float* __restrict__ u,
                                                   uint8_t* __restrict__ selector, const int32_t* __restrict__ src_rows,
                                                   const int32_t* __restrict__ dst_rows) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)R * K) return;
    int r = (int)(t / K);
    const int k = (int)(t - (long long)r * K);
    if (src_rows) {
        // row-mapped form (snf_positions_rows): list entry r reads ray src_rows[r] of the chunk (origins / dirs / ebins) and
        // owns row dst_rows[r] of ids / u -- the eval render keeps the selected samples of an index subset of the camera's rays
        t = (long long)dst_rows[r] * K + k;
        r = src_rows[r];
    }
    const int i = ids ? ids[t] : k;
    const float st = ebins[(size_t)r * (n + 1) + i];
    const float en = ebins[(size_t)r * (n + 1) + i + 1];
    const float mid = (st + en) / 2.f;
    // explicit round-to-nearest mul/add (no FMA contraction) so positions match the torch path bit for bit
    float x = __fadd_rn(origins[r * 3 + 0], __fmul_rn(dirs[r * 3 + 0], mid));
    float y = __fadd_rn(origins[r * 3 + 1], __fmul_rn(dirs[r * 3 + 1], mid));
    float z = __fadd_rn(origins[r * 3 + 2], __fmul_rn(dirs[r * 3 + 2], mid));
    if (contraction != SNF_CONTRACT_NONE) {
        float mag;
        if (contraction == SNF_CONTRACT_LINF) {
            mag = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
        } else {
            mag = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z)));
        }
        if (!(mag < 1.f)) {
            const float s = 2.f - 1.f / mag;
            x = __fmul_rn(s, x / mag);
            y = __fmul_rn(s, y / mag);
            z = __fmul_rn(s, z / mag);
        }
    }
    x = (x + 2.f) / 4.f;
    y = (y + 2.f) / 4.f;
    z = (z + 2.f) / 4.f;
    if (use_selector) {
        const bool s = (x > 0.f) && (x < 1.f) && (y > 0.f) && (y < 1.f) && (z > 0.f) && (z < 1.f);
        const float m = s ? 1.f : 0.f;
        x *= m;
        y *= m;
        z *= m;
        selector[t] = s ? 1 : 0;
    }
    u[t * 3 + 0] = x;
    u[t * 3 + 1] = y;
    u[t * 3 + 2] = z;
}

// ------------------------------------------------------------------------------------------
// One wave per ray.  cdf and the existing bins live in LDS; every lane then resolves its own u's.
__global__ __launch_bounds__(256) void k_pdf_resample(const float* __restrict__ weights,
                                                      const float* __restrict__ sbins_in,
                                                      const float* __restrict__ u_rand,
                                                      const float* __restrict__ nears,
                                                      const float* __restrict__ fars, int R, int P, int S,
                                                      float anneal, float padding_h, float* __restrict__ sbins,
                                                      float* __restrict__ ebins) {
    __shared__ float s_cdf[RAYS_PER_BLOCK][MAX_BINS];
    __shared__ float s_bins[RAYS_PER_BLOCK][MAX_BINS];
    const int wv = threadIdx.x >> 6;
    const int r_raw = blockIdx.x * RAYS_PER_BLOCK + wv;
    const bool active = r_raw < R;
    const int r = active ? r_raw : R - 1;  // idle waves shadow the last ray (no early exit before barriers)
    const int lane = lane_id();
    float* cdf = s_cdf[wv];
    float* eb = s_bins[wv];
    const float* w_in = weights + (size_t)r * P;
    const float eps = 1e-5f;
    // pass 1: padded weights and their sum
    float part = 0.f;
    for (int i = lane; i < P; i += WAVE) {
        float w = w_in[i];
        if (anneal != 1.f) w = powf(w, anneal);
        w += padding_h;
        cdf[i + 1] = w;  // stash
        part += w;
    }
    float wsum = wave_sum(part);
    const float pad = fmaxf(eps - wsum, 0.f);
    const float add = pad / (float)P;
    wsum += pad;
    // pass 2: pdf -> inclusive cumsum, chunk by chunk with a running carry
    float carry = 0.f;
    for (int base = 0; base < P; base += WAVE) {
        const int i = base + lane;
        float pdf = (i < P) ? (cdf[i + 1] + add) / wsum : 0.f;
        float inc = wave_incl_scan(pdf) + carry;
        if (i < P) cdf[i + 1] = fminf(1.f, inc);
        carry = __shfl(inc, WAVE - 1, WAVE);
    }
    if (lane == 0) cdf[0] = 0.f;
    for (int i = lane; i <= P; i += WAVE) eb[i] = sbins_in[(size_t)r * (P + 1) + i];
    __syncthreads();
    const int nb = S + 1;
    const float s_near = spacing_fn(nears[r]);
    const float s_far = spacing_fn(fars[r]);
    // python-double scalars of ray_samplers.py:316-324, rounded to fp32 once like torch does
    const float u_end = (float)(1.0 - (1.0 / (double)nb));
    const float off = u_rand ? u_rand[r] / (float)nb : (float)(1.0 / (2.0 * (double)nb));
    for (int k = lane; k < nb; k += WAVE) {
        const float uu = linspace_at(0.f, u_end, nb, k) + off;
        // searchsorted(cdf[0..P], uu, right): first index with cdf[idx] > uu
        int lo = 0, hi = P + 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= uu) lo = mid + 1; else hi = mid;
        }
        const int below = min(max(lo - 1, 0), P);
        const int above = min(max(lo, 0), P);
        const float c0 = cdf[below], c1 = cdf[above];
        const float b0 = eb[below], b1 = eb[above];
        float tt = (uu - c0) / (c1 - c0);
        // torch.nan_to_num(x, 0): nan -> 0, +-inf -> +-FLT_MAX, then clip(0,1)
        tt = nan_to_num(tt);
        tt = fminf(fmaxf(tt, 0.f), 1.f);
        const float b = b0 + tt * (b1 - b0);
        if (active) {
            sbins[(size_t)r * nb + k] = b;
            ebins[(size_t)r * nb + k] = spacing_fn_inv(b * s_far + (1.f - b) * s_near);
        }
    }
}

// ------------------------------------------------------------------------------------------
// One wave per ray; each lane keeps up to 4 candidates (S <= 256); K rounds of wave arg-max.
__global__ __launch_bounds__(256) void k_topk_sharpen(const float* __restrict__ weights, int R, int S, int K,
                                                      float temperature, int32_t* __restrict__ ids,
                                                      float* __restrict__ sam_w, const int32_t* __restrict__ src_rows,
                                                      const int32_t* __restrict__ dst_rows) {
    int r = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (r >= R) return;
    int ro = r;  // output row
    if (src_rows) {  // row-mapped form (snf_topk_sharpen_rows): weights row src_rows[r] -> ids / sam_w row dst_rows[r]
        ro = dst_rows[r];
        r = src_rows[r];
    }
    const int lane = lane_id();
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = j * WAVE + lane;
        float x = (i < S) ? weights[(size_t)r * S + i] : -INFINITY;
        if (x != x) x = INFINITY;  // torch.topk treats NaN as the largest value
        v[j] = x;
    }
    float my_w = 0.f;  // lane k (< K) keeps the k-th selected weight
    int my_id = 0;
    for (int k = 0; k < K; ++k) {
        // local best (ties -> lowest index)
        float bv = v[0];
        int bi = lane;
#pragma unroll
        for (int j = 1; j < 4; ++j) {
            if (v[j] > bv) { bv = v[j]; bi = j * WAVE + lane; }
        }
#pragma unroll
        for (int d = WAVE / 2; d > 0; d >>= 1) {
            const float ov = __shfl_xor(bv, d, WAVE);
            const int oi = __shfl_xor(bi, d, WAVE);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if ((bi & (WAVE - 1)) == lane) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if ((bi >> 6) == j) v[j] = -INFINITY;
        }
        if (lane == k) { my_w = weights[(size_t)r * S + bi]; my_id = bi; }
    }
    float p = 0.f;
    if (lane < K) p = powf(my_w, temperature);
    const float denom = wave_sum(lane < K ? p : 0.f);
    if (lane < K) {
        ids[(size_t)ro * K + lane] = my_id;
        sam_w[(size_t)ro * K + lane] = p / denom;
    }
}

}  // namespace snf

using namespace snf;

extern "C" int snf_sample_spacing(const float* nears, const float* fars, const float* t_rand, int R, int P,
                                  float* sbins, float* ebins, snf_stream_t stream) {
    SNF_REQUIRE(nears && fars && sbins && ebins, "snf_sample_spacing: null pointer");
    SNF_REQUIRE(R > 0 && P >= 1 && P + 1 <= MAX_BINS, "snf_sample_spacing: bad shape R=%d P=%d", R, P);
    hipLaunchKernelGGL(k_sample_spacing, dim3(ceil_div(R, RAYS_PER_BLOCK)), dim3(256), 0, (hipStream_t)stream,
                       nears, fars, t_rand, R, P, sbins, ebins);
    SNF_LAUNCH_CHECK("snf_sample_spacing");
    return SNF_OK;
}

extern "C" int snf_positions(const float* origins, const float* dirs, const float* ebins, const int32_t* ids,
                             int R, int n, int K, int contraction, int use_selector, float* u,
                             uint8_t* selector, snf_stream_t stream) {
    SNF_REQUIRE(origins && dirs && ebins && u, "snf_positions: null pointer");
    SNF_REQUIRE(R > 0 && n > 0 && K > 0, "snf_positions: bad shape R=%d n=%d K=%d", R, n, K);
    SNF_REQUIRE(ids || K == n, "snf_positions: ids == NULL requires K == n");
    SNF_REQUIRE(contraction >= 0 && contraction <= 2, "snf_positions: bad contraction %d", contraction);
    SNF_REQUIRE(!use_selector || selector, "snf_positions: selector buffer missing");
    const long long total = (long long)R * K;
    hipLaunchKernelGGL(k_positions, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, origins, dirs,
                       ebins, ids, R, n, K, contraction, use_selector, u, selector, (const int32_t*)nullptr,
                       (const int32_t*)nullptr);
    SNF_LAUNCH_CHECK("snf_positions");
    return SNF_OK;
}

extern "C" int snf_positions_rows(const float* origins, const float* dirs, const float* ebins, const int32_t* ids,
                                  const int32_t* src_rows, const int32_t* dst_rows, int M, int n, int K, int contraction,
                                  float* u, snf_stream_t stream) {
    SNF_REQUIRE(origins && dirs && ebins && ids && u && src_rows && dst_rows, "snf_positions_rows: null pointer");
    SNF_REQUIRE(M > 0 && n > 0 && K > 0, "snf_positions_rows: bad shape M=%d n=%d K=%d", M, n, K);
    SNF_REQUIRE(contraction >= 0 && contraction <= 2, "snf_positions_rows: bad contraction %d", contraction);
    const long long total = (long long)M * K;
    hipLaunchKernelGGL(k_positions, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, origins, dirs,
                       ebins, ids, M, n, K, contraction, 0, u, (uint8_t*)nullptr, src_rows, dst_rows);
    SNF_LAUNCH_CHECK("snf_positions_rows");
    return SNF_OK;
}

extern "C" int snf_pdf_resample(const float* weights, const float* sbins_in, const float* u_rand,
                                const float* nears, const float* fars, int R, int P, int S, float anneal,
                                float histogram_padding, float* sbins, float* ebins, snf_stream_t stream) {
    SNF_REQUIRE(weights && sbins_in && nears && fars && sbins && ebins, "snf_pdf_resample: null pointer");
    SNF_REQUIRE(R > 0 && P >= 1 && S >= 1 && P + 1 <= MAX_BINS && S + 1 <= MAX_BINS,
                "snf_pdf_resample: bad shape R=%d P=%d S=%d", R, P, S);
    hipLaunchKernelGGL(k_pdf_resample, dim3(ceil_div(R, RAYS_PER_BLOCK)), dim3(256), 0, (hipStream_t)stream,
                       weights, sbins_in, u_rand, nears, fars, R, P, S, anneal, histogram_padding, sbins, ebins);
    SNF_LAUNCH_CHECK("snf_pdf_resample");
    return SNF_OK;
}

extern "C" int snf_topk_sharpen(const float* weights, int R, int S, int K, float temperature, int32_t* ids,
                                float* sam_weights, snf_stream_t stream) {
    SNF_REQUIRE(weights && ids && sam_weights, "snf_topk_sharpen: null pointer");
    SNF_REQUIRE(R > 0 && S >= 1 && S <= 256 && K >= 1 && K <= 64 && K <= S,
                "snf_topk_sharpen: bad shape R=%d S=%d K=%d (need S<=256, K<=min(64,S))", R, S, K);
    hipLaunchKernelGGL(k_topk_sharpen, dim3(ceil_div(R, RAYS_PER_BLOCK)), dim3(256), 0, (hipStream_t)stream,
                       weights, R, S, K, temperature, ids, sam_weights, (const int32_t*)nullptr, (const int32_t*)nullptr);
    SNF_LAUNCH_CHECK("snf_topk_sharpen");
    return SNF_OK;
}

extern "C" int snf_topk_sharpen_rows(const float* weights, const int32_t* src_rows, const int32_t* dst_rows, int M, int S,
                                     int K, float temperature, int32_t* ids, float* sam_weights, snf_stream_t stream) {
    SNF_REQUIRE(weights && ids && sam_weights && src_rows && dst_rows, "snf_topk_sharpen_rows: null pointer");
    SNF_REQUIRE(M > 0 && S >= 1 && S <= 256 && K >= 1 && K <= 64 && K <= S,
                "snf_topk_sharpen_rows: bad shape M=%d S=%d K=%d (need S<=256, K<=min(64,S))", M, S, K);
    hipLaunchKernelGGL(k_topk_sharpen, dim3(ceil_div(M, RAYS_PER_BLOCK)), dim3(256), 0, (hipStream_t)stream,
                       weights, M, S, K, temperature, ids, sam_weights, src_rows, dst_rows);
    SNF_LAUNCH_CHECK("snf_topk_sharpen_rows");
    return SNF_OK;
}

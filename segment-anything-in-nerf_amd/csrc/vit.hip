// vit.hip -- the non-GEMM pieces of the SAM image encoder forward (reference
// samnerf/segment_anything/modeling/image_encoder.py, common.py) for gfx950.  The dense layers (patch embedding, qkv, proj,
// MLP, neck) are snf_linear_fwd GEMMs; here:
//   snf_patchify          : image [B,3,S,S] -> rows [B*G*G, 3*P*P] in the column order of Conv2d.weight.view(E, 3*P*P)
//   snf_layernorm         : nn.LayerNorm over the last axis (also LayerNorm2d on channel-last rows), optional residual add
//   snf_window_partition  : window_partition with zero padding (image_encoder.py:239-261)
//   snf_window_merge_add  : window_unpartition + the block's `shortcut + x` (image_encoder.py:264-287,178-180)
//   snf_relpos            : the two decomposed relative-position terms rel_h, rel_w (image_encoder.py:323-361)
//   snf_attention         : softmax((q*scale) k^T + rel_h + rel_w) v on the fp32 matrix cores, flash style
// Attention layout: tokens stay in the qkv GEMM's output [B*T, 3*C] (q | k | v, each heads x head_dim) -- no permutes; the
// output goes straight into [B*T, C] at column head*hd, which is what the proj GEMM reads.
//
// snf_attention: one wave = 32 queries of one (batch-window, head); four waves share staged K / V tiles of 32 keys.
// The score tile is formed TRANSPOSED, S^T[key][query] = K Q^T (A = K rows from LDS, B = the lane's own query row held in
// registers), so that in the MFMA accumulator layout a lane owns ONE query column: the online-softmax statistics are
// per-lane scalars (max / sum over the lane's 16 registers + one cross-half shuffle), rescaling the output accumulators is a
// per-lane multiply, and P^T is already the B operand of O^T = V^T P^T -- the same transposed chaining as mlp_chain.hip.
#include "common.hpp"
#include <math.h>
#include <stdlib.h>

namespace snf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int vrow(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }  // C/D row of reg r

__global__ __launch_bounds__(256) void k_patchify(const float* __restrict__ img, int B, int Cin, int S, int P,
                                                  float* __restrict__ rows) {
    const int G = S / P, K = Cin * P * P;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)B * G * G * K) return;
    const int col = (int)(t % K);
    const long long tok = t / K;
    const int c = col / (P * P), ky = (col / P) % P, kx = col % P;
    const int b = (int)(tok / (G * G)), gy = (int)((tok / G) % G), gx = (int)(tok % G);
    rows[t] = img[(((size_t)b * Cin + c) * S + gy * P + ky) * S + gx * P + kx];
}

// y = LN(x') * w + b with x' = x + res (res optional; biased variance, eps inside the sqrt); x' is also written to `sum_out`
// when given -- a block's second residual `x + mlp(..)` folds into the next block's norm1 this way.  One wave per row.
__global__ __launch_bounds__(256) void k_layernorm(const float* __restrict__ x, const float* __restrict__ res, int N, int C,
                                                   const float* __restrict__ w, const float* __restrict__ b, float eps,
                                                   float* __restrict__ sum_out, float* __restrict__ y) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    const float* xr = x + (size_t)row * C;
    const float* rr = res ? res + (size_t)row * C : nullptr;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c] + (rr ? rr[c] : 0.f);
    const float mean = wave_sum(s) / (float)C;
    float v = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float d = xr[c] + (rr ? rr[c] : 0.f) - mean;
        v += d * d;
    }
    const float inv = 1.f / sqrtf(wave_sum(v) / (float)C + eps);
    for (int c = lane; c < C; c += 64) {
        const float xv = xr[c] + (rr ? rr[c] : 0.f);
        if (sum_out) sum_out[(size_t)row * C + c] = xv;
        y[(size_t)row * C + c] = (xv - mean) * inv * w[c] + b[c];
    }
}

// the same with the row held in registers (C = 256 NV, 16-byte aligned rows: ViT-H's 1280 is NV = 5): one read of x (and res),
// float4 accesses, the reductions from registers.  The lane owns four consecutive columns here instead of every 64th, so the two
// wave sums add in another order than k_layernorm's (fp32 rounding only).
template <int NV>
__global__ __launch_bounds__(256) void k_layernorm_reg(const float* __restrict__ x, const float* __restrict__ res, int N, int C,
                                                       const float* __restrict__ w, const float* __restrict__ b, float eps,
                                                       float* __restrict__ sum_out, float* __restrict__ y) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * C);
    const float4* rr = res ? reinterpret_cast<const float4*>(res + (size_t)row * C) : nullptr;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        v[q] = xr[lane + 64 * q];
        if (rr) {
            const float4 r4 = rr[lane + 64 * q];
            v[q].x += r4.x; v[q].y += r4.y; v[q].z += r4.z; v[q].w += r4.w;
        }
        s += (v[q].x + v[q].y) + (v[q].z + v[q].w);
    }
    const float mean = wave_sum(s) / (float)C;
    float var = 0.f;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const float d0 = v[q].x - mean, d1 = v[q].y - mean, d2 = v[q].z - mean, d3 = v[q].w - mean;
        var += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    const float inv = 1.f / sqrtf(wave_sum(var) / (float)C + eps);
    const float4* w4 = reinterpret_cast<const float4*>(w);
    const float4* b4 = reinterpret_cast<const float4*>(b);
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const int c4 = lane + 64 * q;
        if (sum_out) reinterpret_cast<float4*>(sum_out + (size_t)row * C)[c4] = v[q];
        const float4 ww = w4[c4], bb = b4[c4];
        reinterpret_cast<float4*>(y + (size_t)row * C)[c4] =
            make_float4((v[q].x - mean) * inv * ww.x + bb.x, (v[q].y - mean) * inv * ww.y + bb.y,
                        (v[q].z - mean) * inv * ww.z + bb.z, (v[q].w - mean) * inv * ww.w + bb.w);
    }
}

// k_layernorm_reg writing its result as the GEMM's activation operand: bf16 hi / lo k-blocked planes [C/8][M_out][8]
// (csrc/gemm_planes.hip), at the row the window partition gives the token (ws > 0: token (b, y, x) of the [B, H, W] grid goes to
// window (y / ws, x / ws), position (y % ws, x % ws); the padded rows of the planes are never written and stay zero, as
// window_partition pads AFTER the norm, image_encoder.py:170-176,239-261).  A lane owns four consecutive columns: one 8-byte store
// per plane and quad, the two lanes of a k-block adjacent.
typedef __bf16 ln_bf16x2 __attribute__((ext_vector_type(2)));
typedef float ln_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ln_split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    const ln_f32x2 v = {x0, x1};
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, ln_bf16x2));
    const ln_f32x2 r = {x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xFFFF0000u)};
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, ln_bf16x2));
}

template <int NV>
__global__ __launch_bounds__(256) void k_layernorm_planes(const float* __restrict__ x, const float* __restrict__ res, int N, int C,
                                                          const float* __restrict__ w, const float* __restrict__ b, float eps,
                                                          float* __restrict__ sum_out, uint16_t* __restrict__ yhi,
                                                          uint16_t* __restrict__ ylo, int M_out, int H, int W, int ws, int res_ws) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * C);
    // res_ws > 0: the residual is still in WINDOW rows (the projection's output before window_unpartition): token (b, y, x) reads
    // the row of its window position -- `x = shortcut + window_unpartition(proj)` (image_encoder.py:176-180) folded into norm2
    int rrow = row;
    if (res && res_ws > 0) {
        const int nWh = (H + res_ws - 1) / res_ws, nWw = (W + res_ws - 1) / res_ws;
        const int bb = row / (H * W), yy = (row / W) % H, xx = row % W;
        rrow = ((bb * nWh + yy / res_ws) * nWw + xx / res_ws) * (res_ws * res_ws) + (yy % res_ws) * res_ws + xx % res_ws;
    }
    const float4* rr = res ? reinterpret_cast<const float4*>(res + (size_t)rrow * C) : nullptr;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        v[q] = xr[lane + 64 * q];
        if (rr) {
            const float4 r4 = rr[lane + 64 * q];
            v[q].x += r4.x; v[q].y += r4.y; v[q].z += r4.z; v[q].w += r4.w;
        }
        s += (v[q].x + v[q].y) + (v[q].z + v[q].w);
    }
    const float mean = wave_sum(s) / (float)C;
    float var = 0.f;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const float d0 = v[q].x - mean, d1 = v[q].y - mean, d2 = v[q].z - mean, d3 = v[q].w - mean;
        var += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    const float inv = 1.f / sqrtf(wave_sum(var) / (float)C + eps);
    int drow = row;
    if (ws > 0) {
        const int nWh = (H + ws - 1) / ws, nWw = (W + ws - 1) / ws;
        const int bb = row / (H * W), yy = (row / W) % H, xx = row % W;
        drow = ((bb * nWh + yy / ws) * nWw + xx / ws) * (ws * ws) + (yy % ws) * ws + xx % ws;
    }
    const float4* w4 = reinterpret_cast<const float4*>(w);
    const float4* b4 = reinterpret_cast<const float4*>(b);
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const int c4 = lane + 64 * q;  // column quad: k-block c4 / 2, elements 4 (c4 & 1) .. + 3 of it
        if (sum_out) reinterpret_cast<float4*>(sum_out + (size_t)row * C)[c4] = v[q];
        const float4 ww = w4[c4], bb = b4[c4];
        uint32_t h0, h1, l0, l1;
        ln_split2((v[q].x - mean) * inv * ww.x + bb.x, (v[q].y - mean) * inv * ww.y + bb.y, h0, l0);
        ln_split2((v[q].z - mean) * inv * ww.z + bb.z, (v[q].w - mean) * inv * ww.w + bb.w, h1, l1);
        const size_t o = ((size_t)(c4 >> 1) * M_out + drow) * 8 + 4 * (c4 & 1);
        *reinterpret_cast<uint2*>(yhi + o) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(ylo + o) = make_uint2(l0, l1);
    }
}

// x [B,H,W,C] -> windows [B * nWh * nWw, ws, ws, C], zero padded to multiples of ws
__global__ __launch_bounds__(256) void k_window_partition(const float* __restrict__ x, int B, int H, int W, int C, int ws,
                                                          float* __restrict__ out) {
    const int nWh = (H + ws - 1) / ws, nWw = (W + ws - 1) / ws;
    const int tok = blockIdx.x;  // output token
    const int per = ws * ws;
    const int win = tok / per, q = tok % per;
    const int b = win / (nWh * nWw), wy = (win / nWw) % nWh, wx = win % nWw;
    const int y = wy * ws + q / ws, xx = wx * ws + q % ws;
    const bool in = y < H && xx < W;
    const float* src = x + (((size_t)b * H + y) * W + xx) * C;
    float* dst = out + (size_t)tok * C;
    for (int c = threadIdx.x; c < C; c += 256) dst[c] = in ? src[c] : 0.f;
}

// out[b,y,x,:] = shortcut[b,y,x,:] + windows[window(y,x), pos(y,x), :]   (ws == 0: windows is already [B,H,W,C])
__global__ __launch_bounds__(256) void k_window_merge_add(const float* __restrict__ win, const float* __restrict__ shortcut,
                                                          int B, int H, int W, int C, int ws, float* __restrict__ out) {
    const int tok = blockIdx.x;  // b*H*W + y*W + x
    size_t src_tok = tok;
    if (ws > 0) {
        const int nWh = (H + ws - 1) / ws, nWw = (W + ws - 1) / ws;
        const int b = tok / (H * W), y = (tok / W) % H, x = tok % W;
        src_tok = ((size_t)(b * nWh + y / ws) * nWw + x / ws) * (ws * ws) + (y % ws) * ws + x % ws;
    }
    const float* s = win + src_tok * C;
    const float* sc = shortcut + (size_t)tok * C;
    float* d = out + (size_t)tok * C;
    for (int c = threadIdx.x; c < C; c += 256) d[c] = sc[c] + s[c];
}

// rel[bh][i][0..n) = q_i . rel_pos_h[ih - kh + n - 1],  rel[bh][i][n..2n) = q_i . rel_pos_w[iw - kw + n - 1]   (q unscaled)
// qkv [Bw*T, 3*C].  One workgroup per (bh, query row ih): the n queries of that row, the n table rows rel_pos_h[ih - kh + n-1]
// and the whole rel_pos_w table are staged in LDS (pitch hd+1), then every thread forms its (iw, j) dot products from LDS.
// Workgroup size = blockDim.x (1024 threads at n = 64, whose 82 KB of LDS leave one workgroup per CU: with 256 threads that was
// one wave per SIMD).  A thread forms FOUR consecutive j of a query at a time: one LDS read of q[c] serves four products.
__global__ __launch_bounds__(1024) void k_relpos(const float* __restrict__ qkv, int Bw, int T, int heads, int hd, int n,
                                                const float* __restrict__ rph, const float* __restrict__ rpw,
                                                float* __restrict__ rel) {
    extern __shared__ float rp_lds[];
    const int P = hd + 1;
    float* qs = rp_lds;                 // [n][P]
    float* hs = qs + n * P;             // [n][P]      rel_pos_h rows for kh = 0..n-1
    float* ws = hs + n * P;             // [2n-1][P]   rel_pos_w
    const int ih = blockIdx.x, bh = blockIdx.y, b = bh / heads, h = bh % heads, C = heads * hd;
    const int nt = blockDim.x;
    for (int e = threadIdx.x; e < n * hd; e += nt) {
        const int r = e / hd, c = e - r * hd;
        qs[r * P + c] = qkv[((size_t)b * T + ih * n + r) * 3 * C + h * hd + c];
        hs[r * P + c] = rph[(size_t)(ih - r + n - 1) * hd + c];
    }
    for (int e = threadIdx.x; e < (2 * n - 1) * hd; e += nt) {
        const int r = e / hd, c = e - r * hd;
        ws[r * P + c] = rpw[e];
    }
    __syncthreads();
    const int jb = (2 * n + 3) >> 2;  // blocks of four j per query
    if (n * jb < nt) {  // small grids (14 x 14 windows: 98 blocks for 256 threads): one j per thread keeps the lanes busy
        for (int e = threadIdx.x; e < n * 2 * n; e += nt) {
            const int iw = e / (2 * n), j = e - iw * 2 * n;
            const float* q = qs + iw * P;
            const float* r = j < n ? hs + j * P : ws + (iw - (j - n) + n - 1) * P;
            float a = 0.f;
            for (int c = 0; c < hd; ++c) a += q[c] * r[c];
            rel[(((size_t)bh * T) + ih * n + iw) * 2 * n + j] = a;
        }
        return;
    }
    for (int e = threadIdx.x; e < n * jb; e += nt) {
        const int iw = e / jb, j0 = (e - iw * jb) * 4;
        const float* q = qs + iw * P;
        const float* r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = min(j0 + u, 2 * n - 1);
            r[u] = j < n ? hs + j * P : ws + (iw - (j - n) + n - 1) * P;
        }
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < hd; ++c) {  // (each sum in the order of the single-j loop it replaces: same bits)
            const float qc = q[c];
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] += qc * r[u][c];
        }
        float* o = rel + (((size_t)bh * T) + ih * n + iw) * 2 * n + j0;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (j0 + u < 2 * n) o[u] = a[u];
    }
}

// The same for SMALL grids (the encoder's 14 x 14 windows): one workgroup per (window, head) instead of one per query row.  With a
// workgroup per row the launch was 5600 workgroups that each staged the whole rel_pos_w table again for 392 dot products -- 55 us per
// launch, all of it staging latency.  Here the window's T = n*n queries (63 KB at n = 14, hd = 80) and both tables are staged once and
// a thread forms four consecutive j of a query per pass (one LDS read of q[c] serves four products; every sum in k_relpos's order).
__global__ __launch_bounds__(512) void k_relpos_window(const float* __restrict__ qkv, int T, int heads, int hd, int n,
                                                       const float* __restrict__ rph, const float* __restrict__ rpw,
                                                       float* __restrict__ rel) {
    extern __shared__ float rpw_lds[];
    const int P = hd + 1, R = 2 * n - 1;
    float* qs = rpw_lds;          // [T][P]
    float* hs = qs + T * P;       // [2n-1][P]  rel_pos_h
    float* ws = hs + R * P;       // [2n-1][P]  rel_pos_w
    const int bh = blockIdx.x, b = bh / heads, h = bh % heads, C = heads * hd, nt = blockDim.x;
    const int hq = hd >> 2;       // float4 per row (hd % 4 == 0, rows 16-byte aligned: checked by the host)
    for (int e = threadIdx.x; e < T * hq; e += nt) {
        const int r = e / hq, c = (e - r * hq) * 4;
        const float4 v = *reinterpret_cast<const float4*>(qkv + ((size_t)b * T + r) * 3 * C + h * hd + c);
        float* d = qs + r * P + c;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    for (int e = threadIdx.x; e < R * hq; e += nt) {
        const int r = e / hq, c = (e - r * hq) * 4;
        const float4 a = *reinterpret_cast<const float4*>(rph + (size_t)r * hd + c);
        const float4 w = *reinterpret_cast<const float4*>(rpw + (size_t)r * hd + c);
        float* dh = hs + r * P + c;
        float* dw = ws + r * P + c;
        dh[0] = a.x; dh[1] = a.y; dh[2] = a.z; dh[3] = a.w;
        dw[0] = w.x; dw[1] = w.y; dw[2] = w.z; dw[3] = w.w;
    }
    __syncthreads();
    const int jb = (2 * n + 3) >> 2;  // blocks of four j per query
    for (int e = threadIdx.x; e < T * jb; e += nt) {
        const int i = e / jb, j0 = (e - i * jb) * 4;
        const int ih = i / n, iw = i - ih * n;
        const float* q = qs + i * P;
        const float* r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = min(j0 + u, 2 * n - 1);
            r[u] = j < n ? hs + (ih - j + n - 1) * P : ws + (iw - (j - n) + n - 1) * P;
        }
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < hd; ++c) {
            const float qc = q[c];
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] += qc * r[u][c];
        }
        float* o = rel + ((size_t)bh * T + i) * 2 * n + j0;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (j0 + u < 2 * n) o[u] = a[u];
    }
}

// DB = ceil(hd / 32) output blocks of 32 head dims
template <int DB>
__global__ __launch_bounds__(256) void k_attention(const float* __restrict__ qkv, const float* __restrict__ rel, int T,
                                                   int heads, int hd, int n, float scale, float* __restrict__ out) {
    constexpr int DP = DB * 32;          // padded head dim
    constexpr int KP = DP + 1;           // K tile pitch (odd: the A-operand reads walk rows)
    __shared__ float Ks[32 * KP];
    __shared__ float Vs[32 * DP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, half = lane >> 5;
    const int bh = blockIdx.y, b = bh / heads, h = bh % heads, C = heads * hd;
    const int q0 = (blockIdx.x * 4 + wave) * 32;
    const int qi = q0 + li;              // this lane's query (both halves hold the same one)
    const bool qlive = qi < T;
    const float* base = qkv + (size_t)b * T * 3 * C + h * hd;
    // the lane's query row, B operand of S^T = K Q^T: element d = 2j + half for step j
    float qreg[DP / 2];
    {
        const float* qp = base + (size_t)(qlive ? qi : T - 1) * 3 * C;
#pragma unroll
        for (int j = 0; j < DP / 2; ++j) {
            const int d = 2 * j + half;
            qreg[j] = (d < hd) ? qp[d] : 0.f;
        }
    }
    // the 32 queries' relative-position rows of this wave, staged once (read every key tile: from global memory that was 64
    // cache lines per load instruction and bounded the kernel)
    extern __shared__ __attribute__((aligned(16))) float rel_lds[];
    const int RP = 2 * n + 1;
    float* relw = rel_lds + (size_t)wave * 32 * RP;
    if (rel) {
        for (int e = lane; e < 32 * 2 * n; e += 64) {
            const int r = e / (2 * n), j = e - r * 2 * n;
            const int qq = q0 + r < T ? q0 + r : T - 1;
            relw[r * RP + j] = rel[((size_t)bh * T + qq) * 2 * n + j];
        }
    }
    const float* relq = rel ? relw + li * RP : nullptr;
    f32x16 o[DB];
#pragma unroll
    for (int t = 0; t < DB; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    // K / V tiles: global -> registers one tile ahead (the loads fly under the current tile's MFMAs), registers -> LDS
    constexpr int EPT = (32 * DP) / 256;  // staged elements per thread and matrix
    float kreg[EPT], vreg[EPT];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int e = threadIdx.x + 256 * q;
            const int kr = e / DP, d = e - kr * DP;
            const int key = k0 + kr;
            const bool ok = key < T && d < hd;
            const float* kp = base + (size_t)(key < T ? key : T - 1) * 3 * C + C + (d < hd ? d : 0);
            kreg[q] = ok ? kp[0] : 0.f;
            vreg[q] = ok ? kp[C] : 0.f;
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < T; k0 += 32) {
        __syncthreads();  // previous tile consumed
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int e = threadIdx.x + 256 * q;
            const int kr = e / DP, d = e - kr * DP;
            Ks[kr * KP + d] = kreg[q];
            Vs[kr * DP + d] = vreg[q];
        }
        __syncthreads();
        if (k0 + 32 < T) fetch(k0 + 32);
        // ---- S^T[key][query]
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int j = 0; j < DP / 2; ++j) s = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[li * KP + 2 * j + half], qreg[j], s, 0, 0, 0);
        // ---- scale, relative-position bias, mask, online softmax (everything per lane = per query)
        float m_tile = -INFINITY;
        const int kh0 = relq ? k0 / n : 0, kw0 = relq ? k0 - kh0 * n : 0;  // one division per tile, not per register
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + vrow(r, half);
            float v = s[r] * scale;
            if (relq) {
                int kh = kh0, kw = kw0 + vrow(r, half);
                while (kw >= n) { kw -= n; ++kh; }
                if (key < T) v += relq[kh] + relq[n + kw];
            }
            v = key < T ? v : -INFINITY;
            s[r] = v;
            m_tile = fmaxf(m_tile, v);
        }
        m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32, 64));
        const float m_new = fmaxf(m_run, m_tile);
        const float alpha = expf(m_run - m_new);  // exp(-inf) = 0 on the first tile
        float l_tile = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = expf(s[r] - m_new);
            s[r] = p;
            l_tile += p;
        }
        l_tile += __shfl_xor(l_tile, 32, 64);
        l_run = l_run * alpha + l_tile;
        m_run = m_new;
        // ---- O^T[d][query] = alpha * O^T + V^T P^T
#pragma unroll
        for (int t = 0; t < DB; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[vrow(r, half) * DP + t * 32 + li], s[r], o[t], 0, 0, 0);
        }
    }
    if (qlive) {
        const float inv = 1.f / l_run;
        float* op = out + ((size_t)b * T + qi) * C + h * hd;
#pragma unroll
        for (int t = 0; t < DB; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = t * 32 + vrow(r, half);
                if (d < hd) op[d] = o[t][r] * inv;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same attention on the bf16 matrix cores with the 3-term split (hi*hi + hi*lo + lo*hi, fp32 accumulate): per 32-key tile
// 36 v_mfma_f32_32x32x16_bf16 (1152 cycles) instead of 96 v_mfma_f32_32x32x2_f32 (6144 cycles) at head dim 80.
//   S^T[key][query] = K Q^T : A = K rows from LDS planes [key][d] (b128 per fragment), B = the lane's query row, split once
//                             per workgroup and pre-multiplied by the softmax scale;
//   O^T[d][query]  += V^T P^T: the accumulator registers 8s..8s+7 of S^T (after the softmax: P^T) are a valid 16-wide k-step
//                             over the keys vrow(8s + j, half) -- the same observation as in mlp_chain.hip -- so P^T is split
//                             in registers, and V is staged transposed with its key axis in exactly that order
//                             (planes [d][slot], slot = 16s + 8 half + j).
// Online softmax, relative positions and masking are those of k_attention.
typedef __bf16 at_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 at_bf16x2 __attribute__((ext_vector_type(2)));
typedef float at_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t at_cvt_pk(float a, float b) {
    const at_f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, at_bf16x2));
}

__device__ __forceinline__ void at_split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    hi = at_cvt_pk(x0, x1);
    lo = at_cvt_pk(x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xFFFF0000u));
}

// key row kr (0..31) of a tile -> its k-slot in the P^T operand order
__device__ __forceinline__ int at_key_slot(int kr) {
    const int hf = (kr >> 2) & 1, rr = (kr & 3) + 4 * (kr >> 3);
    return (rr >> 3) * 16 + hf * 8 + (rr & 7);
}

// two waves per SIMD (<= 256 registers, no spills at head dim 80): with ONE (312 registers) nothing hid the barriers and the
// load latencies of the key loop -- 14x14 windows 0.207 -> 0.128 ms, 64x64 global without positions 0.85 -> 0.53 ms
#define SNF_ATT_WAVES 2
#ifndef SNF_RELPOS_B3
#define SNF_RELPOS_B3 1  // 0: snf_relpos on the vector ALU for every grid size (A/B)
#endif
// NT: threads of the workgroup = 32 queries per wave x NT / 64 waves.  256 everywhere but on the encoder's 14 x 14 windows (T = 196):
// there ONE workgroup of 7 waves (448 threads: 224 query slots) takes a whole (window, head) -- with 128-query workgroups the second
// one ran 68 of its 128 slots and both staged every K / V tile.
template <int DB, bool PL = false, int NT = 256>
__global__ __launch_bounds__(NT, (NT == 256 ? SNF_ATT_WAVES : 1)) void k_attention_b3(const float* __restrict__ qkv, const float* __restrict__ rel, int T,
                                                      int heads, int hd, int n, float scale, float* __restrict__ out,
                                                      int rel_direct, uint16_t* __restrict__ out_hi = nullptr,
                                                      uint16_t* __restrict__ out_lo = nullptr, int M_out = 0,
                                                      const float* __restrict__ rph = nullptr,
                                                      const float* __restrict__ rpw = nullptr) {
    constexpr int DP = DB * 32;          // padded head dim
    constexpr int KS = DP / 16;          // k-steps over the head dim
    constexpr int KPB = DP + 8;          // bf16 pitch of the K planes  [32 keys][DP]
    constexpr int VPB = 40;              // bf16 pitch of the V^T planes [DP][32 key slots]
    __shared__ __attribute__((aligned(16))) uint16_t Kh[32 * KPB];
    __shared__ __attribute__((aligned(16))) uint16_t Kl[32 * KPB];
    __shared__ __attribute__((aligned(16))) uint16_t Vh[DP * VPB];
    __shared__ __attribute__((aligned(16))) uint16_t Vl[DP * VPB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, half = lane >> 5;
    const int bh = blockIdx.y, b = bh / heads, h = bh % heads, C = heads * hd;
    const int q0 = (blockIdx.x * (NT / 64) + wave) * 32;
    const int qi = q0 + li;
    const bool qlive = qi < T;
    const float* base = qkv + (size_t)b * T * 3 * C + h * hd;
    // the lane's (scaled) query row as B operands: k-step s holds d = 16s + 8 half + j
    at_bf16x8 qh[KS], ql[KS];
    {
        // (all loads first, unconditional at clamped dims, the pad dims zeroed after: a guarded load per element was a basic block and a
        //  wait per element -- 48 memory latencies in a row before the first tile, twice that again for the position tables below; on the
        //  14 x 14 windows, 7 tiles per workgroup, that prologue was most of the kernel)
        const float* qp = base + (size_t)(qlive ? qi : T - 1) * 3 * C;
        float2 raw[KS][4];
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int p2 = 0; p2 < 4; ++p2) raw[s][p2] = *reinterpret_cast<const float2*>(qp + min(16 * s + 8 * half + 2 * p2, hd - 2));
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            uint32_t hq[4], lq[4];
#pragma unroll
            for (int p2 = 0; p2 < 4; ++p2) {
                const bool in = 16 * s + 8 * half + 2 * p2 < hd;  // (head dim even)
                at_split2(in ? raw[s][p2].x * scale : 0.f, in ? raw[s][p2].y * scale : 0.f, hq[p2], lq[p2]);
            }
            qh[s] = __builtin_bit_cast(at_bf16x8, make_uint4(hq[0], hq[1], hq[2], hq[3]));
            ql[s] = __builtin_bit_cast(at_bf16x8, make_uint4(lq[0], lq[1], lq[2], lq[3]));
        }
    }
    extern __shared__ __attribute__((aligned(16))) float rel_lds[];
    const int RP = 2 * n + 1;
    const uint32_t inv_n = n > 0 ? (65536u + (uint32_t)n - 1u) / (uint32_t)n : 0u;
    float* relw = rel_lds + (size_t)wave * 32 * RP;
    // rel_direct (grid side a multiple of 32: a 32-key tile lies inside ONE row of keys): the lane reads its query's one row term
    // and 16 column terms per tile straight from global memory (four float4) -- no per-wave copy of the [32][2n] position rows in
    // LDS, which at n = 64 was 66 KB per workgroup and left one workgroup per CU
    // ... the one ROW term a tile needs is requested a tile ahead; the COLUMN terms (the same n values of a query for every row of keys)
    // are copied once into the wave's LDS rows [32][n + 4]: read from global memory they were four 16-byte loads per lane and tile at
    // 2n-float strides -- 256 cache lines per wave and tile, issued between the score MFMAs and the softmax with nothing to hide them
    const float* __restrict__ relg = (rel && rel_direct) ? rel + ((size_t)bh * T + (qlive ? qi : T - 1)) * 2 * n : nullptr;
    const int RWP = n + 4;
    if (relg) {
        float* rw = rel_lds + (size_t)wave * 32 * RWP;
        for (int r = 0; r < 32; ++r) {
            const int qq = q0 + r < T ? q0 + r : T - 1;
            const float* src = rel + ((size_t)bh * T + qq) * 2 * n + n;
            for (int j = lane; j < n; j += 64) rw[r * RWP + j] = src[j];
        }
    }
    const float* relc = rel_lds + (size_t)wave * 32 * RWP + li * RWP;  // (rel_direct only)
    float rrow_cur = relg ? relg[0] : 0.f, rrow_nxt = 0.f;
    const bool has_rel = rel != nullptr || rph != nullptr;
    if (rph != nullptr) {
        // Small grids (2n - 1 <= 32: the encoder's 14 x 14 windows): the decomposed relative-position terms of this wave's 32 queries
        // are formed HERE, on the matrix cores, from the query fragments already in registers -- P^T[r][q] = rel_pos[r] . q for all
        // 2n - 1 table rows (one 32-row MFMA tile per table), then rel_h[q][kh] = P_h[q][ih - kh + n - 1], rel_w[q][kw] =
        // P_w[q][iw - kw + n - 1] scattered into the wave's LDS rows -- instead of a separate kernel (snf_relpos: 37 us per launch, all
        // staging latency) writing them to memory and this one reading them back.  q is scaled here and unscaled there: x 1 / scale.
        const int qc = qlive ? qi : T - 1, ih = qc / n, iw = qc - ih * n;
        const float unscale = 1.f / scale;
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            const bool trow = li < 2 * n - 1;
            const float* __restrict__ tabrow = (tb == 0 ? rph : rpw) + (size_t)(trow ? li : 2 * n - 2) * hd;
            float2 raw[KS][4];
#pragma unroll
            for (int s2 = 0; s2 < KS; ++s2)
#pragma unroll
                for (int p2 = 0; p2 < 4; ++p2) raw[s2][p2] = *reinterpret_cast<const float2*>(tabrow + min(16 * s2 + 8 * half + 2 * p2, hd - 2));
            f32x16 pacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) pacc[r] = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < KS; ++s2) {
                uint32_t hr[4], lr[4];
#pragma unroll
                for (int p2 = 0; p2 < 4; ++p2) {
                    const bool in = trow && 16 * s2 + 8 * half + 2 * p2 < hd;
                    at_split2(in ? raw[s2][p2].x : 0.f, in ? raw[s2][p2].y : 0.f, hr[p2], lr[p2]);
                }
                const at_bf16x8 rh = __builtin_bit_cast(at_bf16x8, make_uint4(hr[0], hr[1], hr[2], hr[3]));
                const at_bf16x8 rl = __builtin_bit_cast(at_bf16x8, make_uint4(lr[0], lr[1], lr[2], lr[3]));
                pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rl, qh[s2], pacc, 0, 0, 0);
                pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rh, ql[s2], pacc, 0, 0, 0);
                pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rh, qh[s2], pacc, 0, 0, 0);
            }
            const int pos = tb == 0 ? ih : iw;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kk = pos + n - 1 - vrow(r, half);  // table row vrow(r, half) serves key row / column kk of this query
                if (kk >= 0 && kk < n) relw[li * RP + tb * n + kk] = pacc[r] * unscale;
            }
        }
    }
    if (rel && !rel_direct) {
        for (int e = lane; e < 32 * 2 * n; e += 64) {
            const int r = e / (2 * n), j = e - r * 2 * n;
            const int qq = q0 + r < T ? q0 + r : T - 1;
            relw[r * RP + j] = rel[((size_t)bh * T + qq) * 2 * n + j];
        }
    }
    const float* relq = (has_rel && !rel_direct) ? relw + li * RP : nullptr;
    f32x16 o[DB];
#pragma unroll
    for (int t = 0; t < DB; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    // staging (threads 0 .. 255; head dim even), one tile ahead in registers:
    //   K: thread (key kr = t / 8, c = t % 8) takes the dim pairs 2 (c + 8 q), q < DP / 16 -- 64 contiguous bytes per 8 lanes;
    //   V: thread (key pair kp2 = t / 16, d0 = t % 16) takes dims d0 + 16 q of the keys 2 kp2, 2 kp2 + 1.
    // Every address is the wave-uniform head base + a 32-bit lane offset (row term once per tile, the q term a constant), and every load is
    // unconditional (clamped row / dim, zeroed after): as (e / 48, e % 48) index arithmetic with a branch per guarded load the fetch was 250
    // VALU instructions and 24 basic blocks per tile, and each branch made the compiler drain the loads in flight.
    constexpr int SQ = DP / 16;
    const int st = (int)threadIdx.x;
    const bool stager = NT == 256 || st < 256;  // (wave-uniform: 256 threads = 4 waves)
    const int s_kr = (st >> 3) & 31, s_c = st & 7, s_kp2 = (st >> 4) & 15, s_d = st & 15;
    const int s_slot = at_key_slot(2 * s_kp2);  // keys 2kp2 and 2kp2+1 sit in adjacent slots
    const int row3c = 3 * C;
    float2 kreg[SQ], vreg[SQ];
    auto fetch = [&](int k0) {
        const int kkey = k0 + s_kr, vkey = k0 + 2 * s_kp2;
        // (unsigned 32-bit offsets from the uniform base: scalar base + vector offset addressing, no 64-bit lane arithmetic; T 3C < 2^31: host)
        const uint32_t koff = (uint32_t)(min(kkey, T - 1) * row3c + C);
        const uint32_t voff0 = (uint32_t)(min(vkey, T - 1) * row3c + 2 * C), voff1 = (uint32_t)(min(vkey + 1, T - 1) * row3c + 2 * C);
#pragma unroll
        for (int q = 0; q < SQ; ++q) {
            const int dk = 2 * (s_c + 8 * q), dv = s_d + 16 * q;
            kreg[q] = *reinterpret_cast<const float2*>(base + (koff + (uint32_t)min(dk, hd - 2)));
            vreg[q].x = base[voff0 + (uint32_t)min(dv, hd - 1)];
            vreg[q].y = base[voff1 + (uint32_t)min(dv, hd - 1)];
        }
    };
    if (stager) fetch(0);
    for (int k0 = 0; k0 < T; k0 += 32) {
        __syncthreads();  // previous tile consumed
        if (stager) {
            // (pad dims and keys past T are zeroed HERE, where the values are consumed: a select right behind the load made the fetch
            //  wait for every load as it issued it; one wave-uniform branch around all passes, not one per pass)
            if (DP > hd) {
#pragma unroll
                for (int q = 0; q < SQ; ++q) {
                    if (2 * (s_c + 8 * q) >= hd) kreg[q] = make_float2(0.f, 0.f);
                    if (s_d + 16 * q >= hd) vreg[q] = make_float2(0.f, 0.f);
                }
            }
            if (k0 + 32 > T) {  // only the last tile of a T that is no multiple of 32
#pragma unroll
                for (int q = 0; q < SQ; ++q) {
                    if (k0 + s_kr >= T) kreg[q] = make_float2(0.f, 0.f);
                    if (k0 + 2 * s_kp2 >= T) vreg[q].x = 0.f;
                    if (k0 + 2 * s_kp2 + 1 >= T) vreg[q].y = 0.f;
                }
            }
#pragma unroll
            for (int q = 0; q < SQ; ++q) {
                uint32_t hh, ll;
                at_split2(kreg[q].x, kreg[q].y, hh, ll);
                *reinterpret_cast<uint32_t*>(&Kh[s_kr * KPB + 2 * (s_c + 8 * q)]) = hh;
                *reinterpret_cast<uint32_t*>(&Kl[s_kr * KPB + 2 * (s_c + 8 * q)]) = ll;
                at_split2(vreg[q].x, vreg[q].y, hh, ll);
                *reinterpret_cast<uint32_t*>(&Vh[(s_d + 16 * q) * VPB + s_slot]) = hh;
                *reinterpret_cast<uint32_t*>(&Vl[(s_d + 16 * q) * VPB + s_slot]) = ll;
            }
        }
        __syncthreads();
        if (stager && k0 + 32 < T) fetch(k0 + 32);
        if (relg) rrow_nxt = relg[min((k0 + 32) / n, n - 1)];
        // ---- S^T[key][query] (already scaled)
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const at_bf16x8 kh = *reinterpret_cast<const at_bf16x8*>(&Kh[li * KPB + 16 * ks + 8 * half]);
            const at_bf16x8 kl = *reinterpret_cast<const at_bf16x8*>(&Kl[li * KPB + 16 * ks + 8 * half]);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh[ks], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[ks], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[ks], s, 0, 0, 0);
        }
        // ---- relative-position bias, mask, online softmax (everything per lane = per query)
        float m_tile = -INFINITY;
        const bool ragged = k0 + 32 > T;
        const int kh0 = has_rel ? k0 / n : 0, kw0 = has_rel ? k0 - kh0 * n : 0;
        // position bias of the lane's 16 scores, gathered FIRST (one launch-uniform branch around all 16, not one per score: the per-score
        // `if (relq)` left 16 basic blocks, each waiting for its own two LDS reads)
        float bias[16];
        if (relg) {  // registers 4g .. 4g+3 are the keys k0 + 8g + 4 half + {0..3}
            const float rrow = rrow_cur;
            rrow_cur = rrow_nxt;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 c4 = *reinterpret_cast<const float4*>(relc + kw0 + 8 * g4 + 4 * half);
                bias[4 * g4] = rrow + c4.x; bias[4 * g4 + 1] = rrow + c4.y; bias[4 * g4 + 2] = rrow + c4.z; bias[4 * g4 + 3] = rrow + c4.w;
            }
        } else if (relq) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // key -> (row, column) of the n x n grid with one multiply (exact for key < 2^16, n < 2^8: inv_n = 2^16 / n rounded
                // up); the while loop this replaces cost ~10 VALU instructions per element
                const int key = k0 + vrow(r, half);
                const int kc = key < T ? key : T - 1;
                const int kh = (int)(((uint32_t)kc * inv_n) >> 16), kw = kc - kh * n;
                bias[r] = relq[kh] + relq[n + kw];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) bias[r] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + vrow(r, half);
            float v = s[r] + bias[r];
            if (ragged) v = key < T ? v : -INFINITY;  // (only the last tile of a T that is no multiple of 32 has keys past T)
            s[r] = v;
            m_tile = fmaxf(m_tile, v);
        }
        m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32, 64));
        const float m_new = fmaxf(m_run, m_tile);
        // e^x = 2^(x log2 e) on the hardware exponential (v_exp_f32, 1 ulp): two instructions per weight where expf() spends ~15
        // on range reduction the arguments (<= 0, scores of magnitude ~10) do not need; the product's rounding moves a weight by
        // <= |x| 2^-24 relative -- below the 1e-6 the bf16 split already puts on the scores
        constexpr float LOG2E = 1.4426950408889634f;
        const float mb = m_new * LOG2E;
        const float alpha = __builtin_amdgcn_exp2f(m_run * LOG2E - mb);  // (v_exp_f32; results below 2^-126 flush to 0)
        float l_tile = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pv = __builtin_amdgcn_exp2f(s[r] * LOG2E - mb);
            s[r] = pv;
            l_tile += pv;
        }
        l_tile += __shfl_xor(l_tile, 32, 64);
        l_run = l_run * alpha + l_tile;
        m_run = m_new;
        // ---- O^T[d][query] = alpha * O^T + V^T P^T
        at_bf16x8 ph[2], pl[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            uint32_t hp[4], lp[4];
#pragma unroll
            for (int p2 = 0; p2 < 4; ++p2) at_split2(s[8 * s2 + 2 * p2], s[8 * s2 + 2 * p2 + 1], hp[p2], lp[p2]);
            ph[s2] = __builtin_bit_cast(at_bf16x8, make_uint4(hp[0], hp[1], hp[2], hp[3]));
            pl[s2] = __builtin_bit_cast(at_bf16x8, make_uint4(lp[0], lp[1], lp[2], lp[3]));
        }
        // (the running maximum settles after the first tiles: once no lane of the wave moved it, alpha is exactly 1 everywhere)
        const bool rescale = __any(alpha != 1.f);
#pragma unroll
        for (int t = 0; t < DB; ++t) {
            if (rescale) {
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const at_bf16x8 vh = *reinterpret_cast<const at_bf16x8*>(&Vh[(t * 32 + li) * VPB + 16 * s2 + 8 * half]);
                const at_bf16x8 vl = *reinterpret_cast<const at_bf16x8*>(&Vl[(t * 32 + li) * VPB + 16 * s2 + 8 * half]);
                o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph[s2], o[t], 0, 0, 0);
                o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl[s2], o[t], 0, 0, 0);
                o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph[s2], o[t], 0, 0, 0);
            }
        }
    }
    if constexpr (PL) {
      if (qlive) {
        // the projection GEMM's operand (csrc/gemm_planes.hip): bf16 hi / lo k-blocked planes [C/8][M_out][8].  Registers 4q .. 4q+3
        // of a lane are the four consecutive features 32 t + 8 q + 4 half + {0..3} of its query: one 8-byte store per plane
        // (hd % 8 == 0: a head starts on a k-block)
        const float inv = 1.f / l_run;
        const size_t row = (size_t)b * T + qi;
#pragma unroll
        for (int t = 0; t < DB; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int d0 = t * 32 + 8 * q + 4 * half;
                if (d0 < hd) {  // hd % 4 == 0
                    uint32_t h0, h1, l0, l1;
                    at_split2(o[t][4 * q] * inv, o[t][4 * q + 1] * inv, h0, l0);
                    at_split2(o[t][4 * q + 2] * inv, o[t][4 * q + 3] * inv, h1, l1);
                    const int c = h * hd + d0;
                    const size_t oo = ((size_t)(c >> 3) * M_out + row) * 8 + (c & 7);
                    *reinterpret_cast<uint2*>(out_hi + oo) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2*>(out_lo + oo) = make_uint2(l0, l1);
                }
            }
      }
    } else if (qlive) {
        const float inv = 1.f / l_run;
        float* op = out + ((size_t)b * T + qi) * C + h * hd;
#pragma unroll
        for (int t = 0; t < DB; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = t * 32 + vrow(r, half);
                if (d < hd) op[d] = o[t][r] * inv;
            }
    }
}

// snf_relpos for LARGE grids (the encoder's 64 x 64 global blocks) on the matrix cores: rel = Q . table^T is a [T, hd] x [hd, 2n-1]
// product per (image, head) and table -- 8 GFLOP per launch on the 3-term split -- which k_relpos formed on the vector ALU from LDS
// (one LDS read per multiply-add: 0.13 ms per launch).  A wave takes 32 queries: its query fragments as in k_attention_b3 (unscaled),
// the table rows of a 32-row tile straight from global memory (40 KB per table: cache resident) as the first operand, 15 MFMAs per
// (table, row tile); accumulator register r of lane (query, half) is P[query][row vrow(r, half)], i.e. the term of key row / column
// kk = i + n - 1 - row of that query, stored where k_relpos stores it.  The products are the in-kernel ones of the windowed blocks
// (k_attention_b3's `rph` path): same 3-term arithmetic, same order.
template <int DB>
__global__ __launch_bounds__(256) void k_relpos_b3(const float* __restrict__ qkv, int T, int heads, int hd, int n,
                                                   const float* __restrict__ rph, const float* __restrict__ rpw,
                                                   float* __restrict__ rel) {
    constexpr int KS = DB * 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, half = lane >> 5;
    const int bh = blockIdx.y, b = bh / heads, h = bh % heads, C = heads * hd;
    const int qi = (blockIdx.x * 4 + wave) * 32 + li;
    const bool qlive = qi < T;
    const int qc = qlive ? qi : T - 1, ih = qc / n, iw = qc - ih * n;
    const float* qp = qkv + ((size_t)b * T + qc) * 3 * C + h * hd;
    at_bf16x8 qh[KS], ql[KS];
    {
        float2 raw[KS][4];
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int p2 = 0; p2 < 4; ++p2) raw[s][p2] = *reinterpret_cast<const float2*>(qp + min(16 * s + 8 * half + 2 * p2, hd - 2));
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            uint32_t hq[4], lq[4];
#pragma unroll
            for (int p2 = 0; p2 < 4; ++p2) {
                const bool in = 16 * s + 8 * half + 2 * p2 < hd;
                at_split2(in ? raw[s][p2].x : 0.f, in ? raw[s][p2].y : 0.f, hq[p2], lq[p2]);
            }
            qh[s] = __builtin_bit_cast(at_bf16x8, make_uint4(hq[0], hq[1], hq[2], hq[3]));
            ql[s] = __builtin_bit_cast(at_bf16x8, make_uint4(lq[0], lq[1], lq[2], lq[3]));
        }
    }
    float* __restrict__ orow = rel + ((size_t)bh * T + qc) * 2 * n;
    const int nrows = 2 * n - 1, rtiles = (nrows + 31) >> 5;
    for (int tb = 0; tb < 2; ++tb) {
        const float* __restrict__ tab = tb == 0 ? rph : rpw;
        const int pos = tb == 0 ? ih : iw;
        for (int rt = 0; rt < rtiles; ++rt) {
            const int row = rt * 32 + li;
            const bool trow = row < nrows;
            const float* __restrict__ tabrow = tab + (size_t)(trow ? row : nrows - 1) * hd;
            float2 raw[KS][4];
#pragma unroll
            for (int s2 = 0; s2 < KS; ++s2)
#pragma unroll
                for (int p2 = 0; p2 < 4; ++p2) raw[s2][p2] = *reinterpret_cast<const float2*>(tabrow + min(16 * s2 + 8 * half + 2 * p2, hd - 2));
            f32x16 pacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) pacc[r] = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < KS; ++s2) {
                uint32_t hr[4], lr[4];
#pragma unroll
                for (int p2 = 0; p2 < 4; ++p2) {
                    const bool in = trow && 16 * s2 + 8 * half + 2 * p2 < hd;
                    at_split2(in ? raw[s2][p2].x : 0.f, in ? raw[s2][p2].y : 0.f, hr[p2], lr[p2]);
                }
                const at_bf16x8 rh = __builtin_bit_cast(at_bf16x8, make_uint4(hr[0], hr[1], hr[2], hr[3]));
                const at_bf16x8 rl = __builtin_bit_cast(at_bf16x8, make_uint4(lr[0], lr[1], lr[2], lr[3]));
                pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rl, qh[s2], pacc, 0, 0, 0);
                pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rh, ql[s2], pacc, 0, 0, 0);
                pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rh, qh[s2], pacc, 0, 0, 0);
            }
            if (qlive) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kk = pos + n - 1 - (rt * 32 + vrow(r, half));  // table row -> key row / column of this query
                    if (kk >= 0 && kk < n) orow[tb * n + kk] = pacc[r];
                }
            }
        }
    }
}

}  // namespace snf

using namespace snf;

namespace snf {
// Sam.preprocess (segment_anything/modeling/sam.py:164-174): (x - pixel_mean) / pixel_std per channel, zero-padded right / bottom
// to the encoder's square S x S input.  One thread per output pixel group of 4; uint8 (what SamPredictor.set_image hands over,
// predictor.py:58-66) or float input.
template <typename T>
__global__ __launch_bounds__(256) void k_sam_preprocess(const T* __restrict__ img, int B, int C, int h, int w, int S,
                                                        const float* __restrict__ mean, const float* __restrict__ stdv,
                                                        float* __restrict__ out) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)B * C * S * (S / 4);
    if (t >= total) return;
    const int xq = (int)(t % (S / 4));
    const int y = (int)((t / (S / 4)) % S);
    const int c = (int)((t / ((long long)(S / 4) * S)) % C);
    const int b = (int)(t / ((long long)(S / 4) * S * C));
    const float m = mean[c], sd = stdv[c];
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x = xq * 4 + j;
        v[j] = (y < h && x < w) ? ((float)img[(((size_t)b * C + c) * h + y) * w + x] - m) / sd : 0.f;
    }
    *reinterpret_cast<float4*>(out + (((size_t)b * C + c) * S + y) * S + xq * 4) = make_float4(v[0], v[1], v[2], v[3]);
}
}  // namespace snf

extern "C" int snf_sam_preprocess(const void* img, int is_uint8, int B, int C, int h, int w, int S, const float* mean,
                                  const float* stdv, float* out, snf_stream_t stream) {
    SNF_REQUIRE(img && mean && stdv && out, "snf_sam_preprocess: null pointer");
    SNF_REQUIRE(B > 0 && C > 0 && h > 0 && w > 0 && h <= S && w <= S && S % 4 == 0 && ((uintptr_t)out % 16) == 0,
                "snf_sam_preprocess: bad shape B=%d C=%d h=%d w=%d S=%d (the image must fit the S x S input, S %% 4 == 0)", B, C, h, w, S);
    const long long total = (long long)B * C * S * (S / 4);
    dim3 grid((unsigned)((total + 255) / 256));
    if (is_uint8)
        hipLaunchKernelGGL(snf::k_sam_preprocess<unsigned char>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned char*)img, B, C,
                           h, w, S, mean, stdv, out);
    else
        hipLaunchKernelGGL(snf::k_sam_preprocess<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)img, B, C, h, w, S, mean,
                           stdv, out);
    SNF_LAUNCH_CHECK("snf_sam_preprocess");
    return SNF_OK;
}

extern "C" int snf_patchify(const float* img, int B, int Cin, int S, int P, float* rows, snf_stream_t stream) {
    SNF_REQUIRE(img && rows && B > 0 && Cin > 0 && P > 0 && S > 0 && S % P == 0, "snf_patchify: bad argument");
    const long long total = (long long)B * (S / P) * (S / P) * Cin * P * P;
    hipLaunchKernelGGL(k_patchify, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, img, B, Cin, S, P,
                       rows);
    SNF_LAUNCH_CHECK("snf_patchify");
    return SNF_OK;
}

extern "C" int snf_layernorm(const float* x, const float* residual, int N, int C, const float* weight, const float* bias,
                             float eps, float* sum_out, float* y, snf_stream_t stream) {
    SNF_REQUIRE(x && weight && bias && y && N > 0 && C > 0, "snf_layernorm: bad argument");
    const bool al16 = ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)weight | (uintptr_t)bias | (uintptr_t)residual | (uintptr_t)sum_out) & 15) == 0);
    static const int reg_on = 1;
    const int nv = (reg_on && al16 && C % 256 == 0) ? C / 256 : 0;
#define SNF_LN(NV_) hipLaunchKernelGGL(k_layernorm_reg<NV_>, dim3(ceil_div(N, 4)), dim3(256), 0, (hipStream_t)stream, x, residual, N, \
                                       C, weight, bias, eps, sum_out, y)
    if (nv == 1) SNF_LN(1); else if (nv == 2) SNF_LN(2); else if (nv == 3) SNF_LN(3); else if (nv == 4) SNF_LN(4);
    else if (nv == 5) SNF_LN(5); else if (nv == 6) SNF_LN(6); else if (nv == 8) SNF_LN(8);
    else
#undef SNF_LN
    hipLaunchKernelGGL(k_layernorm, dim3(ceil_div(N, 4)), dim3(256), 0, (hipStream_t)stream, x, residual, N, C, weight, bias,
                       eps, sum_out, y);
    SNF_LAUNCH_CHECK("snf_layernorm");
    return SNF_OK;
}

static int layernorm_planes_launch(const float* x, const float* residual, int N, int C, const float* weight, const float* bias,
                                   float eps, float* sum_out, uint16_t* y_hi, uint16_t* y_lo, int M_out, int H, int W, int ws, int res_ws,
                                   snf_stream_t stream) {
    SNF_REQUIRE(x && weight && bias && y_hi && y_lo && N > 0 && C > 0, "snf_layernorm_planes: bad argument");
    SNF_REQUIRE(C % 256 == 0 && C / 256 <= 8 && C / 256 != 7,
                "snf_layernorm_planes: C=%d (the row is held in registers: a multiple of 256 up to 2048, not 1792)", C);
    SNF_REQUIRE(((((uintptr_t)x | (uintptr_t)y_hi | (uintptr_t)y_lo | (uintptr_t)weight | (uintptr_t)bias | (uintptr_t)residual |
                   (uintptr_t)sum_out) & 15) == 0), "snf_layernorm_planes: unaligned pointer");
    if (ws > 0) {
        SNF_REQUIRE(H > 0 && W > 0 && N % (H * W) == 0, "snf_layernorm_planes: N=%d is not whole [H=%d, W=%d] grids", N, H, W);
        const long long wins = (long long)(N / (H * W)) * ((H + ws - 1) / ws) * ((W + ws - 1) / ws);
        SNF_REQUIRE((long long)M_out == wins * ws * ws, "snf_layernorm_planes: M_out=%d != %lld window rows", M_out, wins * ws * ws);
    } else {
        SNF_REQUIRE(M_out == N, "snf_layernorm_planes: M_out=%d != N=%d without windows", M_out, N);
    }
    const int nv = C / 256;
#define SNF_LNP(NV_) hipLaunchKernelGGL(k_layernorm_planes<NV_>, dim3(ceil_div(N, 4)), dim3(256), 0, (hipStream_t)stream, x, residual, \
                                        N, C, weight, bias, eps, sum_out, y_hi, y_lo, M_out, H, W, ws, res_ws)
    if (nv == 1) SNF_LNP(1); else if (nv == 2) SNF_LNP(2); else if (nv == 3) SNF_LNP(3); else if (nv == 4) SNF_LNP(4);
    else if (nv == 5) SNF_LNP(5); else if (nv == 6) SNF_LNP(6); else SNF_LNP(8);
#undef SNF_LNP
    SNF_LAUNCH_CHECK("snf_layernorm_planes");
    return SNF_OK;
}

extern "C" int snf_layernorm_planes(const float* x, const float* residual, int N, int C, const float* weight, const float* bias,
                                    float eps, float* sum_out, uint16_t* y_hi, uint16_t* y_lo, int M_out, int H, int W, int ws,
                                    snf_stream_t stream) {
    return layernorm_planes_launch(x, residual, N, C, weight, bias, eps, sum_out, y_hi, y_lo, M_out, H, W, ws, 0, stream);
}

// snf_layernorm_planes whose residual is still in window rows ([B * nWh * nWw * res_ws^2, C], the projection's output): the
// window_unpartition + `shortcut + x` of the block (snf_window_merge_add) folded into norm2; sum_out = the merged token rows
extern "C" int snf_layernorm_planes_merge(const float* x, const float* residual_windows, int N, int C, const float* weight,
                                          const float* bias, float eps, float* sum_out, uint16_t* y_hi, uint16_t* y_lo, int H, int W,
                                          int res_ws, snf_stream_t stream) {
    SNF_REQUIRE(residual_windows && sum_out && res_ws > 0 && H > 0 && W > 0 && N % (H * W) == 0,
                "snf_layernorm_planes_merge: bad argument (N=%d must be whole [H=%d, W=%d] grids)", N, H, W);
    return layernorm_planes_launch(x, residual_windows, N, C, weight, bias, eps, sum_out, y_hi, y_lo, N, H, W, 0, res_ws, stream);
}

extern "C" int snf_window_partition(const float* x, int B, int H, int W, int C, int ws, float* out, snf_stream_t stream) {
    SNF_REQUIRE(x && out && B > 0 && H > 0 && W > 0 && C > 0 && ws > 0, "snf_window_partition: bad argument");
    const int nW = ((H + ws - 1) / ws) * ((W + ws - 1) / ws);
    hipLaunchKernelGGL(k_window_partition, dim3(B * nW * ws * ws), dim3(256), 0, (hipStream_t)stream, x, B, H, W, C, ws, out);
    SNF_LAUNCH_CHECK("snf_window_partition");
    return SNF_OK;
}

extern "C" int snf_window_merge_add(const float* windows, const float* shortcut, int B, int H, int W, int C, int ws,
                                    float* out, snf_stream_t stream) {
    SNF_REQUIRE(windows && shortcut && out && B > 0 && H > 0 && W > 0 && C > 0 && ws >= 0, "snf_window_merge_add: bad argument");
    hipLaunchKernelGGL(k_window_merge_add, dim3(B * H * W), dim3(256), 0, (hipStream_t)stream, windows, shortcut, B, H, W, C, ws,
                       out);
    SNF_LAUNCH_CHECK("snf_window_merge_add");
    return SNF_OK;
}

extern "C" int snf_relpos(const float* qkv, int Bw, int T, int heads, int head_dim, int n, const float* rel_pos_h,
                          const float* rel_pos_w, float* rel, snf_stream_t stream) {
    SNF_REQUIRE(qkv && rel_pos_h && rel_pos_w && rel && Bw > 0 && heads > 0 && head_dim > 0 && n > 0 && T == n * n,
                "snf_relpos: bad argument (T must be n*n)");
    // small grids: the window's queries and both tables in one workgroup (k_relpos_window)
    const size_t lds_w = (size_t)(T + 2 * (2 * n - 1)) * (head_dim + 1) * sizeof(float);
    if (n <= 16 && lds_w <= 100 * 1024 && (head_dim % 4) == 0 &&
        ((((uintptr_t)qkv | (uintptr_t)rel_pos_h | (uintptr_t)rel_pos_w) & 15) == 0) && ((heads * head_dim) % 4) == 0) {
        static int attr = 0;
        if ((int)lds_w > 48 * 1024 && (int)lds_w > attr) {
            attr = (int)lds_w;
            (void)hipFuncSetAttribute((const void*)k_relpos_window, hipFuncAttributeMaxDynamicSharedMemorySize, attr);
        }
        hipLaunchKernelGGL(k_relpos_window, dim3(Bw * heads), dim3(512), lds_w, (hipStream_t)stream, qkv, T, heads, head_dim, n,
                           rel_pos_h, rel_pos_w, rel);
        SNF_LAUNCH_CHECK("snf_relpos");
        return SNF_OK;
    }
    // large grids in the bf16-split gemm mode: the matrix-core kernel
    if (SNF_RELPOS_B3 && b3_enabled() && n >= 32 && head_dim <= 96 && (head_dim % 2) == 0 &&
        ((((uintptr_t)qkv | (uintptr_t)rel_pos_h | (uintptr_t)rel_pos_w) & 7) == 0) && ((heads * head_dim) % 2) == 0) {
        const dim3 grid(ceil_div(T, 128), Bw * heads);
        const int DB = (head_dim + 31) / 32;
        if (DB == 1) hipLaunchKernelGGL(k_relpos_b3<1>, grid, dim3(256), 0, (hipStream_t)stream, qkv, T, heads, head_dim, n, rel_pos_h, rel_pos_w, rel);
        else if (DB == 2) hipLaunchKernelGGL(k_relpos_b3<2>, grid, dim3(256), 0, (hipStream_t)stream, qkv, T, heads, head_dim, n, rel_pos_h, rel_pos_w, rel);
        else hipLaunchKernelGGL(k_relpos_b3<3>, grid, dim3(256), 0, (hipStream_t)stream, qkv, T, heads, head_dim, n, rel_pos_h, rel_pos_w, rel);
        SNF_LAUNCH_CHECK("snf_relpos");
        return SNF_OK;
    }
    const size_t lds = (size_t)(4 * n - 1) * (head_dim + 1) * sizeof(float);
    SNF_REQUIRE(lds <= 160 * 1024, "snf_relpos: n * head_dim too large for the LDS staging");
    if (lds > 48 * 1024) hipFuncSetAttribute((const void*)k_relpos, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int threads = lds > 40 * 1024 ? 1024 : 256;  // LDS-limited to one or two workgroups per CU: make them big
    hipLaunchKernelGGL(k_relpos, dim3(n, Bw * heads), dim3(threads), lds, (hipStream_t)stream, qkv, Bw, T, heads, head_dim, n,
                       rel_pos_h, rel_pos_w, rel);
    SNF_LAUNCH_CHECK("snf_relpos");
    return SNF_OK;
}

static int attention_launch(const float* qkv, const float* rel, int Bw, int T, int heads, int head_dim, int n, float scale,
                            float* out, uint16_t* out_hi, uint16_t* out_lo, snf_stream_t stream, const float* rph = nullptr,
                            const float* rpw = nullptr) {
    SNF_REQUIRE(qkv && (out || out_hi) && Bw > 0 && T > 0 && heads > 0 && head_dim > 0 && head_dim <= 96,
                "snf_attention: bad argument (head_dim <= 96)");
    SNF_REQUIRE(!rel || (n > 0 && T == n * n), "snf_attention: relative positions need T == n*n");
    dim3 grid(ceil_div(T, 128), Bw * heads);
    const int DB = (head_dim + 31) / 32;
    static const int b3_env = 1;
    static const int direct_env = 1;
    // (the b3 kernel is the only reader of the [32][n + 4] layout: its own launch conditions are part of this one, otherwise the
    //  fall-back k_attention would stage [32][2n + 1] floats per wave into an LDS sized for the smaller layout)
    const bool b3_ok = b3_env && b3_enabled() && (head_dim % 2) == 0 && ((uintptr_t)qkv & 7) == 0;
    const int rel_direct = (rel && direct_env && b3_ok && (n % 32) == 0 && (((uintptr_t)rel) & 15) == 0) ? 1 : 0;
    SNF_REQUIRE(!rph || (rpw && !rel && n > 0 && 2 * n - 1 <= 32 && T == n * n && b3_env && b3_enabled() && (head_dim % 2) == 0 &&
                         (((uintptr_t)rph | (uintptr_t)rpw | (uintptr_t)qkv) & 7) == 0),
                "snf_attention_planes_rp: tables need T == n*n, 2n-1 <= 32, an even head dim, 8-byte aligned pointers and the bf16-split gemm mode");
    const size_t lds = ((rel && !rel_direct) || rph) ? (size_t)4 * 32 * (2 * n + 1) * sizeof(float)
                                                     : rel_direct ? (size_t)4 * 32 * (n + 4) * sizeof(float) : 0;
    SNF_REQUIRE(lds <= 100 * 1024, "snf_attention: grid side n=%d too large for the relative-position staging", n);
    // one 7-wave workgroup per (window, head) where two 4-wave ones would leave the second mostly empty (the 14 x 14 windows)
    const bool wide = out_hi != nullptr && T > 128 && T <= 224;
    const size_t lds7 = lds / 4 * 7;
#define SNF_ATT(DB_)                                                                                                        \
    do {                                                                                                                    \
        if (lds > 32 * 1024)                                                                                                \
            hipFuncSetAttribute((const void*)k_attention<DB_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);       \
        hipLaunchKernelGGL(k_attention<DB_>, grid, dim3(256), lds, (hipStream_t)stream, qkv, rel, T, heads, head_dim, n, scale, \
                           out);                                                                                            \
    } while (0)
    SNF_REQUIRE((long long)T * 3 * heads * head_dim < (1LL << 31), "snf_attention: T x 3C too large for 32-bit row offsets");
    if (b3_ok) {
#define SNF_ATT_B3(DB_)                                                                                                     \
    do {                                                                                                                    \
        if (lds > 16 * 1024)                                                                                                \
        {                                                                                                                   \
            (void)hipFuncSetAttribute((const void*)k_attention_b3<DB_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
            (void)hipFuncSetAttribute((const void*)k_attention_b3<DB_, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        }                                                                                                                   \
        if (out_hi && wide) {                                                                                        \
            (void)hipFuncSetAttribute((const void*)k_attention_b3<DB_, true, 448>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                      (int)lds7);                                                                               \
            hipLaunchKernelGGL((k_attention_b3<DB_, true, 448>), dim3(1, Bw * heads), dim3(448), lds7, (hipStream_t)stream, qkv,   \
                               rel, T, heads, head_dim, n, scale, out, rel_direct, out_hi, out_lo, Bw * T, rph, rpw);           \
        } else if (out_hi)                                                                                                  \
            hipLaunchKernelGGL((k_attention_b3<DB_, true>), grid, dim3(256), lds, (hipStream_t)stream, qkv, rel, T, heads,      \
                               head_dim, n, scale, out, rel_direct, out_hi, out_lo, Bw * T, rph, rpw);                         \
        else                                                                                                                \
            hipLaunchKernelGGL((k_attention_b3<DB_, false>), grid, dim3(256), lds, (hipStream_t)stream, qkv, rel, T, heads,     \
                               head_dim, n, scale, out, rel_direct, out_hi, out_lo, Bw * T);                                   \
    } while (0)
        if (DB == 1) SNF_ATT_B3(1); else if (DB == 2) SNF_ATT_B3(2); else SNF_ATT_B3(3);
#undef SNF_ATT_B3
    } else {
        SNF_REQUIRE(out_hi == nullptr, "snf_attention_planes: needs the bf16-split gemm mode (snf_set_gemm_mode(1))");
        if (DB == 1) SNF_ATT(1); else if (DB == 2) SNF_ATT(2); else SNF_ATT(3);
    }
#undef SNF_ATT
    SNF_LAUNCH_CHECK("snf_attention");
    return SNF_OK;
}

extern "C" int snf_attention(const float* qkv, const float* rel, int Bw, int T, int heads, int head_dim, int n, float scale,
                             float* out, snf_stream_t stream) {
    SNF_REQUIRE(out, "snf_attention: null output");
    return attention_launch(qkv, rel, Bw, T, heads, head_dim, n, scale, out, nullptr, nullptr, stream);
}

// snf_attention_planes for small grids (2n - 1 <= 32) with the decomposed relative-position terms formed inside the kernel from the
// tables rel_pos_h / rel_pos_w [2n-1][head_dim] (snf_relpos + snf_attention_planes in one launch)
extern "C" int snf_attention_planes_rp(const float* qkv, const float* rel_pos_h, const float* rel_pos_w, int Bw, int T, int heads,
                                       int head_dim, int n, float scale, uint16_t* out_hi, uint16_t* out_lo, snf_stream_t stream) {
    SNF_REQUIRE(out_hi && out_lo && rel_pos_h && rel_pos_w && (head_dim % 8) == 0 && (((uintptr_t)out_hi | (uintptr_t)out_lo) & 15) == 0,
                "snf_attention_planes_rp: null / unaligned pointer or head_dim=%d not a multiple of 8", head_dim);
    return attention_launch(qkv, nullptr, Bw, T, heads, head_dim, n, scale, nullptr, out_hi, out_lo, stream, rel_pos_h, rel_pos_w);
}

// snf_attention with the output written as the projection GEMM's operand: bf16 hi / lo k-blocked planes [C/8][Bw*T][8]
extern "C" int snf_attention_planes(const float* qkv, const float* rel, int Bw, int T, int heads, int head_dim, int n, float scale,
                                    uint16_t* out_hi, uint16_t* out_lo, snf_stream_t stream) {
    SNF_REQUIRE(out_hi && out_lo && (head_dim % 8) == 0 && (((uintptr_t)out_hi | (uintptr_t)out_lo) & 15) == 0,
                "snf_attention_planes: null / unaligned planes or head_dim=%d not a multiple of 8", head_dim);
    return attention_launch(qkv, rel, Bw, T, heads, head_dim, n, scale, nullptr, out_hi, out_lo, stream);
}

// fused_head.hip -- eval / render path: a feature head's two hash grids and its first (hidden) layer in ONE kernel, with the
// per-sample features staged in LDS (gfx950).
//
// Reference: SAMField.get_outputs (samnerf/sam_field.py:112-140: two F = 8 hash grids at the top-K sample positions -> cat ->
// tcnn CutlassMLP) followed by MeanRenderer (samnerf/sam_model.py:126-137), in the no-grad render pass of
// samnerf/sam_model.py:337-419.  The train step keeps grid and layer apart on purpose -- the first layer's weight gradient needs the
// encoding in HBM -- but a render pass has no backward, so the [N, 192] encoding (805 MB written and read
// back per 512 x 512 image) never has to exist:
//
//   phase 1  a workgroup takes 64 samples (4 rays x K = 16).  Its 256 threads evaluate the 64 x 24 (sample, level) pairs -- a thread
//            owns six of them, consecutive lanes = consecutive samples of ONE level slab, 16 x 16-byte corner loads in flight,
//            trilinear blend in the reference's order (encodings.py:308-349) -- and write the 8 features of a pair as bf16 hi + lo
//            planes into LDS: A[sample][level * 8 + f], the A operand of the layer, already split for the 3-product arithmetic.
//   phase 2  four waves = 2 row blocks of 32 samples x 2 column halves of 128 units: A fragments from LDS (one ds_read_b128 per
//            plane and k-step), W fragments from global memory in MFMA operand order (snf_split_weights_b3 writes the constant
//            weights once per render as bf16 hi / lo planes laid out [k-step][column tile][lane][8]: a wave's fragment is 1 KB
//            contiguous, the 196 KB of planes stay L2-resident), hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16, fp32 accumulate.
//   phase 3  ReLU and the weighted mean over a ray's 16 samples in the accumulator registers (the 16 rows of a ray are 8 registers
//            in each half-wave: one cross-half exchange), one [ray, 256] row of Hbar per 16 samples -- the linear last layer then
//            runs on rays (sum_k w_k (W h_k) = W (sum_k w_k h_k)), as in the train schedule.
#include "common.hpp"
#include "grid_device.hpp"

namespace snf {

typedef __bf16 fh_bf16x8 __attribute__((ext_vector_type(8)));
typedef float fh_f32x16 __attribute__((ext_vector_type(16)));

constexpr int FH_M = 64;   // samples per workgroup
constexpr int FH_T = 256;  // threads

// W [O][I] fp32 -> bf16 hi / lo planes in B-fragment order of v_mfma_f32_32x32x16_bf16:
//   plane[((s * (O / 32) + t) * 64 + lane) * 8 + e] = W[32 t + (lane & 31)][16 s + 8 (lane >> 5) + e]
__global__ __launch_bounds__(256) void k_split_weights_frag(const float* __restrict__ W, int O, int I, uint16_t* __restrict__ hi,
                                                            uint16_t* __restrict__ lo) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int total = (I / 16) * (O / 32) * 64;
    if (g >= total) return;
    const int lane = g & 63, t = (g >> 6) % (O / 32), s = (g >> 6) / (O / 32);
    const float* __restrict__ src = W + (size_t)(32 * t + (lane & 31)) * I + 16 * s + 8 * (lane >> 5);
    const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
    uint32_t h[4], l[4];
    gd_split2(a.x, a.y, h[0], l[0]);
    gd_split2(a.z, a.w, h[1], l[1]);
    gd_split2(b.x, b.y, h[2], l[2]);
    gd_split2(b.z, b.w, h[3], l[3]);
    *reinterpret_cast<uint4*>(hi + (size_t)g * 8) = make_uint4(h[0], h[1], h[2], h[3]);
    *reinterpret_cast<uint4*>(lo + (size_t)g * 8) = make_uint4(l[0], l[1], l[2], l[3]);
}

template <int NT>  // 32-column accumulator tiles per wave: O = 64 * NT hidden units (4: the heads' 256)
__global__ __launch_bounds__(FH_T) void k_grid_head_fused(const float* __restrict__ u, const float* __restrict__ tabA,
                                                          const float* __restrict__ scA, int LA, const float* __restrict__ tabB,
                                                          const float* __restrict__ scB, int LB, int log2_T,
                                                          const uint16_t* __restrict__ Whi, const uint16_t* __restrict__ Wlo,
                                                          const float* __restrict__ row_weight, float* __restrict__ Hbar) {
    extern __shared__ __attribute__((aligned(16))) uint16_t fh_lds[];
    const int L = LA + LB, I = L * 8, pitch = I + 8;  // (pitch: 400 B rows at I = 192 -- 16-lane groups of a b128 access hit 16 slots)
    uint16_t* Ah = fh_lds;
    uint16_t* Al = fh_lds + FH_M * pitch;
    constexpr int O = 64 * NT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, half = lane >> 5;
    const int n0 = blockIdx.x * FH_M;
    const uint32_t mask = (1u << log2_T) - 1u;
    // ---- phase 1: the tile's features into LDS, split for the 3-product arithmetic
    for (int p = tid; p < FH_M * L; p += FH_T) {
        const int lvl = p >> 6, m = p & (FH_M - 1);
        const bool second = lvl >= LA;
        const int ll = second ? lvl - LA : lvl;
        const float s = second ? scB[ll] : scA[ll];
        const float* __restrict__ slab = (second ? tabB : tabA) + ((size_t)ll << log2_T) * 8;
        const Corners c = corners_of(u, n0 + m, s, mask);
        float4 f0[8], f1[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            f0[k] = *reinterpret_cast<const float4*>(slab + (size_t)c.idx[k] * 8);
            f1[k] = *reinterpret_cast<const float4*>(slab + (size_t)c.idx[k] * 8 + 4);
        }
        const float ox = c.ox, oy = c.oy, oz = c.oz;
        const float mx = 1.f - ox, my = 1.f - oy, mz = 1.f - oz;
        float r[8];
        auto blend = [&](float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
            const float f03 = a0 * ox + a3 * mx;
            const float f12 = a1 * ox + a2 * mx;
            const float f56 = a5 * ox + a6 * mx;
            const float f47 = a4 * ox + a7 * mx;
            const float f0312 = f03 * oy + f12 * my;
            const float f4756 = f47 * oy + f56 * my;
            return f0312 * oz + f4756 * mz;
        };
        r[0] = blend(f0[0].x, f0[1].x, f0[2].x, f0[3].x, f0[4].x, f0[5].x, f0[6].x, f0[7].x);
        r[1] = blend(f0[0].y, f0[1].y, f0[2].y, f0[3].y, f0[4].y, f0[5].y, f0[6].y, f0[7].y);
        r[2] = blend(f0[0].z, f0[1].z, f0[2].z, f0[3].z, f0[4].z, f0[5].z, f0[6].z, f0[7].z);
        r[3] = blend(f0[0].w, f0[1].w, f0[2].w, f0[3].w, f0[4].w, f0[5].w, f0[6].w, f0[7].w);
        r[4] = blend(f1[0].x, f1[1].x, f1[2].x, f1[3].x, f1[4].x, f1[5].x, f1[6].x, f1[7].x);
        r[5] = blend(f1[0].y, f1[1].y, f1[2].y, f1[3].y, f1[4].y, f1[5].y, f1[6].y, f1[7].y);
        r[6] = blend(f1[0].z, f1[1].z, f1[2].z, f1[3].z, f1[4].z, f1[5].z, f1[6].z, f1[7].z);
        r[7] = blend(f1[0].w, f1[1].w, f1[2].w, f1[3].w, f1[4].w, f1[5].w, f1[6].w, f1[7].w);
        uint32_t h[4], l[4];
        gd_split2(r[0], r[1], h[0], l[0]);
        gd_split2(r[2], r[3], h[1], l[1]);
        gd_split2(r[4], r[5], h[2], l[2]);
        gd_split2(r[6], r[7], h[3], l[3]);
        *reinterpret_cast<uint4*>(Ah + m * pitch + lvl * 8) = make_uint4(h[0], h[1], h[2], h[3]);
        *reinterpret_cast<uint4*>(Al + m * pitch + lvl * 8) = make_uint4(l[0], l[1], l[2], l[3]);
    }
    __syncthreads();
    // ---- phase 2: H^pre[64, O] = A[64, I] W^T on the bf16 matrix cores; wave = (row block, column half)
    const int rb = wave >> 1, ch = wave & 1;
    fh_f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const int ksteps = I / 16;
    const uint16_t* __restrict__ arow_h = Ah + (rb * 32 + li) * pitch + half * 8;
    const uint16_t* __restrict__ arow_l = Al + (rb * 32 + li) * pitch + half * 8;
    for (int s = 0; s < ksteps; ++s) {
        const fh_bf16x8 ah = *reinterpret_cast<const fh_bf16x8*>(arow_h + s * 16);
        const fh_bf16x8 al = *reinterpret_cast<const fh_bf16x8*>(arow_l + s * 16);
        const size_t wbase = ((size_t)s * (O / 32) + (size_t)ch * NT) * 64 + lane;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const fh_bf16x8 bh = *reinterpret_cast<const fh_bf16x8*>(Whi + (wbase + (size_t)t * 64) * 8);
            const fh_bf16x8 bl = *reinterpret_cast<const fh_bf16x8*>(Wlo + (wbase + (size_t)t * 64) * 8);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
        }
    }
    // ---- phase 3: ReLU + weighted mean over the 16 samples of a ray, in registers.  Accumulator layout: column = lane & 31,
    // row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5): registers 0..7 hold rows of the block's first ray, 8..15 of its second
    float wq[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) wq[reg] = row_weight[n0 + rb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half];
    const int ray0 = (n0 >> 4) + rb * 2;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = (ch * NT + t) * 32 + li;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            float part = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) part += wq[r * 8 + q] * fmaxf(acc[t][r * 8 + q], 0.f);
            part += __shfl_xor(part, 32, 64);
            if (half == 0) Hbar[(size_t)(ray0 + r) * O + col] = part;
        }
    }
}

}  // namespace snf

using namespace snf;

extern "C" int snf_split_weights_b3(const float* W, int O, int I, void* hi, void* lo, snf_stream_t stream) {
    SNF_REQUIRE(W && hi && lo, "snf_split_weights_b3: null pointer");
    SNF_REQUIRE(O > 0 && I > 0 && O % 32 == 0 && I % 16 == 0 && (((uintptr_t)W | (uintptr_t)hi | (uintptr_t)lo) & 15) == 0,
                "snf_split_weights_b3: W [O=%d, I=%d] needs O %% 32 == 0, I %% 16 == 0 and 16-byte aligned pointers", O, I);
    const int total = (I / 16) * (O / 32) * 64;
    hipLaunchKernelGGL(k_split_weights_frag, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, W, O, I, (uint16_t*)hi,
                       (uint16_t*)lo);
    SNF_LAUNCH_CHECK("snf_split_weights_b3");
    return SNF_OK;
}

extern "C" int snf_grid_head_fused_fwd(const float* u, const float* tableA, const float* scalingsA, int LA, const float* tableB,
                                       const float* scalingsB, int LB, int log2_T, const void* Whi, const void* Wlo, int O,
                                       const float* row_weight, int group, float* Hbar, int N, snf_stream_t stream) {
    SNF_REQUIRE(u && tableA && scalingsA && tableB && scalingsB && Whi && Wlo && row_weight && Hbar,
                "snf_grid_head_fused_fwd: null pointer");
    const int I = (LA + LB) * 8;
    SNF_REQUIRE(LA > 0 && LB > 0 && (LA + LB) % 2 == 0 && I <= 256 && log2_T >= 1 && log2_T <= 26 && (O == 256 || O == 128) &&
                    group == 16 && N > 0 && N % FH_M == 0,
                "snf_grid_head_fused_fwd: needs two F = 8 grids with an even level count (<= 32 levels: got %d + %d), 128 or 256 hidden "
                "units (got %d), groups of 16 samples (got %d) and N %% 64 == 0 (got %d)", LA, LB, O, group, N);
    SNF_REQUIRE((((uintptr_t)tableA | (uintptr_t)tableB | (uintptr_t)Whi | (uintptr_t)Wlo | (uintptr_t)Hbar) & 15) == 0,
                "snf_grid_head_fused_fwd: unaligned pointer");
    const size_t lds = (size_t)2 * FH_M * (I + 8) * sizeof(uint16_t);
    static size_t attr[2] = {0, 0};
    const int which = O == 256 ? 1 : 0;
    if (lds > 48 * 1024 && lds > attr[which]) {
        attr[which] = lds;
        if (which) (void)hipFuncSetAttribute((const void*)k_grid_head_fused<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        else (void)hipFuncSetAttribute((const void*)k_grid_head_fused<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (which)
        hipLaunchKernelGGL(k_grid_head_fused<4>, dim3(N / FH_M), dim3(FH_T), lds, (hipStream_t)stream, u, tableA, scalingsA, LA, tableB,
                           scalingsB, LB, log2_T, (const uint16_t*)Whi, (const uint16_t*)Wlo, row_weight, Hbar);
    else
        hipLaunchKernelGGL(k_grid_head_fused<2>, dim3(N / FH_M), dim3(FH_T), lds, (hipStream_t)stream, u, tableA, scalingsA, LA, tableB,
                           scalingsB, LB, log2_T, (const uint16_t*)Whi, (const uint16_t*)Wlo, row_weight, Hbar);
    SNF_LAUNCH_CHECK("snf_grid_head_fused_fwd");
    return SNF_OK;
}

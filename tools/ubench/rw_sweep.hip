// Sweep of Adam-shaped streaming read-modify-write kernels (4 fp32 arrays read, 4 written) to find the launch shape and
// access form with the highest HBM throughput on MI355X.  build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/rw_sweep tools/ubench/rw_sweep.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <bool NTL, bool NTS> struct IO {
    static __device__ __forceinline__ f4 ld(const f4* p) { return NTL ? __builtin_nontemporal_load(p) : *p; }
    static __device__ __forceinline__ void st(f4* p, f4 v) { if (NTS) __builtin_nontemporal_store(v, p); else *p = v; }
};
__device__ __forceinline__ void upd(f4& P, f4 G, f4& M, f4& V) {
    M = 0.9f * M + 0.1f * G; V = 0.999f * V + 0.001f * G * G;
    f4 d; for (int c = 0; c < 4; c++) d[c] = __builtin_sqrtf(V[c]) + 1e-8f;
    for (int c = 0; c < 4; c++) P[c] -= 1e-3f * M[c] * __builtin_amdgcn_rcpf(d[c]);
}
// MODE 0: grid-stride; MODE 1: block-contiguous chunks (each block owns n4/gridDim consecutive float4s)
template <int U, int MODE, bool NTL, bool NTS, bool ZERO>
__global__ void rw4(f4* __restrict__ p, f4* __restrict__ g, f4* __restrict__ m, f4* __restrict__ v, long n4) {
    using io = IO<NTL, NTS>;
    long start, end, stride;
    if (MODE == 0) { start = blockIdx.x * (long)blockDim.x + threadIdx.x; end = n4; stride = (long)gridDim.x * blockDim.x; }
    else { long per = (n4 + gridDim.x - 1) / gridDim.x; start = blockIdx.x * per + threadIdx.x; end = min(n4, (blockIdx.x + 1) * per); stride = blockDim.x; }
    long i = start;
    for (; i + (U - 1) * stride < end; i += U * stride) {
        f4 P[U], G[U], M[U], V[U];
#pragma unroll
        for (int u = 0; u < U; u++) { long j = i + u * stride; P[u] = io::ld(p + j); G[u] = io::ld(g + j); M[u] = io::ld(m + j); V[u] = io::ld(v + j); }
#pragma unroll
        for (int u = 0; u < U; u++) upd(P[u], G[u], M[u], V[u]);
#pragma unroll
        for (int u = 0; u < U; u++) { long j = i + u * stride; io::st(p + j, P[u]); io::st(m + j, M[u]); io::st(v + j, V[u]); if (ZERO) io::st(g + j, f4{0, 0, 0, 0}); }
    }
    for (; i < end; i += stride) {
        f4 P = io::ld(p + i), G = io::ld(g + i), M = io::ld(m + i), V = io::ld(v + i);
        upd(P, G, M, V);
        io::st(p + i, P); io::st(m + i, M); io::st(v + i, V); if (ZERO) io::st(g + i, f4{0, 0, 0, 0});
    }
}
// m and v interleaved (one 32 B record per float4 of parameters): 3 read streams + 3 written
template <int U, bool ZERO>
__global__ void rw3(f4* __restrict__ p, f4* __restrict__ g, f4* __restrict__ mv, long n4) {
    using io = IO<true, true>;
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += stride) {
        f4 P = io::ld(p + i), G = io::ld(g + i), M = io::ld(mv + 2 * i), V = io::ld(mv + 2 * i + 1);
        upd(P, G, M, V);
        io::st(p + i, P); io::st(mv + 2 * i, M); io::st(mv + 2 * i + 1, V); if (ZERO) io::st(g + i, f4{0, 0, 0, 0});
    }
}

static float* buf[4];
static hipEvent_t e0, e1;
template <typename F> static float timeit(F launch) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    return best;
}
#define RUN(U, MODE, NTL, NTS, ZERO, grid, blk) do { \
    float ms = timeit([&] { rw4<U, MODE, NTL, NTS, ZERO><<<grid, blk>>>((f4*)buf[0], (f4*)buf[1], (f4*)buf[2], (f4*)buf[3], n4); }); \
    double bytes = (ZERO ? 32.0 : 28.0) * n; \
    printf("rw4 U=%d mode=%d ntl=%d nts=%d zero=%d grid=%5d blk=%4d  %.4f ms  %.0f GB/s\n", U, MODE, NTL, NTS, ZERO, grid, blk, ms, bytes / ms * 1e-6); } while (0)

int main(int argc, char** argv) {
    long n = (argc > 1 ? atol(argv[1]) : 100L << 20);
    long n4 = n / 4;
    for (int i = 0; i < 4; i++) { CK(hipMalloc(&buf[i], n * 4)); CK(hipMemset(buf[i], 0, n * 4)); }
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (argc > 2) {   // fine sweep at low occupancy
        for (int g : {256, 512, 768, 1024}) for (int blk : {64, 128, 256, 512}) {
            RUN(1, 0, true, true, true, g, blk); RUN(2, 0, true, true, true, g, blk); RUN(4, 0, true, true, true, g, blk); RUN(8, 0, true, true, true, g, blk);
        }
        for (int g : {256, 512, 1024}) { RUN(1, 1, true, true, true, g, 256); RUN(4, 1, true, true, true, g, 256); }
        return 0;
    }
    int grids[] = {256 * 2, 256 * 4, 256 * 8, 256 * 16, 256 * 32, 256 * 64};
    for (int g : grids) { RUN(2, 0, true, true, true, g, 256); }
    for (int g : grids) { RUN(1, 0, true, true, true, g, 256); }
    for (int g : grids) { RUN(4, 0, true, true, true, g, 256); }
    for (int g : {256, 512, 1024, 2048}) { RUN(2, 0, true, true, true, g, 512); RUN(2, 0, true, true, true, g, 1024); }
    for (int g : grids) { RUN(2, 1, true, true, true, g, 256); }
    RUN(2, 0, false, true, true, 2048, 256); RUN(2, 0, true, false, true, 2048, 256); RUN(2, 0, false, false, true, 2048, 256);
    RUN(2, 0, true, true, false, 2048, 256); RUN(1, 0, true, true, false, 4096, 256);
    {   // one block per 256 float4s, no loop at all
        int g = (int)((n4 + 255) / 256);
        RUN(1, 0, true, true, true, g, 256);
        g = (int)((n4 + 1023) / 1024); RUN(1, 0, true, true, true, g, 1024);
    }
    for (int g : {1024, 2048, 4096, 8192}) {
        float ms = timeit([&] { rw3<1, true><<<g, 256>>>((f4*)buf[0], (f4*)buf[1], (f4*)buf[2], n4 / 2); });
        printf("rw3 (m,v interleaved) grid=%d  %.4f ms  %.0f GB/s\n", g, ms, 32.0 * (n / 2) / ms * 1e-6);
    }
    return 0;
}

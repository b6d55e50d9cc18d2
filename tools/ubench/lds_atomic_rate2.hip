// Micro-benchmark: LDS read-modify-write rates on gfx950, 32- and 64-bit (per-CU lane-ops per clock).
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_atomic_rate2.hip -o build/lds_atomic_rate2 && build/lds_atomic_rate2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
    extern __shared__ float acc[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 16384; i += 1024) acc[i] = 0.f;
    __syncthreads();
    uint32_t x = tid * 2654435761u + blockIdx.x * 40503u + 12345u;
    float v = 1.0f + tid * 1e-6f;
    unsigned long long* a64 = reinterpret_cast<unsigned long long*>(acc);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            x = x * 1664525u + 1013904223u;
            const uint32_t a = (x >> 8) & 16383u;
            if (MODE == 0) atomicAdd(&acc[a], v);
            else if (MODE == 1) atomicAdd(reinterpret_cast<uint32_t*>(acc) + a, 3u);
            else if (MODE == 5) atomicAdd(a64 + (a >> 1), (unsigned long long)x);
            else if (MODE == 7) { atomicAdd(a64 + (a >> 1), (unsigned long long)x); atomicAdd(a64 + ((a >> 1) ^ 1), (unsigned long long)x + 1); }
        }
    }
    __syncthreads();
    float s = 0.f;
    for (int i = tid; i < 16384; i += 1024) s += acc[i];
    if (s == 123.456f) out[0] = s;
}
template <int MODE>
void run(const char* name, int per) {
    float* d; hipMalloc(&d, 4);
    const int iters = 64, blocks = 512;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 65536, 0, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 65536, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double ops = (double)blocks * 1024 * iters * 8 * per;
    printf("%-28s %8.3f ms  %8.2f G lane-ops/s  = %.3f lane-ops/clk/CU (256 CU @2.1GHz)\n", name, ms, ops / ms / 1e6,
           ops / (ms * 1e-3) / 256 / 2.1e9);
}
int main() {
    run<0>("atomicAdd float", 1);
    run<1>("ds_add_u32 random", 1);
    run<5>("ds_add_u64 random", 1);
    run<7>("2x ds_add_u64 adjacent", 2);
    return 0;
}

// Micro-benchmark: throughput of LDS read-modify-write flavours on gfx950 (per-CU lane-ops per clock).
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_atomic_rate.hip -o lds_atomic_rate && ./lds_atomic_rate
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
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            x = x * 1664525u + 1013904223u;
            const uint32_t a = (x >> 8) & 16383u;
            if (MODE == 0) atomicAdd(&acc[a], v);                                   // ds_add_f32
            else if (MODE == 1) atomicAdd(reinterpret_cast<uint32_t*>(acc) + a, 3u);  // ds_add_u32
            else if (MODE == 2) acc[a] = v;                                         // ds_write_b32
            else if (MODE == 3) acc[a] += v;                                        // non-atomic read + write
            else if (MODE == 4) atomicAdd(&acc[(a & ~7u) | j], v);                  // 8 consecutive floats of a row
        }
    }
    __syncthreads();
    float s = 0.f;
    for (int i = tid; i < 16384; i += 1024) s += acc[i];
    if (s == 123.456f) out[0] = s;
}

template <int MODE>
void run(const char* name) {
    float* d; hipMalloc(&d, 4);
    const int iters = 64, blocks = 512;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 65536, 0, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 65536, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double ops = (double)blocks * 1024 * iters * 8;
    printf("%-28s %8.3f ms  %8.2f G lane-ops/s  = %.3f lane-ops/clk/CU (256 CU @2.1GHz)\n", name, ms, ops / ms / 1e6,
           ops / (ms * 1e-3) / 256 / 2.1e9);
}

int main() {
    run<0>("ds_add_f32 random");
    run<1>("ds_add_u32 random");
    run<2>("ds_write_b32 random");
    run<3>("read+add+write (racy)");
    run<4>("ds_add_f32 row of 8");
    return 0;
}

// PMC calibration for the fused Adam kernel's access pattern (MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE must be
// calibrated on a known byte count in one's own access pattern).  Four streaming kernels with trivially known traffic, run
// under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes):
//   copy_plain : 16 B/lane loads, 16 B/lane stores               read N*4,   write N*4
//   copy_nt    : nontemporal 16 B loads / stores (Adam's form)   read N*4,   write N*4
//   rw4_nt     : 4 arrays read + the same 4 written (Adam shape) read N*16,  write N*16
//   fill_nt    : nontemporal stores only                          read 0,     write N*4
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/pmc_calib tools/ubench/pmc_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void copy_plain(const f4* __restrict__ a, f4* __restrict__ b, long n4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void copy_nt(const f4* __restrict__ a, f4* __restrict__ b, long n4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(__builtin_nontemporal_load(a + i), b + i);
}
__global__ void rw4_nt(f4* __restrict__ p, f4* __restrict__ g, f4* __restrict__ m, f4* __restrict__ v, long n4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f4 P = __builtin_nontemporal_load(p + i), G = __builtin_nontemporal_load(g + i);
        f4 M = __builtin_nontemporal_load(m + i), V = __builtin_nontemporal_load(v + i);
        M = 0.9f * M + 0.1f * G; V = 0.999f * V + 0.001f * G * G; P = P - 1e-3f * M;
        __builtin_nontemporal_store(P, p + i); __builtin_nontemporal_store(M, m + i);
        __builtin_nontemporal_store(V, v + i); __builtin_nontemporal_store(f4{0, 0, 0, 0}, g + i);
    }
}
__global__ void fill_nt(f4* __restrict__ b, long n4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(f4{1, 2, 3, 4}, b + i);
}

int main(int argc, char** argv) {
    long n = (argc > 1 ? atol(argv[1]) : 100L << 20);   // floats per array (default 100 Mi = 400 MiB per array, past the 256 MiB L3)
    long n4 = n / 4;
    float* buf[4];
    for (int i = 0; i < 4; i++) { CK(hipMalloc(&buf[i], n * 4)); CK(hipMemset(buf[i], 0, n * 4)); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int grid = 256 * 8, blk = 256;
    const char* names[4] = {"copy_plain", "copy_nt", "rw4_nt", "fill_nt"};
    double bytes[4] = {8.0 * n, 8.0 * n, 32.0 * n, 4.0 * n};
    for (int k = 0; k < 4; k++) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipEventRecord(e0));
            if (k == 0) copy_plain<<<grid, blk>>>((f4*)buf[0], (f4*)buf[1], n4);
            if (k == 1) copy_nt<<<grid, blk>>>((f4*)buf[0], (f4*)buf[1], n4);
            if (k == 2) rw4_nt<<<grid, blk>>>((f4*)buf[0], (f4*)buf[1], (f4*)buf[2], (f4*)buf[3], n4);
            if (k == 3) fill_nt<<<grid, blk>>>((f4*)buf[0], n4);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("%-10s n=%ld bytes=%.0f best_ms=%.4f GB/s=%.1f\n", names[k], n, bytes[k], best, bytes[k] / best * 1e-6);
    }
    return 0;
}

// streams.hip -- HIP streams restricted to a subset of the chip's compute units.
//
// The train step runs three task streams (step_program.py); the reference gets concurrency between its nerfacto branch and its feature
// heads only by accident of the CUDA stream scheduler (samnerf/sam_model.py:226-301 runs them back to back on one stream).  Here the
// schedule may pin a task, or single launches of it, to part of the 256 CUs: a bandwidth-bound table kernel (2 workgroups of 75 KB LDS
// and 128 VGPRs per CU: a CU it occupies takes nothing else) does not need every CU to reach its HBM rate, while the matrix kernels
// of the other tasks need CUs that are not full.  hipExtStreamCreateWithCUMask is the runtime's interface for exactly that.
#include "common.hpp"

using namespace snf;

// The first `n_cus` bits of the mask are set: the kernel driver maps mask bits to physical CUs symmetrically over the XCDs and shader
// engines, so "the first n" is an even n/8 share of every XCD.  n_cus is clamped to [8, device CU count].
extern "C" int snf_stream_create_cu_mask(int n_cus, snf_stream_t* out_stream) {
    SNF_REQUIRE(out_stream, "snf_stream_create_cu_mask: null out pointer");
    int dev = 0, total = 0;
    SNF_REQUIRE(hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&total, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
                    total > 0,
                "snf_stream_create_cu_mask: cannot query the device");
    if (n_cus > total) n_cus = total;
    if (n_cus < 8) n_cus = 8;
    uint32_t mask[16] = {0};  // 512 bits: more CUs than any gfx9 part has
    SNF_REQUIRE(total <= 512, "snf_stream_create_cu_mask: %d CUs exceed the mask", total);
    for (int i = 0; i < n_cus; ++i) mask[i >> 5] |= 1u << (i & 31);
    hipStream_t st = nullptr;
    const hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)((total + 31) / 32), mask);
    SNF_REQUIRE(e == hipSuccess, "snf_stream_create_cu_mask: hipExtStreamCreateWithCUMask failed (%s)", hipGetErrorString(e));
    *out_stream = (snf_stream_t)st;
    return SNF_OK;
}

// priority: 0 = the device's default, negative = higher, positive = lower; clamped to the device's range
// (hipDeviceGetStreamPriorityRange: -1 .. 1 on gfx950).  The dispatcher serves the queues of higher priority first when CU slots free up.
extern "C" int snf_stream_create_priority(int priority, snf_stream_t* out_stream) {
    SNF_REQUIRE(out_stream, "snf_stream_create_priority: null out pointer");
    int least = 0, greatest = 0;
    SNF_REQUIRE(hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess, "snf_stream_create_priority: cannot query the range");
    if (priority > least) priority = least;
    if (priority < greatest) priority = greatest;
    hipStream_t st = nullptr;
    const hipError_t e = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, priority);
    SNF_REQUIRE(e == hipSuccess, "snf_stream_create_priority: hipStreamCreateWithPriority failed (%s)", hipGetErrorString(e));
    *out_stream = (snf_stream_t)st;
    return SNF_OK;
}

extern "C" int snf_stream_destroy(snf_stream_t stream) {
    SNF_REQUIRE(stream, "snf_stream_destroy: null stream");
    const hipError_t e = hipStreamDestroy((hipStream_t)stream);
    SNF_REQUIRE(e == hipSuccess, "snf_stream_destroy: %s", hipGetErrorString(e));
    return SNF_OK;
}

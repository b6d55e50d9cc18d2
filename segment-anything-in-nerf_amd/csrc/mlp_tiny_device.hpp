// mlp_tiny_device.hpp -- the proposal net's 10 -> 16 -> 1 forward for ONE sample on the vector ALU, shared by k_mlp_tiny_fwd (mlp_tiny.hip)
// and the eval render's fused proposal kernel (k_prop_density_fwd, hashgrid.hip): the multiply-adds in the order and with the roundings
// of `a += w * x` under contraction, spelled out with fmaf so that both kernels round alike whatever surrounds the call.
#pragma once
#include "common.hpp"

namespace snf {

// w0 [H][I], w1 [H] (LDS or global); h: the hidden layer after ReLU; returns the output
template <int I, int H>
__device__ __forceinline__ float mt_forward(const float (&x)[I], const float* __restrict__ w0, const float* __restrict__ w1,
                                            float (&h)[H]) {
    float y = 0.f;
#pragma unroll
    for (int j = 0; j < H; ++j) {
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < I; ++i) a = fmaf(w0[j * I + i], x[i], a);
        h[j] = fmaxf(a, 0.f);
        y = fmaf(w1[j], h[j], y);
    }
    return y;
}

}  // namespace snf

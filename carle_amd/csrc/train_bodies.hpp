// train_bodies.hpp -- small device-function bodies of the predictor-update path that more than one translation unit
// launches: as stand-alone kernels (nets_staged.hip / nets_staged_bwd.hpp) and inside the multi-body launches of the
// paired RND2D + AE2D training step at tiled sizes (nets_tiled.hip: rnd_bwd_pair_kernel, rnd_ae_train_tail_kernel).
// Inline device code only: no kernels, no host code.
#pragma once
#include "common.hpp"

namespace carle {

// adds `rows` partial rows of `nparams` floats in a fixed order onto grad[0..nparams): workgroup bidx (1024 threads)
// takes parameters 32 bidx .. 32 bidx + 31, thread = (parameter lane, one of 32 row lanes); deterministic
__device__ __forceinline__ void staged_grad_reduce_body(float (&part)[32][33], const float *__restrict__ partial, int rows,
                                                        float *__restrict__ grad, int nparams, int bidx) {
    const int pl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int p = bidx * 32 + pl;
    float s0 = 0.0f, s1 = 0.0f;
    if (p < nparams) {
        int r = rl;
        for (; r + 32 < rows; r += 64) {
            s0 += partial[(int64_t)r * nparams + p];
            s1 += partial[(int64_t)(r + 32) * nparams + p];
        }
        if (r < rows) s0 += partial[(int64_t)r * nparams + p];
    }
    part[rl][pl] = s0 + s1;
    __syncthreads();
    if (rl == 0 && p < nparams) {
        float t = 0.0f;
#pragma unroll
        for (int i = 0; i < 32; ++i) t += part[i][pl];
        grad[p] += t;
    }
}

static constexpr int kRndK = 16;                      // RND2D output width (rnd_dim, mcl.py:245)

__device__ __forceinline__ float rnd_dz(const float *__restrict__ pred, const float *__restrict__ targ,
                                        int64_t n, int k, float gs) {
    const float p = pred[n * kRndK + k], t = targ[n * kRndK + k];
    return gs * (p - t) * (1.0f - p * p);                 // d/dz of gs/2 (t - tanh z)^2
}

// linear layer: grad[77 + k*D + f] += sum_n dz[n][k] * feat[n][f];  grad[77 + 16 D + k] += sum_n dz[n][k]
__device__ __forceinline__ void rnd_bwd_linear_body(
    const float *__restrict__ ws2, const float *__restrict__ pred, const float *__restrict__ targ,
    float *__restrict__ grad, float gs, int64_t n, int d, int64_t bidx) {
    const int64_t idx = bidx * 256 + threadIdx.x;
    const int64_t nw = (int64_t)kRndK * d;
    if (idx >= nw + kRndK) return;
    const bool bias = idx >= nw;
    const int k = bias ? (int)(idx - nw) : (int)(idx / d);
    const int f = bias ? 0 : (int)(idx % d);
    float acc = 0.0f;
    for (int64_t i = 0; i < n; ++i) {
        const float dz = rnd_dz(pred, targ, i, k, gs);
        acc = fmaf(dz, bias ? 1.0f : ws2[(i * 2) * (int64_t)d + f], acc);
    }
    grad[77 + idx] += acc;
}

// AE2D: bonus of grid g from its squared-error partials (fixed order), reward += scale * bonus (mcl.py:403-405, 504-512)
__device__ __forceinline__ void ae_mse_finish_body(const float *__restrict__ partial, float *__restrict__ bonus,
                                                   float *__restrict__ reward, float reward_scale, int64_t n, int chunks,
                                                   float inv_pixels, int64_t g) {
    if (g >= n) return;
    float s = 0.0f;
    for (int c = 0; c < chunks; ++c) s += partial[g * chunks + c];
    const float b = s * inv_pixels;
    bonus[g] = b;
    if (reward) reward[g] = reward_accumulate(reward[g], reward_scale, b);      // mcl.py:403-405: product, then sum
}

}  // namespace carle

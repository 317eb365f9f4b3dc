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
// (returns the reward it left, 0 without a reward vector or beyond the batch)
__device__ __forceinline__ float ae_mse_finish_body(const float *__restrict__ partial, float *__restrict__ bonus,
                                                    float *__restrict__ reward, float reward_scale, int64_t n, int chunks,
                                                    float inv_pixels, int64_t g) {
    if (g >= n) return 0.0f;
    float s = 0.0f;
    for (int c = 0; c < chunks; ++c) s += partial[g * chunks + c];
    const float b = s * inv_pixels;
    bonus[g] = b;
    if (!reward) return 0.0f;
    const float r = reward_accumulate(reward[g], reward_scale, b);              // mcl.py:403-405: product, then sum
    reward[g] = r;
    return r;
}

// Reward moments of a step whose rewards are final only in a finish kernel (wrapper stacks): every workgroup of that kernel
// leaves (sum, sum of squares) of the rewards it wrote, in fp64, in `partials[bidx]`, and workgroup 0 registers the pairs
// in the status block (words CARLE_ST_MOM_*): the next launch on the status block -- the environment step of the next
// step -- or carle_env_settle merges them into the running triple (settle_moments_wave, ca_cells.hpp), exactly as for
// the one-launch RND2D step.  `part` = 16 pairs of LDS; every thread of the workgroup calls (has = it holds a reward).
__device__ __forceinline__ void reward_moment_partials(double2 (&part)[16], bool has, float value, double *__restrict__ partials,
                                                       int32_t *__restrict__ status, int bidx, int bcount, int64_t n) {
    double sm = has ? (double)value : 0.0, q = sm * sm;
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) { sm += __shfl_xor(sm, k, 64); q += __shfl_xor(q, k, 64); }
    const int wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) part[wv] = make_double2(sm, q);
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int i = 0; i < nw; ++i) { a += part[i].x; b += part[i].y; }
        reinterpret_cast<double2 *>(partials)[bidx] = make_double2(a, b);
        if (bidx == 0) {
            uint4 w5;
            w5.x = (uint32_t)reinterpret_cast<uint64_t>(partials);
            w5.y = (uint32_t)(reinterpret_cast<uint64_t>(partials) >> 32);
            w5.z = (uint32_t)bcount;
            w5.w = (uint32_t)n;
            reinterpret_cast<uint4 *>(status)[5] = w5;                  // CARLE_ST_MOM_PARTIALS / _COUNT / _N
        }
    }
}

}  // namespace carle

// morpho.hip -- MorphoBonus (carle/mcl.py:107-195): pattern-matching reward.
//
//   my_grid  = |universe - action|                                   (mcl.py:172; the grid BEFORE the step, fp32)
//   conv_obs = conv2d(my_grid, target_patterns)                      (mcl.py:173; K patterns of 8x8, no padding)
//   reward  += reward_scale * (max over (k, y, x) + min over (k, y, x))    per instance (mcl.py:177-181)
//
// One workgroup per (instance, band of 16 output rows): the band's 23 grid rows and the K patterns sit in LDS, a thread
// takes output positions of the band, holds the 8x8 window in registers and runs the K patterns over it (the pattern
// weights are LDS broadcasts); per-band maxima / minima go to a partial array, a finish kernel folds them (max and min are
// exact whatever the order) and adds the reward.  Compute-trivial (12 patterns: 768 multiply-adds per output position).
#include "common.hpp"

namespace carle {

static constexpr int kMorphoBand = 16;       // output rows per workgroup
static constexpr int kMorphoP = 8;           // pattern edge (mcl.py:144: dim=8)
static constexpr int kMorphoMaxK = 32;       // patterns (the reference's two files give 12)
static constexpr int kMorphoMaxW = 512;

__global__ __launch_bounds__(256) void morpho_match_kernel(const float *__restrict__ universe, const float *__restrict__ action,
                                                           const float *__restrict__ patterns, int k_patterns, int h, int w,
                                                           int bands, float *__restrict__ partial) {
    __shared__ float rows[(kMorphoBand + kMorphoP - 1) * kMorphoMaxW];
    __shared__ float pat[kMorphoMaxK * kMorphoP * kMorphoP];
    __shared__ float red[2][4];
    const int tid = threadIdx.x;
    const int64_t g = blockIdx.x / bands;
    const int band = (int)(blockIdx.x % bands);
    const int oh = h - kMorphoP + 1, ow = w - kMorphoP + 1;
    const int y0 = band * kMorphoBand, ny = min(kMorphoBand, oh - y0);       // output rows y0 .. y0 + ny
    const int nrows = ny + kMorphoP - 1;
    const float *u = universe + g * (int64_t)h * w, *a = action + g * (int64_t)h * w;
    for (int i = tid; i < nrows * w; i += 256) {
        const int r = i / w, c = i - r * w;
        const int64_t idx = (int64_t)(y0 + r) * w + c;
        rows[r * kMorphoMaxW + c] = fabsf(u[idx] - a[idx]);
    }
    for (int i = tid; i < k_patterns * kMorphoP * kMorphoP; i += 256) pat[i] = patterns[i];
    __syncthreads();
    float vmax = -INFINITY, vmin = INFINITY;
    for (int p = tid; p < ny * ow; p += 256) {
        const int y = p / ow, x = p - y * ow;
        float win[kMorphoP][kMorphoP];
#pragma unroll
        for (int dy = 0; dy < kMorphoP; ++dy)
#pragma unroll
            for (int dx = 0; dx < kMorphoP; ++dx) win[dy][dx] = rows[(y + dy) * kMorphoMaxW + x + dx];
        for (int k = 0; k < k_patterns; ++k) {
            const float *pk = pat + k * kMorphoP * kMorphoP;
            float acc = 0.0f;
#pragma unroll
            for (int dy = 0; dy < kMorphoP; ++dy)
#pragma unroll
                for (int dx = 0; dx < kMorphoP; ++dx) acc = fmaf(win[dy][dx], pk[dy * kMorphoP + dx], acc);
            vmax = fmaxf(vmax, acc);
            vmin = fminf(vmin, acc);
        }
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        vmax = fmaxf(vmax, __shfl_xor(vmax, s, 64));
        vmin = fminf(vmin, __shfl_xor(vmin, s, 64));
    }
    if ((tid & 63) == 0) { red[0][tid >> 6] = vmax; red[1][tid >> 6] = vmin; }
    __syncthreads();
    if (tid == 0) {
        partial[2 * (int64_t)blockIdx.x] = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
        partial[2 * (int64_t)blockIdx.x + 1] = fminf(fminf(red[1][0], red[1][1]), fminf(red[1][2], red[1][3]));
    }
}

// out2[2 i] / [2 i + 1] = max / min of instance i; reward[i] += scale * (max + min)  (sum, product, sum rounded one by one,
// as the three torch operations of mcl.py:181)
__global__ __launch_bounds__(256) void morpho_finish_kernel(const float *__restrict__ partial, int bands, int64_t n,
                                                            float *__restrict__ out2, float *__restrict__ reward, float scale) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float vmax = -INFINITY, vmin = INFINITY;
    for (int b = 0; b < bands; ++b) {
        vmax = fmaxf(vmax, partial[2 * (i * bands + b)]);
        vmin = fminf(vmin, partial[2 * (i * bands + b) + 1]);
    }
    out2[2 * i] = vmax;
    out2[2 * i + 1] = vmin;
    if (reward) reward[i] = reward_accumulate(reward[i], scale, vmax + vmin);   // no contraction into an fma (common.hpp)
}

}  // namespace carle

using namespace carle;

extern "C" int64_t carle_morpho_partial_floats(int64_t n, int h) {
    if (n < 0 || h < kMorphoP) return 0;
    const int64_t bands = (h - kMorphoP + 1 + kMorphoBand - 1) / kMorphoBand;
    return 2 * n * bands;
}

extern "C" int carle_morpho_match_f32(const float *universe, const float *action, const float *patterns, int k_patterns,
                                      int64_t n, int h, int w, float *partial, float *out2, float *reward, float reward_scale,
                                      void *stream) {
    if (!universe || !action || !patterns || !partial || !out2)
        return fail_arg(CARLE_E_NULL, "carle_morpho_match_f32: NULL pointer");
    if (n < 0 || h < kMorphoP || w < kMorphoP || w > kMorphoMaxW || k_patterns < 1 || k_patterns > kMorphoMaxK)
        return fail_arg(CARLE_E_DIMS, "carle_morpho_match_f32: 8 <= H, 8 <= W <= 512 and 1 .. 32 patterns of 8x8 required");
    if (n == 0) return 0;
    const int bands = (h - kMorphoP + 1 + kMorphoBand - 1) / kMorphoBand;
    if (n * bands > 0x7FFFFFFFll) return fail_arg(CARLE_E_DIMS, "carle_morpho_match_f32: batch too large");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(morpho_match_kernel, dim3((unsigned)(n * bands)), dim3(256), 0, s, universe, action, patterns,
                       k_patterns, h, w, bands, partial);
    hipLaunchKernelGGL(morpho_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, partial, bands, n, out2,
                       reward, reward_scale);
    return check_launch("carle_morpho_match_f32");
}

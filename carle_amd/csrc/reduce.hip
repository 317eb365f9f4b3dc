// reduce.hip -- per-instance (weighted) sums over a grid: the reductions behind the hand-coded
// reward wrappers of the reference: live-cell counts (mcl.py:769,832), masked
// sums (CornerBonus, mcl.py:222-223), centre-of-mass numerators (SpeedDetector, mcl.py:772-775) and
// per-instance action sums (ParsimonyBonus, mcl.py:102-103).  One pass over the grid computes the
// plain sum and up to three plane-weighted sums; fp64 accumulation, deterministic order.
#include "common.hpp"

namespace carle {

__global__ __launch_bounds__(256) void instance_sums_kernel(const float *__restrict__ x, int64_t elems,
                                                            const float *__restrict__ w0,
                                                            const float *__restrict__ w1,
                                                            const float *__restrict__ w2,
                                                            float *__restrict__ out) {
    __shared__ double part[4][4];
    const int64_t g = blockIdx.x;
    const float *gx = x + g * elems;
    double s = 0.0, a = 0.0, b = 0.0, c = 0.0;
    for (int64_t i = threadIdx.x; i < elems; i += 256) {
        const double v = (double)gx[i];
        s += v;
        if (w0) a = fma(v, (double)w0[i], a);
        if (w1) b = fma(v, (double)w1[i], b);
        if (w2) c = fma(v, (double)w2[i], c);
    }
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) {
        s += __shfl_xor(s, k, 64); a += __shfl_xor(a, k, 64);
        b += __shfl_xor(b, k, 64); c += __shfl_xor(c, k, 64);
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { part[0][wv] = s; part[1][wv] = a; part[2][wv] = b; part[3][wv] = c; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const double t = (part[threadIdx.x][0] + part[threadIdx.x][1]) + (part[threadIdx.x][2] + part[threadIdx.x][3]);
        out[g * 4 + threadIdx.x] = (float)t;
    }
}

}  // namespace carle

using namespace carle;

extern "C" int carle_instance_sums_f32(const float *x, int64_t n, int64_t elems, const float *w0,
                                       const float *w1, const float *w2, float *out4, void *stream) {
    if (!x || !out4) return fail_arg(CARLE_E_NULL, "carle_instance_sums_f32: NULL pointer");
    if (n < 0 || elems < 0) return fail_arg(CARLE_E_DIMS, "carle_instance_sums_f32: negative size");
    if (n == 0) return 0;
    if (n > 0x7FFFFFFFll) return fail_arg(CARLE_E_DIMS, "carle_instance_sums_f32: batch too large");
    hipLaunchKernelGGL(instance_sums_kernel, dim3((unsigned)n), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, elems, w0, w1, w2, out4);
    return check_launch("carle_instance_sums_f32");
}

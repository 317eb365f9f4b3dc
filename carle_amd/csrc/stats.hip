// stats.hip -- reward statistics for the batch-sharded multi-GPU run.
//
// The reference has no reward normalisation; the north star asks for running reward statistics
// shared across the 8 batch shards.  Each rank reduces its local (N,1) reward vector to the
// moment triple (count, mean, M2) with the kernel below (fp64, deterministic tree), the triples
// are all-gathered over RCCL (24 B per rank) and Chan-merged identically on every rank
// (carle_amd/dist.py).  The same triple also yields the global scalars the reference's harness
// computes per step: reward.sum() (train_mcl.py:68-69, eval.py:69-70).
#include "common.hpp"

namespace carle {

// out[0] = n, out[1] = mean, out[2] = M2 = sum (x - mean)^2, all fp64.  One 256-thread block.
__global__ __launch_bounds__(256) void batch_moments_kernel(const float *__restrict__ x, int64_t n,
                                                            double *__restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += (double)x[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    const double mean = n > 0 ? red[0] / (double)n : 0.0;
    __syncthreads();
    double q = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const double d = (double)x[i] - mean;
        q += d * d;
    }
    red[threadIdx.x] = q;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = (double)n;
        out[1] = mean;
        out[2] = red[0];
    }
}

// y = (x - mean) / sqrt(var + eps) with moments[0..2] = (count, mean, M2) of the running stats
__global__ __launch_bounds__(256) void normalize_kernel(const float *__restrict__ x,
                                                        float *__restrict__ y, int64_t n,
                                                        const double *__restrict__ moments,
                                                        double eps) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double cnt = moments[0], mean = moments[1];
    const double var = cnt > 1.0 ? moments[2] / cnt : 1.0;
    y[i] = (float)(((double)x[i] - mean) / sqrt(var + eps));
}

}  // namespace carle

using namespace carle;

extern "C" int carle_batch_moments_f32(const float *x, int64_t n, double *out3, void *stream) {
    if (!x || !out3) return fail_arg(CARLE_E_NULL, "carle_batch_moments_f32: NULL pointer");
    if (n < 0) return fail_arg(CARLE_E_DIMS, "carle_batch_moments_f32: negative n");
    hipLaunchKernelGGL(batch_moments_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream),
                       x, n, out3);
    return check_launch("carle_batch_moments_f32");
}

extern "C" int carle_normalize_f32(const float *x, float *y, int64_t n, const double *moments3,
                                   double eps, void *stream) {
    if (!x || !y || !moments3) return fail_arg(CARLE_E_NULL, "carle_normalize_f32: NULL pointer");
    if (n < 0) return fail_arg(CARLE_E_DIMS, "carle_normalize_f32: negative n");
    if (n == 0) return 0;
    hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, y, n, moments3, eps);
    return check_launch("carle_normalize_f32");
}

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

// out[0] = n, out[1] = mean, out[2] = M2 = sum (x - mean)^2, all fp64.  One 256-thread block,
// one pass: fp64 sums of x and x^2 (exact enough for fp32 inputs: 2^-53 relative per add),
// wave-level shuffle reduction + one LDS exchange; deterministic order.
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
    return v;
}

__global__ __launch_bounds__(1024) void batch_moments_kernel(const float *__restrict__ x, int64_t n,
                                                             double *__restrict__ out,
                                                             double *__restrict__ acc) {
    __shared__ double part[2][16];
    double s = 0.0, q = 0.0;
    // 16-byte loads, all issued before the first use (latency-bound kernel: one round trip)
    const int64_t n4 = n >> 2;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    for (int64_t i = threadIdx.x; i < n4; i += 1024) {
        const float4 v = x4[i];
        s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
        q += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
    }
    for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 1024) {
        const double v = (double)x[i];
        s += v;
        q = fma(v, v, q);
    }
    s = wave_sum(s);
    q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) {
        part[0][threadIdx.x >> 6] = s;
        part[1][threadIdx.x >> 6] = q;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double sum = 0.0, sq = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { sum += part[0][k]; sq += part[1][k]; }
        const double mean = n > 0 ? sum / (double)n : 0.0;
        double m2 = sq - sum * mean;
        if (m2 < 0.0) m2 = 0.0;
        out[0] = (double)n;
        out[1] = mean;
        out[2] = m2;
        if (acc) {                                  // Chan merge of the batch into the running triple
            const double na = acc[0], ma = acc[1], qa = acc[2], nb = (double)n, tot = na + nb;
            if (tot > 0.0) {
                const double delta = mean - ma;
                acc[0] = tot;
                acc[1] = ma + delta * nb / tot;
                acc[2] = qa + m2 + delta * delta * na * nb / tot;
            }
        }
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

// Folds the all-gathered per-rank triples (rank order) into `last` and `last` into the running
// `state`; clears `pending` (the local accumulator that was exchanged).  One thread: 3*W doubles.
__device__ __forceinline__ void chan(double &n, double &mean, double &m2, double nb, double mb,
                                     double qb) {
    const double tot = n + nb;
    if (tot > 0.0) {
        const double delta = mb - mean;
        mean += delta * nb / tot;
        m2 += qb + delta * delta * n * nb / tot;
        n = tot;
    }
}

__global__ void merge_moments_kernel(const double *__restrict__ gathered, int world,
                                     double *__restrict__ state, double *__restrict__ last,
                                     double *__restrict__ pending) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double n = 0.0, mean = 0.0, m2 = 0.0;
    for (int r = 0; r < world; ++r) chan(n, mean, m2, gathered[3 * r], gathered[3 * r + 1], gathered[3 * r + 2]);
    last[0] = n; last[1] = mean; last[2] = m2;
    double sn = state[0], sm = state[1], sq = state[2];
    chan(sn, sm, sq, n, mean, m2);
    state[0] = sn; state[1] = sm; state[2] = sq;
    if (pending) { pending[0] = 0.0; pending[1] = 0.0; pending[2] = 0.0; }
}

}  // namespace carle

using namespace carle;

extern "C" int carle_batch_moments_f32(const float *x, int64_t n, double *out3, double *acc3,
                                       void *stream) {
    if (!x || !out3) return fail_arg(CARLE_E_NULL, "carle_batch_moments_f32: NULL pointer");
    if (!aligned16(x)) return fail_arg(CARLE_E_ALIGN, "carle_batch_moments_f32: x must be 16-byte aligned");
    if (n < 0) return fail_arg(CARLE_E_DIMS, "carle_batch_moments_f32: negative n");
    hipLaunchKernelGGL(batch_moments_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream),
                       x, n, out3, acc3);
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

extern "C" int carle_merge_moments(const double *gathered, int world, double *state3, double *last3,
                                   double *pending3, void *stream) {
    if (!gathered || !state3 || !last3) return fail_arg(CARLE_E_NULL, "carle_merge_moments: NULL pointer");
    if (world < 1) return fail_arg(CARLE_E_DIMS, "carle_merge_moments: world < 1");
    hipLaunchKernelGGL(merge_moments_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream),
                       gathered, world, state3, last3, pending3);
    return check_launch("carle_merge_moments");
}

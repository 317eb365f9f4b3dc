// optim.hip -- the predictor update of the reward wrappers (RND2D.update_predictor / AE2D.update_predictor,
// carle/mcl.py:333-341, 498-502): optimizer.step() of torch.optim.Adam on the gradient the training kernels accumulated,
// in ONE launch.  The reference's optimiser is Adam(lr = 6e-2) with torch's defaults (betas 0.9 / 0.999, eps 1e-8, no
// weight decay, no amsgrad; mcl.py:243,278-279); the arithmetic below is torch's (_single_tensor_adam / _multi_tensor_adam,
// identical element-wise), operation by operation in fp32, with the step-dependent scalars formed by the host in double
// precision exactly as torch forms them:
//     exp_avg     <- exp_avg + (1 - beta1) * (grad - exp_avg)                      lerp_
//     exp_avg_sq  <- exp_avg_sq * beta2;  exp_avg_sq + (1 - beta2) * (grad * grad) mul_, addcmul_
//     denom       <- sqrt(exp_avg_sq) / sqrt(1 - beta2^t) + eps                    sqrt, div_, add_
//     param       <- param + (-lr / (1 - beta1^t)) * (exp_avg / denom)             addcdiv_
// The parameters stay the nn.Sequential's own tensors (state_dicts, evaluation/*.pt): up to kMaxAdamParams of them are
// updated in place through a pointer table, and the flat blob the bonus kernels read is refreshed in the same pass.
#include "common.hpp"

namespace carle {

struct AdamParams {
    float *ptr[CARLE_ADAM_MAX_PARAMS];
    int64_t end[CARLE_ADAM_MAX_PARAMS];      // exclusive end offset of parameter j inside the flat vectors
    int n;
};

__global__ __launch_bounds__(256) void adam_update_kernel(AdamParams pp, int64_t total, float *__restrict__ blob,
                                                          float *__restrict__ grad, float *__restrict__ exp_avg,
                                                          float *__restrict__ exp_avg_sq, float w1, float beta2, float w2,
                                                          float bc2_sqrt, float eps, float neg_step_size, int zero_grad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int j = 0;
    while (j < pp.n - 1 && i >= pp.end[j]) ++j;
    const int64_t begin = j ? pp.end[j - 1] : 0;
    float *p = pp.ptr[j] + (i - begin);
    const float g = grad[i];
    float m = exp_avg[i], v = exp_avg_sq[i];
    m = m + w1 * (g - m);
    v = v * beta2;
    v = v + w2 * (g * g);
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    const float x = *p + neg_step_size * (m / denom);
    exp_avg[i] = m;
    exp_avg_sq[i] = v;
    *p = x;
    if (blob) blob[i] = x;
    if (zero_grad) grad[i] = 0.0f;
}

}  // namespace carle

using namespace carle;

extern "C" int carle_adam_update_f32(float *const *param_ptrs, const int64_t *param_counts, int n_params, float *blob,
                                     float *grad, float *exp_avg, float *exp_avg_sq, double lr, double beta1, double beta2,
                                     double eps, int64_t step, int zero_grad, void *stream) {
    if (!param_ptrs || !param_counts || !grad || !exp_avg || !exp_avg_sq)
        return fail_arg(CARLE_E_NULL, "carle_adam_update_f32: NULL pointer");
    if (n_params < 1 || n_params > CARLE_ADAM_MAX_PARAMS)
        return fail_arg(CARLE_E_DIMS, "carle_adam_update_f32: 1 .. CARLE_ADAM_MAX_PARAMS parameter tensors");
    if (step < 1) return fail_arg(CARLE_E_DIMS, "carle_adam_update_f32: step counts from 1 (the update being made)");
    AdamParams pp;
    int64_t total = 0;
    for (int j = 0; j < n_params; ++j) {
        if (!param_ptrs[j] || param_counts[j] < 1)
            return fail_arg(CARLE_E_NULL, "carle_adam_update_f32: NULL / empty parameter tensor");
        total += param_counts[j];
        pp.ptr[j] = param_ptrs[j];
        pp.end[j] = total;
    }
    for (int j = n_params; j < CARLE_ADAM_MAX_PARAMS; ++j) { pp.ptr[j] = nullptr; pp.end[j] = total; }
    pp.n = n_params;
    // torch/optim/adam.py: bias_correction1 = 1 - beta1 ** step; bias_correction2 = 1 - beta2 ** step;
    // step_size = lr / bias_correction1; bias_correction2_sqrt = bias_correction2 ** 0.5 -- Python floats (double),
    // handed to the element-wise kernels as fp32 scalars
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    const float w1 = (float)(1.0 - beta1), b2 = (float)beta2, w2 = (float)(1.0 - beta2);
    const float bc2_sqrt = (float)sqrt(bc2), neg_step = (float)(-(lr / bc1));
    const unsigned blocks = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(adam_update_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), pp, total, blob,
                       grad, exp_avg, exp_avg_sq, w1, b2, w2, bc2_sqrt, (float)eps, neg_step, zero_grad ? 1 : 0);
    return check_launch("carle_adam_update_f32");
}

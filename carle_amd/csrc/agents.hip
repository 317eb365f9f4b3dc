// agents.hip -- device-side action generation.
//
// The reference's RandomAgent draws (N,1,AW,AH) uniforms with torch.rand ON THE CPU and the
// wrappers copy the action to the device every step (carle/agents.py:35-42, carle/mcl.py:400):
// 64 MiB of PCIe traffic per step at N=4096.  Here the toggles are generated where they are used:
// a stand-alone Bernoulli kernel (this file) and a fused variant inside the 64x64 step kernel
// (ca_step_cells.hip) that never materialises the action at all.
#include "common.hpp"
#include "philox.hpp"

namespace carle {

__global__ __launch_bounds__(256) void bernoulli_f32_kernel(float4 *__restrict__ out, int64_t quads,
                                                            float p, uint64_t seed, uint64_t offset) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= quads) return;
    const uint32_t nib = bernoulli_nibble((uint64_t)q, seed, offset, p);
    float4 o;
    o.x = (nib & 1u) ? 1.0f : 0.0f;
    o.y = (nib & 2u) ? 1.0f : 0.0f;
    o.z = (nib & 4u) ? 1.0f : 0.0f;
    o.w = (nib & 8u) ? 1.0f : 0.0f;
    out[q] = o;
}

}  // namespace carle

using namespace carle;

extern "C" int carle_bernoulli_f32(float *out, int64_t count, float p, uint64_t seed, uint64_t offset,
                                   void *stream) {
    if (!out) return fail_arg(CARLE_E_NULL, "carle_bernoulli_f32: out is NULL");
    if (count < 0 || (count % 4) != 0)
        return fail_arg(CARLE_E_DIMS, "carle_bernoulli_f32: count must be a non-negative multiple of 4");
    if (!aligned16(out)) return fail_arg(CARLE_E_ALIGN, "carle_bernoulli_f32: out must be 16-byte aligned");
    if (count == 0) return 0;
    const int64_t quads = count / 4, blocks = (quads + 255) / 256;
    if (blocks > 0x7FFFFFFFll) return fail_arg(CARLE_E_DIMS, "carle_bernoulli_f32: too many elements");
    hipLaunchKernelGGL(bernoulli_f32_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), reinterpret_cast<float4 *>(out), quads, p, seed,
                       offset);
    return check_launch("carle_bernoulli_f32");
}

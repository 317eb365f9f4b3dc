// abi.hip -- library-level C-ABI entry points: version, error text, and the step epilogue
// (master reset + counters, carle/env.py:208-216,142-145,200,230).
#include "common.hpp"

namespace carle {

std::string &last_error() {
    thread_local std::string msg;
    return msg;
}

// Applies the whole-batch master reset decided by the flags the step kernel published, zeroes
// the reward vector, maintains the counters and recycles the other flag slot.
__global__ __launch_bounds__(256) void ca_finish_kernel(uint4 *__restrict__ out, int64_t n16,
                                                        unsigned char *__restrict__ tail,
                                                        int tail_bytes, float *__restrict__ reward,
                                                        int64_t n, int32_t *__restrict__ st,
                                                        int slot, int has_action) {
    const bool reset = has_action && st[CARLE_ST_NOT_ALL_ONES(slot)] == 0;
    const bool any = has_action && st[CARLE_ST_ANY_NONZERO(slot)] != 0;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    if (reset) {
        const uint4 z = {0u, 0u, 0u, 0u};
        for (int64_t i = tid; i < n16; i += nthreads) out[i] = z;
        if (tid < tail_bytes) tail[tid] = 0;
    }
    if (reward)
        for (int64_t i = tid; i < n; i += nthreads) reward[i] = 0.0f;
    if (tid == 0) {
        int64_t *step_number = reinterpret_cast<int64_t *>(st + CARLE_ST_STEP_NUMBER);
        int64_t *since = reinterpret_cast<int64_t *>(st + CARLE_ST_STEPS_SINCE_ACTION);
        int64_t *resets = reinterpret_cast<int64_t *>(st + CARLE_ST_RESETS);
        if (!any) *since += 1;                      // env.py:199-200
        if (reset) {                                // env.py:216 -> reset(): env.py:142-145
            *step_number = 0;
            *since = 0;
            *resets += 1;
        } else {
            *step_number += 1;                      // env.py:230
        }
        st[CARLE_ST_LAST_RESET] = reset ? 1 : 0;
        st[CARLE_ST_ANY_NONZERO(slot ^ 1)] = 0;     // the next step publishes into the other slot
        st[CARLE_ST_NOT_ALL_ONES(slot ^ 1)] = 0;
    }
}

}  // namespace carle

using namespace carle;

extern "C" int carle_abi_version(void) { return CARLE_ABI_VERSION; }

extern "C" const char *carle_last_error(void) { return last_error().c_str(); }

extern "C" const char *carle_build_arch(void) {
#ifdef CARLE_BUILD_ARCH
    return CARLE_BUILD_ARCH;
#else
    return "unknown";
#endif
}

extern "C" int carle_ca_finish(void *u_out, int64_t u_out_bytes, float *reward, int64_t n,
                               int32_t *status, int slot, int has_action, void *stream) {
    if (!status) return fail_arg(CARLE_E_NULL, "carle_ca_finish: status is NULL");
    if (!u_out && u_out_bytes > 0) return fail_arg(CARLE_E_NULL, "carle_ca_finish: u_out is NULL");
    if (slot != 0 && slot != 1) return fail_arg(CARLE_E_DIMS, "carle_ca_finish: slot must be 0 or 1");
    if (u_out_bytes < 0 || n < 0) return fail_arg(CARLE_E_DIMS, "carle_ca_finish: negative size");
    if (!aligned16(u_out) || !aligned16(status))
        return fail_arg(CARLE_E_ALIGN, "carle_ca_finish: pointers must be 16-byte aligned");
    const int64_t n16 = u_out_bytes / 16;
    const int tail = (int)(u_out_bytes % 16);
    // small fixed grid: in the common (no reset) case every block exits after one flag load
    hipLaunchKernelGGL(ca_finish_kernel, dim3(32), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<uint4 *>(u_out), n16,
                       reinterpret_cast<unsigned char *>(u_out) + n16 * 16, tail, reward, n, status,
                       slot, (n > 0 && has_action) ? 1 : 0);
    return check_launch("carle_ca_finish");
}

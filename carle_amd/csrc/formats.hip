// formats.hip -- conversions between the grid formats (CARLE_FMT_U8 <-> CARLE_FMT_B1) and the
// stand-alone whole-batch action predicates of CARLE.step (carle/env.py:191, 208).
// (fp32 <-> b1 lives beside the packed step kernel in ca_step_b1.hip.)
#include "common.hpp"

namespace carle {

// one wave packs 64 words: iteration i loads 64 consecutive bytes (one word) as one byte per
// lane, the ballot of "non-zero" is the packed word, kept by lane i; one coalesced 512-B store
__global__ __launch_bounds__(64) void pack_u8_b1_kernel(const uint8_t *__restrict__ src,
                                                        uint64_t *__restrict__ dst, int64_t words,
                                                        int32_t *__restrict__ status, int slot) {
    const int lane = threadIdx.x;
    const int64_t w0 = (int64_t)blockIdx.x * 64;
    uint64_t mine = 0ull;
    uint32_t all_and = 0xFFu, all_or = 0u;
    const int cnt = (int)((words - w0) < 64 ? (words - w0) : 64);
    for (int i = 0; i < cnt; ++i) {
        const uint32_t v = src[(w0 + i) * 64 + lane];
        all_and &= v;
        all_or |= v;
        const uint64_t b = __ballot(v != 0u);
        if (lane == i) mine = b;
    }
    if (lane < cnt) dst[w0 + lane] = mine;
    if (status != nullptr) {
        const bool any = __ballot(all_or != 0u) != 0ull;
        const bool notone = __ballot(all_and != 1u || all_or != 1u) != 0ull;
        if (lane == 0) {
            if (any) status[CARLE_ST_ANY_NONZERO(slot)] = 1;
            if (notone) status[CARLE_ST_NOT_ALL_ONES(slot)] = 1;
        }
    }
}

// one thread expands 16 cells (a quarter word) into 16 bytes
__global__ __launch_bounds__(256) void unpack_b1_u8_kernel(const uint64_t *__restrict__ src,
                                                           uint4 *__restrict__ dst, int64_t chunks) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= chunks) return;
    const uint32_t bits = (uint32_t)(src[i >> 2] >> (16 * (i & 3))) & 0xFFFFu;
    auto four = [](uint32_t nib) { return ((nib & 0xFu) * 0x00204081u) & 0x01010101u; };   // bit j -> byte j
    uint4 o;
    o.x = four(bits);
    o.y = four(bits >> 4);
    o.z = four(bits >> 8);
    o.w = four(bits >> 12);
    dst[i] = o;
}

// whole-batch predicates of an fp32 action, grid-stride; flags are only ever SET (benign races)
__global__ __launch_bounds__(256) void action_flags_f32_kernel(const float *__restrict__ a, int64_t count,
                                                               int32_t *__restrict__ status, int slot) {
    uint32_t all_and = 0xFFFFFFFFu, all_or = 0u;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const uint32_t v = __float_as_uint(a[i]);
        all_and &= v;
        all_or |= v;
    }
    const bool seen = !(all_and == 0xFFFFFFFFu && all_or == 0u);
    const bool any = __ballot((all_or & 0x7FFFFFFFu) != 0u) != 0ull;
    const bool notone = __ballot(seen && (all_and != 0x3F800000u || all_or != 0x3F800000u)) != 0ull;
    if ((threadIdx.x & 63) == 0) {
        if (any) status[CARLE_ST_ANY_NONZERO(slot)] = 1;
        if (notone) status[CARLE_ST_NOT_ALL_ONES(slot)] = 1;
    }
}

}  // namespace carle

using namespace carle;

extern "C" int carle_pack_u8_b1(const uint8_t *src, uint64_t *dst, int64_t n, int h, int w,
                                int32_t *status, int slot, void *stream) {
    if (!src || !dst) return fail_arg(CARLE_E_NULL, "carle_pack_u8_b1: NULL pointer");
    if (n < 0 || h < 1 || w < 64 || (w % 64) != 0)
        return fail_arg(CARLE_E_DIMS, "carle_pack_u8_b1: W must be a positive multiple of 64");
    if (slot != 0 && slot != 1) return fail_arg(CARLE_E_DIMS, "carle_pack_u8_b1: slot must be 0 or 1");
    const int64_t words = n * (int64_t)h * (w / 64);
    if (words == 0) return 0;
    const int64_t blocks = (words + 63) / 64;
    if (blocks > 0x7FFFFFFFll) return fail_arg(CARLE_E_DIMS, "carle_pack_u8_b1: batch too large");
    hipLaunchKernelGGL(pack_u8_b1_kernel, dim3((unsigned)blocks), dim3(64), 0,
                       static_cast<hipStream_t>(stream), src, dst, words, status, slot);
    return check_launch("carle_pack_u8_b1");
}

extern "C" int carle_unpack_b1_u8(const uint64_t *src, uint8_t *dst, int64_t n, int h, int w,
                                  void *stream) {
    if (!src || !dst) return fail_arg(CARLE_E_NULL, "carle_unpack_b1_u8: NULL pointer");
    if (n < 0 || h < 1 || w < 64 || (w % 64) != 0)
        return fail_arg(CARLE_E_DIMS, "carle_unpack_b1_u8: W must be a positive multiple of 64");
    if (!aligned16(dst)) return fail_arg(CARLE_E_ALIGN, "carle_unpack_b1_u8: dst must be 16-byte aligned");
    const int64_t chunks = n * (int64_t)h * (w / 16);
    if (chunks == 0) return 0;
    const int64_t blocks = (chunks + 255) / 256;
    if (blocks > 0x7FFFFFFFll) return fail_arg(CARLE_E_DIMS, "carle_unpack_b1_u8: batch too large");
    hipLaunchKernelGGL(unpack_b1_u8_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), src, reinterpret_cast<uint4 *>(dst), chunks);
    return check_launch("carle_unpack_b1_u8");
}

extern "C" int carle_action_flags_f32(const float *action, int64_t count, int32_t *status, int slot,
                                      void *stream) {
    if (!status) return fail_arg(CARLE_E_NULL, "carle_action_flags_f32: status is NULL");
    if (slot != 0 && slot != 1) return fail_arg(CARLE_E_DIMS, "carle_action_flags_f32: slot must be 0 or 1");
    if (count < 0) return fail_arg(CARLE_E_DIMS, "carle_action_flags_f32: negative count");
    if (count == 0) return 0;
    if (!action) return fail_arg(CARLE_E_NULL, "carle_action_flags_f32: action is NULL");
    int64_t blocks = (count + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(action_flags_f32_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), action, count, status, slot);
    return check_launch("carle_action_flags_f32");
}

// philox.hpp -- Philox4x32-10 counter-based RNG (Salmon et al., SC'11) for the device-side agents.
// Counter layout shared by the stand-alone generator (agents.hip) and the fused random-toggle
// step kernel: quad q of the flattened action tensor (elements 4q..4q+3) uses counter
// (q_lo, q_hi, offset_lo, offset_hi) and key (seed_lo, seed_hi); element j toggles iff
// (word_j >> 8) * 2^-24 <= p.  Restated in numpy in oracle/philox.py.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace carle {

__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint32_t k0, uint32_t k1) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x;
        const uint32_t hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
        c = make_uint4(hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0);
        k0 += W0;
        k1 += W1;
    }
    return c;
}

// 4 Bernoulli(p) toggles for action quad q; bit j of the result = element 4q+j toggles
__device__ __forceinline__ uint32_t bernoulli_nibble(uint64_t q, uint64_t seed, uint64_t offset, float p) {
    const uint4 x = philox4x32_10(make_uint4((uint32_t)q, (uint32_t)(q >> 32), (uint32_t)offset,
                                             (uint32_t)(offset >> 32)),
                                  (uint32_t)seed, (uint32_t)(seed >> 32));
    constexpr float kScale = 1.0f / 16777216.0f;
    return ((float)(x.x >> 8) * kScale <= p ? 1u : 0u) | ((float)(x.y >> 8) * kScale <= p ? 2u : 0u) |
           ((float)(x.z >> 8) * kScale <= p ? 4u : 0u) | ((float)(x.w >> 8) * kScale <= p ? 8u : 0u);
}

}  // namespace carle

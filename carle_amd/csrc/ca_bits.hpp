// ca_bits.hpp -- bit-sliced Life-like rule evaluation shared by every CA kernel.
//
// All functions work on machine words in which every BIT is an independent cell, whatever the
// kernel's cell->bit layout is.  The caller supplies, per word of cells:
//   a0,a1  the 2-bit horizontal triple sum (west+centre+east) of the row ABOVE
//   b0,b1  the same for the row BELOW
//   p0,p1  the 2-bit pair sum (west+east) of the cell's own row
//   c      the cell itself
// and gets the next state for an arbitrary 18-bit birth/survive rule.  This is the arithmetic
// of carle/env.py:219-227 (3x3 Moore count, `elem == count` membership, dead/alive blend).
//
// gfx950 has V_BITOP3_B32: any boolean function of three 32-bit operands in ONE VALU op, selected
// by an 8-bit truth table.  Everything below is written in terms of bop3<TT>() so that a full
// adder is two instructions (xor3 + majority) and a rule term is two (equality mask + merge).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace carle {

// truth tables: evaluate the boolean function on these three constants
static constexpr uint32_t kA = 0xF0, kB = 0xCC, kC = 0xAA;
static constexpr uint32_t kXor3 = (kA ^ kB ^ kC) & 0xFF;
static constexpr uint32_t kMaj = ((kA & kB) | (kA & kC) | (kB & kC)) & 0xFF;
static constexpr uint32_t kAnd3 = (kA & kB & kC) & 0xFF;
static constexpr uint32_t kAxorBC = (kA ^ (kB & kC)) & 0xFF;      // a ^ (b & c)
static constexpr uint32_t kMux = ((kA & kB) | (~kA & kC)) & 0xFF;  // a ? b : c
static constexpr uint32_t kOrAnd = (kA | (kB & kC)) & 0xFF;        // a | (b & c)
static constexpr uint32_t kOrAndN = (kA | (kB & ~kC)) & 0xFF;      // a | (b & ~c)

template <uint32_t TT>
__device__ __forceinline__ uint32_t bop3(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, TT);
}
template <uint32_t TT>
__device__ __forceinline__ uint64_t bop3(uint64_t a, uint64_t b, uint64_t c) {
    const uint32_t lo = __builtin_amdgcn_bitop3_b32((uint32_t)a, (uint32_t)b, (uint32_t)c, TT);
    const uint32_t hi = __builtin_amdgcn_bitop3_b32((uint32_t)(a >> 32), (uint32_t)(b >> 32),
                                                    (uint32_t)(c >> 32), TT);
    return ((uint64_t)hi << 32) | lo;
}

// rows above / below when lane = row: DPP wave rotates / shifts by one lane (a VALU move, no LDS)
__device__ __forceinline__ uint64_t dpp64_ror1(uint64_t x) {      // lane i <- lane i-1 (0 <- 63)
    const uint32_t lo = __builtin_amdgcn_update_dpp(0u, (uint32_t)x, 0x13C, 0xF, 0xF, false);
    const uint32_t hi = __builtin_amdgcn_update_dpp(0u, (uint32_t)(x >> 32), 0x13C, 0xF, 0xF, false);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t dpp64_rol1(uint64_t x) {      // lane i <- lane i+1 (63 <- 0)
    const uint32_t lo = __builtin_amdgcn_update_dpp(0u, (uint32_t)x, 0x134, 0xF, 0xF, false);
    const uint32_t hi = __builtin_amdgcn_update_dpp(0u, (uint32_t)(x >> 32), 0x134, 0xF, 0xF, false);
    return ((uint64_t)hi << 32) | lo;
}

// lane i <- lane i-1, lane 0 keeps `old` (wave_shr:1); lane i <- lane i+1, lane 63 keeps `old`
__device__ __forceinline__ uint64_t dpp64_shr1(uint64_t x, uint64_t old) {
    const uint32_t lo = __builtin_amdgcn_update_dpp((uint32_t)old, (uint32_t)x, 0x138, 0xF, 0xF, false);
    const uint32_t hi = __builtin_amdgcn_update_dpp((uint32_t)(old >> 32), (uint32_t)(x >> 32), 0x138, 0xF, 0xF, false);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t dpp64_shl1(uint64_t x, uint64_t old) {
    const uint32_t lo = __builtin_amdgcn_update_dpp((uint32_t)old, (uint32_t)x, 0x130, 0xF, 0xF, false);
    const uint32_t hi = __builtin_amdgcn_update_dpp((uint32_t)(old >> 32), (uint32_t)(x >> 32), 0x130, 0xF, 0xF, false);
    return ((uint64_t)hi << 32) | lo;
}

template <typename T>
struct Count4 { T s0, s1, s2, s3; };   // neighbour count 0..8, one bit-plane per binary digit

// horizontal sums of one row from its west-shifted / centre / east-shifted words   (4 ops)
template <typename T>
__device__ __forceinline__ void row_sums(T wst, T c, T est, T &h0, T &h1, T &p0, T &p1) {
    p0 = wst ^ est;
    p1 = wst & est;
    h0 = bop3<kXor3>(wst, c, est);
    h1 = bop3<kMaj>(wst, c, est);
}

// count = above-triple + below-triple + own-pair  (max 3 + 3 + 2 = 8)               (7 ops)
template <typename T>
__device__ __forceinline__ Count4<T> neighbour_count(T a0, T a1, T b0, T b1, T p0, T p1) {
    Count4<T> r;
    r.s0 = bop3<kXor3>(a0, b0, p0);
    const T c0 = bop3<kMaj>(a0, b0, p0);        // carry out of bit 0
    const T t1 = bop3<kXor3>(a1, b1, p1);       // weight-2 inputs: a1, b1, p1, c0
    const T m1 = bop3<kMaj>(a1, b1, p1);
    r.s1 = t1 ^ c0;
    r.s2 = bop3<kAxorBC>(m1, t1, c0);           // weight 4: m1 + (t1 & c0)
    r.s3 = bop3<kAnd3>(m1, t1, c0);             // count == 8
    return r;
}

template <int K>
struct EqTable {   // truth table of "(s0,s1,s2) spells K & 7" with a = s0, b = s1, c = s2
    static constexpr uint32_t value =
        (((K & 1) ? kA : ~kA) & ((K & 2) ? kB : ~kB) & ((K & 4) ? kC : ~kC)) & 0xFF;
};

template <int K, typename T>
__device__ __forceinline__ T rule_term(T acc, const Count4<T> &n, T c, uint32_t sk, uint32_t bk) {
    T eq;
    if (K == 8) eq = n.s3;
    else if (K == 0) eq = bop3<EqTable<0>::value>(n.s0, n.s1, n.s2) & ~n.s3;   // 8 also has s0=s1=s2=0
    else eq = bop3<EqTable<K>::value>(n.s0, n.s1, n.s2);
    if (sk & bk) return acc | eq;                       // uniform selects: rule bits are SGPRs
    if (sk) return bop3<kOrAnd>(acc, eq, c);
    return bop3<kOrAndN>(acc, eq, c);
}

// Rule lookup, sparse form: one equality mask per neighbour count the rule actually uses
// (wave-uniform branches), 2 ops per used count.  Life-like rules use 2-5 counts.
template <typename T>
__device__ __forceinline__ T apply_rule_sparse(const Count4<T> &n, T c, uint32_t rule) {
    T nxt = T(0);
#define CARLE_RULE_TERM(K)                                                          \
    {                                                                               \
        const uint32_t sk = (rule >> (K)) & 1u, bk = (rule >> (9 + (K))) & 1u;      \
        if (sk | bk) nxt = rule_term<K>(nxt, n, c, sk, bk);                         \
    }
    CARLE_RULE_TERM(0) CARLE_RULE_TERM(1) CARLE_RULE_TERM(2) CARLE_RULE_TERM(3) CARLE_RULE_TERM(4)
    CARLE_RULE_TERM(5) CARLE_RULE_TERM(6) CARLE_RULE_TERM(7) CARLE_RULE_TERM(8)
#undef CARLE_RULE_TERM
    return nxt;
}

// Rule lookup, dense form: multiplexer tree over the count bits; the select masks are scalar
// constants (17 ops, independent of the rule).
template <typename T>
__device__ __forceinline__ T apply_rule(const Count4<T> &n, T c, uint32_t rule) {
    auto leaf = [&](int k) -> T {       // next state if count == k: alive ? survive_k : birth_k
        const T sk = ((rule >> k) & 1u) ? ~T(0) : T(0);
        const T bk = ((rule >> (9 + k)) & 1u) ? ~T(0) : T(0);
        return bop3<kMux>(c, sk, bk);
    };
    const T m01 = bop3<kMux>(n.s0, leaf(1), leaf(0));
    const T m23 = bop3<kMux>(n.s0, leaf(3), leaf(2));
    const T m45 = bop3<kMux>(n.s0, leaf(5), leaf(4));
    const T m67 = bop3<kMux>(n.s0, leaf(7), leaf(6));
    const T m03 = bop3<kMux>(n.s1, m23, m01);
    const T m47 = bop3<kMux>(n.s1, m67, m45);
    const T m07 = bop3<kMux>(n.s2, m47, m03);
    return bop3<kMux>(n.s3, leaf(8), m07);      // s3 set implies count == 8 exactly
}

// The same for NW words at once: the uniform branch on each rule bit is taken once per wave, not
// once per word (kernels that keep a whole row of words per lane).
template <int NW, typename T>
__device__ __forceinline__ void apply_rule_words(const Count4<T> (&n)[NW], const T (&c)[NW],
                                                 uint32_t rule, T (&nxt)[NW]) {
    const int used = __builtin_popcount((rule | (rule >> 9)) & 0x1FFu);
    if (used > 7) {
#pragma unroll
        for (int j = 0; j < NW; ++j) nxt[j] = apply_rule(n[j], c[j], rule);
        return;
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) nxt[j] = T(0);
#define CARLE_RULE_TERM(K)                                                          \
    {                                                                               \
        const uint32_t sk = (rule >> (K)) & 1u, bk = (rule >> (9 + (K))) & 1u;      \
        if (sk | bk) {                                                              \
            _Pragma("unroll") for (int j = 0; j < NW; ++j)                          \
                nxt[j] = rule_term<K>(nxt[j], n[j], c[j], sk, bk);                  \
        }                                                                           \
    }
    CARLE_RULE_TERM(0) CARLE_RULE_TERM(1) CARLE_RULE_TERM(2) CARLE_RULE_TERM(3) CARLE_RULE_TERM(4)
    CARLE_RULE_TERM(5) CARLE_RULE_TERM(6) CARLE_RULE_TERM(7) CARLE_RULE_TERM(8)
#undef CARLE_RULE_TERM
}

// dense rules (many used counts) -> mux tree; sparse rules -> equality masks (uniform branch)
template <typename T>
__device__ __forceinline__ T apply_rule_auto(const Count4<T> &n, T c, uint32_t rule) {
    const int used = __builtin_popcount((rule | (rule >> 9)) & 0x1FFu);
    if (used <= 7) return apply_rule_sparse(n, c, rule);
    return apply_rule(n, c, rule);
}

}  // namespace carle

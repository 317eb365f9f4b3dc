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
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace carle {

template <typename T>
struct Count4 { T s0, s1, s2, s3; };   // neighbour count 0..8, one bit-plane per binary digit

// horizontal sums of one row from its west-shifted / centre / east-shifted words
template <typename T>
__device__ __forceinline__ void row_sums(T wst, T c, T est, T &h0, T &h1, T &p0, T &p1) {
    p0 = wst ^ est;
    p1 = wst & est;
    h0 = p0 ^ c;
    h1 = p1 | (p0 & c);                 // majority(wst, c, est)
}

// count = above-triple + below-triple + own-pair  (max 3 + 3 + 2 = 8)
template <typename T>
__device__ __forceinline__ Count4<T> neighbour_count(T a0, T a1, T b0, T b1, T p0, T p1) {
    Count4<T> r;
    T x0 = a0 ^ b0;
    r.s0 = x0 ^ p0;
    T c0 = (a0 & b0) | (x0 & p0);       // carry out of bit 0
    T x = a1 ^ b1, y = a1 & b1;         // weight-2 inputs: a1, b1, p1, c0
    T z = p1 ^ c0, w = p1 & c0;
    r.s1 = x ^ z;
    T xz = x & z;                       // weight-4 inputs: y, w, xz (at most two are set)
    r.s2 = y ^ w ^ xz;
    r.s3 = y & w;                       // count == 8
    return r;
}

// Rule lookup, sparse form: one equality mask per neighbour count that the rule actually uses
// (wave-uniform branches on the SGPR rule bits), 3 three-input boolean ops per used count and
// 32-bit half.  Life-like rules use 2-5 counts, so this beats the dense mux tree below.
template <typename T>
__device__ __forceinline__ T apply_rule_sparse(const Count4<T> &n, T c, uint32_t rule) {
    T nxt = T(0);
#pragma unroll
    for (int k = 0; k <= 8; ++k) {
        const uint32_t sk = (rule >> k) & 1u, bk = (rule >> (9 + k)) & 1u;
        if (sk | bk) {                                     // uniform: rule bits live in SGPRs
            T eq;
            if (k == 8) {
                eq = n.s3;                                 // count 8 <=> s3
            } else {
                const T b0 = (k & 1) ? n.s0 : ~n.s0;
                const T b1 = (k & 2) ? n.s1 : ~n.s1;
                const T b2 = (k & 4) ? n.s2 : ~n.s2;
                eq = (b0 & b1) & (b2 & ~n.s3);
            }
            if (sk & bk) nxt |= eq;
            else if (sk) nxt |= eq & c;
            else nxt |= eq & ~c;
        }
    }
    return nxt;
}

// Rule lookup as a multiplexer tree over the count bits; the 18 rule bits are wave-uniform
// (SGPRs), so every select mask below is a scalar constant and each line is one 3-input
// boolean op per 32-bit half.
template <typename T>
__device__ __forceinline__ T apply_rule(const Count4<T> &n, T c, uint32_t rule) {
    auto leaf = [&](int k) -> T {       // next state if count == k: alive ? survive_k : birth_k
        T sk = ((rule >> k) & 1u) ? ~T(0) : T(0);
        T bk = ((rule >> (9 + k)) & 1u) ? ~T(0) : T(0);
        return (c & sk) | (~c & bk);
    };
    auto mux = [](T sel, T hi, T lo) -> T { return (sel & hi) | (~sel & lo); };
    T m01 = mux(n.s0, leaf(1), leaf(0));
    T m23 = mux(n.s0, leaf(3), leaf(2));
    T m45 = mux(n.s0, leaf(5), leaf(4));
    T m67 = mux(n.s0, leaf(7), leaf(6));
    T m03 = mux(n.s1, m23, m01);
    T m47 = mux(n.s1, m67, m45);
    T m07 = mux(n.s2, m47, m03);
    return mux(n.s3, leaf(8), m07);     // s3 set implies count == 8 exactly
}

// dense rules (many used counts) -> mux tree (17 ops/half); sparse rules -> equality masks
template <typename T>
__device__ __forceinline__ T apply_rule_auto(const Count4<T> &n, T c, uint32_t rule) {
    const int used = __builtin_popcount((rule | (rule >> 9)) & 0x1FFu);
    if (used <= 5) return apply_rule_sparse(n, c, rule);   // uniform branch
    return apply_rule(n, c, rule);
}

}  // namespace carle

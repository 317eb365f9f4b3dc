// rnd_device.hpp -- device side of the fused RND2D forward for 64x64 binary grids, shared by the
// stand-alone bonus kernel (nets_fused.hip) and the kernel that fuses the bonus into the CA step
// (ca_rnd_fused.hip).  See nets_fused.hip for the method (512-entry conv1 tables in LDS, one
// wavefront per grid, zero-bordered wave-private planes).
#pragma once
#include "common.hpp"

namespace carle {

static constexpr int kLutFloats = 512 * 4 + 512 * 2;          // predictor float4 + random float2
static constexpr int kLinStride = 68;                          // padded row of the 16x64 weights
static constexpr int kLinFloats = 16 * kLinStride + 16;        // weights + bias, per net
static constexpr int kFusedWsFloats = kLutFloats + 2 * kLinFloats;
static constexpr int kPlaneStride = 24, kPlaneRows = 18;
static constexpr int kPlaneFloats = kPlaneRows * kPlaneStride; // zero-bordered 16x16 map

__device__ __forceinline__ uint32_t nz_bit(uint32_t bits) { return min(bits & 0x7FFFFFFFu, 1u); }

// 8 nibbles (32 bits) -> 8 bytes
__device__ __forceinline__ uint64_t spread_nibbles(uint32_t x) {
    uint64_t t = x;
    t = (t | (t << 16)) & 0x0000FFFF0000FFFFull;
    t = (t | (t << 8)) & 0x00FF00FF00FF00FFull;
    t = (t | (t << 4)) & 0x0F0F0F0F0F0F0F0Full;
    return t;
}

// nibble j of a 16-bit field -> nibble 4j of a 64-bit word
__device__ __forceinline__ uint64_t spread_field(uint32_t x) {
    return (uint64_t)(x & 0xFu) | ((uint64_t)(x & 0xF0u) << 12) | ((uint64_t)(x & 0xF00u) << 24) |
           ((uint64_t)(x & 0xF000u) << 36);
}

// LDS image of one workgroup of WAVES wavefronts
template <int WAVES>
struct RndShared {
    float4 lut_p[512];
    float2 lut_r[512];
    float lin[2 * kLinFloats];
    float planes[WAVES][6][kPlaneFloats];
    float feat[WAVES][2][64];
};

// tables -> LDS (coalesced copies), zero the plane borders once; the caller synchronises
template <int WAVES>
__device__ __forceinline__ void rnd_shared_init(RndShared<WAVES> &s, const float *__restrict__ ws, int tid) {
    constexpr int T = 64 * WAVES;
    const float4 *src = reinterpret_cast<const float4 *>(ws);
    for (int i = tid; i < 512; i += T) s.lut_p[i] = src[i];
    const float2 *src2 = reinterpret_cast<const float2 *>(ws + 2048);
    for (int i = tid; i < 512; i += T) s.lut_r[i] = src2[i];
    for (int i = tid; i < 2 * kLinFloats; i += T) s.lin[i] = ws[kLutFloats + i];
    float *pl = &s.planes[0][0][0];
    for (int i = tid; i < WAVES * 6 * kPlaneFloats; i += T) pl[i] = 0.0f;
}

// Stages 4-5 on ONE wavefront (all 64 lanes): conv2 + relu + 2x2 pool of both nets from the six
// zero-bordered 16x16 planes, the two 64->16 linear layers + tanh, and the bonus
// mean_k (target - prediction)^2 (carle/mcl.py:343-352).  The result is complete in lane 0.
__device__ __forceinline__ float rnd_tail(const float *__restrict__ planes, float (*__restrict__ feat)[64],
                                          const float *__restrict__ lin, const float *__restrict__ ws,
                                          int lane, int64_t g, float *__restrict__ pred_out,
                                          float *__restrict__ target_out) {
    float bonus_value;
    // ---- 4. conv2 (C->1, 3x3, zero pad) + relu + 2x2 pool: lane = pooled output (fy, fx)
    const int fy = lane >> 3, fx = lane & 7;
#pragma unroll
    for (int net = 0; net < 2; ++net) {
        const int nch = net == 0 ? 4 : 2, chbase = net == 0 ? 0 : 4;
        const float *cw2 = ws + kFusedWsFloats + (net == 0 ? 0 : 40);   // uniform -> SGPR loads
        float o00 = 0.f, o01 = 0.f, o10 = 0.f, o11 = 0.f;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            if (cc < nch) {
                const float *pl = planes + (chbase + cc) * kPlaneFloats + (2 * fy) * kPlaneStride + 2 * fx;
                float patch[4][4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float2 lo = *reinterpret_cast<const float2 *>(pl + r * kPlaneStride);
                    const float2 hi = *reinterpret_cast<const float2 *>(pl + r * kPlaneStride + 2);
                    patch[r][0] = lo.x; patch[r][1] = lo.y; patch[r][2] = hi.x; patch[r][3] = hi.y;
                }
                // same accumulation order as the staged kernel (channel, oy, ox, ky, kx)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) o00 = fmaf(cw2[cc * 9 + ky * 3 + kx], patch[ky][kx], o00);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) o01 = fmaf(cw2[cc * 9 + ky * 3 + kx], patch[ky][1 + kx], o01);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) o10 = fmaf(cw2[cc * 9 + ky * 3 + kx], patch[1 + ky][kx], o10);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) o11 = fmaf(cw2[cc * 9 + ky * 3 + kx], patch[1 + ky][1 + kx], o11);
            }
        }
        const float mx = fmaxf(fmaxf(o00, o01), fmaxf(o10, o11)) + cw2[net == 0 ? 36 : 18];
        feat[net][fy * 8 + fx] = fmaxf(mx, 0.0f);
    }
    __builtin_amdgcn_wave_barrier();

    // ---- 5. linear(64->16) + tanh for both nets; lane = (net, k16, half)
    {
        const int half = lane & 1, kk = (lane >> 1) & 15, net = lane >> 5;
        const float4 *f4 = reinterpret_cast<const float4 *>(&feat[net][32 * half]);
        const float4 *w4 = reinterpret_cast<const float4 *>(&lin[net * kLinFloats + kk * kLinStride + 32 * half]);
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 f = f4[i], w = w4[i];
            acc = fmaf(w.x, f.x, acc); acc = fmaf(w.y, f.y, acc);
            acc = fmaf(w.z, f.z, acc); acc = fmaf(w.w, f.w, acc);
        }
        acc += __shfl_xor(acc, 1, 64);
        const float out = tanhf(acc + lin[net * kLinFloats + 16 * kLinStride + kk]);
        if (half == 0) {
            if (net == 0 && pred_out) pred_out[g * 16 + kk] = out;
            if (net == 1 && target_out) target_out[g * 16 + kk] = out;
        }
        const float other = __shfl_xor(out, 32, 64);
        float sq = (out - other) * (out - other);
#pragma unroll
        for (int sh = 2; sh < 32; sh <<= 1) sq += __shfl_xor(sq, sh, 64);
        bonus_value = sq * (1.0f / 16.0f);               // complete in lane 0
    }
    return bonus_value;
}

// RND2D bonus of one grid from the wave's 16x4 patches: lane (q = L/16, k = L%16) holds rows
// 16q..16q+15, columns 4k..4k+3 as nibble r of `c`.  Returns mean_k (target - prediction)^2
// (carle/mcl.py:343-352), complete in lane 0; optionally stores the two 16-vectors.
template <int WAVES>
__device__ __forceinline__ float rnd_bonus_patch16(RndShared<WAVES> &s, const float *__restrict__ ws,
                                                   uint64_t c, int lane, int wv, int64_t g,
                                                   float *__restrict__ pred_out,
                                                   float *__restrict__ target_out) {
    const int q = lane >> 4, k = lane & 15;
    // ---- 2. 6-bit row segments [west | 4 cells | east], one byte per row; zero padding at
    //         the grid border (conv1 uses zero padding, mcl.py:263)
    uint64_t cw = ((uint64_t)__shfl((uint32_t)(c >> 32), lane - 1, 64) << 32) | __shfl((uint32_t)c, lane - 1, 64);
    uint64_t ce = ((uint64_t)__shfl((uint32_t)(c >> 32), lane + 1, 64) << 32) | __shfl((uint32_t)c, lane + 1, 64);
    if (k == 0) cw = 0;
    if (k == 15) ce = 0;
    const uint64_t wb = (cw >> 3) & 0x1111111111111111ull, eb = ce & 0x1111111111111111ull;
    const uint64_t ea = (spread_nibbles((uint32_t)c) << 1) | spread_nibbles((uint32_t)wb) |
                        (spread_nibbles((uint32_t)eb) << 5);                    // rows 0..7
    const uint64_t ebb = (spread_nibbles((uint32_t)(c >> 32)) << 1) | spread_nibbles((uint32_t)(wb >> 32)) |
                         (spread_nibbles((uint32_t)(eb >> 32)) << 5);           // rows 8..15
    uint32_t top = __shfl((uint32_t)(ebb >> 56), lane - 16, 64);               // row 16q-1
    uint32_t bot = __shfl((uint32_t)ea & 0xFFu, lane + 16, 64);                // row 16q+16
    if (q == 0) top = 0;
    if (q == 3) bot = 0;
    uint32_t seg[18];
    seg[0] = top;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        seg[1 + r] = (uint32_t)(ea >> (8 * r)) & 0xFFu;
        seg[9 + r] = (uint32_t)(ebb >> (8 * r)) & 0xFFu;
    }
    seg[17] = bot;

    // ---- 3. conv1 (table lookup) + relu + 4x4 max-pool -> zero-bordered planes in LDS
    float *planes = &s.planes[wv][0][0];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f, m5 = 0.f;
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) {
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) {
                const uint32_t p = ((seg[4 * j + dy] >> dx) & 7u) |
                                   (((seg[4 * j + dy + 1] >> dx) & 7u) << 3) |
                                   (((seg[4 * j + dy + 2] >> dx) & 7u) << 6);
                const float4 a = s.lut_p[p];
                const float2 b = s.lut_r[p];
                m0 = fmaxf(m0, a.x); m1 = fmaxf(m1, a.y); m2 = fmaxf(m2, a.z); m3 = fmaxf(m3, a.w);
                m4 = fmaxf(m4, b.x); m5 = fmaxf(m5, b.y);
            }
        }
        const int off = (4 * q + j + 1) * kPlaneStride + k + 1;
        planes[0 * kPlaneFloats + off] = m0;
        planes[1 * kPlaneFloats + off] = m1;
        planes[2 * kPlaneFloats + off] = m2;
        planes[3 * kPlaneFloats + off] = m3;
        planes[4 * kPlaneFloats + off] = m4;
        planes[5 * kPlaneFloats + off] = m5;
    }
    __builtin_amdgcn_wave_barrier();

    __builtin_amdgcn_wave_barrier();
    const float bonus_value = rnd_tail(planes, s.feat[wv], s.lin, ws, lane, g, pred_out, target_out);
    __builtin_amdgcn_wave_barrier();
    return bonus_value;
}

}  // namespace carle

// rnd_device.hpp -- device side of the fused RND2D forward for 64x64 binary grids, shared by the
// stand-alone bonus kernel (nets_fused.hip) and the kernel that fuses the bonus into the CA step
// (ca_rnd_fused.hip).  See nets_fused.hip for the method (512-entry conv1 tables in LDS, one
// workgroup per grid, zero-bordered LDS planes).
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

// LDS image of one workgroup (4 wavefronts) that owns one 64x64 grid at a time
struct RndBlockShared {
    float4 lut_p[512];                 // conv1 + relu of the predictor, per 9-bit neighbourhood pattern
    float2 lut_r[512];                 // ... of the random network
    float lin[2 * kLinFloats];
    float planes[6][kPlaneFloats];     // pooled conv1 maps, zero-bordered 16x16
    uint64_t rows[66];                 // the grid, one 64-bit word per row at [1..64]; [0], [65] stay zero
};

// tables -> LDS (coalesced copies), zero the plane borders and the two padding rows once; the
// caller synchronises
__device__ __forceinline__ void rnd_block_init(RndBlockShared &s, const float *__restrict__ ws, int tid) {
    const float4 *src = reinterpret_cast<const float4 *>(ws);
    for (int i = tid; i < 512; i += 256) s.lut_p[i] = src[i];
    const float2 *src2 = reinterpret_cast<const float2 *>(ws + 2048);
    for (int i = tid; i < 512; i += 256) s.lut_r[i] = src2[i];
    static_assert((2 * kLinFloats) % 4 == 0 && (6 * kPlaneFloats) % 4 == 0 && kLutFloats % 4 == 0, "float4 copies");
    const float4 *lsrc = reinterpret_cast<const float4 *>(ws + kLutFloats);
    float4 *ldst = reinterpret_cast<float4 *>(s.lin);
    for (int i = tid; i < 2 * kLinFloats / 4; i += 256) ldst[i] = lsrc[i];
    float4 *pl = reinterpret_cast<float4 *>(&s.planes[0][0]);
    for (int i = tid; i < 6 * kPlaneFloats / 4; i += 256) pl[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 66) s.rows[tid] = 0ull;
}

// The 16 lanes of a DPP row hold the nibbles (columns 4k..4k+3, k = lane % 16) of ONE grid row:
// OR them into the row's 64-bit word.  Lanes k = 0..7 build the low dword, k = 8..15 the high one
// (3 row_shr steps each); returns the dword in lanes k = 7 and k = 15.
__device__ __forceinline__ uint32_t row_nibbles_to_dword(uint32_t nib, int k) {
    uint32_t x = nib << (4 * (k & 7));
    x |= __builtin_amdgcn_update_dpp(0u, x, 0x111, 0xF, 0xF, true);   // row_shr:1
    x |= __builtin_amdgcn_update_dpp(0u, x, 0x112, 0xF, 0xF, true);   // row_shr:2
    x |= __builtin_amdgcn_update_dpp(0u, x, 0x114, 0xF, 0xF, true);   // row_shr:4
    return x;
}

// conv1 (table lookup) + relu + 4x4 max-pool of all six channels; thread = pooling window
// (py = tid/16, px = tid%16), zero padding at the grid border (mcl.py:263)
__device__ __forceinline__ void rnd_block_conv1(RndBlockShared &s, int tid) {
    const int py = tid >> 4, px = tid & 15;
    uint32_t seg[6];                                  // [west | 4 cells | east] of rows 4py-1..4py+4
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const uint64_t row = s.rows[4 * py + r];      // rows[1 + (4py - 1 + r)]
        seg[r] = (uint32_t)(px == 0 ? row << 1 : row >> (4 * px - 1)) & 0x3Fu;
    }
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f, m5 = 0.f;
#pragma unroll
    for (int dy = 0; dy < 4; ++dy) {
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
            const uint32_t p = ((seg[dy] >> dx) & 7u) | (((seg[dy + 1] >> dx) & 7u) << 3) |
                               (((seg[dy + 2] >> dx) & 7u) << 6);
            const float4 pa = s.lut_p[p];
            const float2 pb = s.lut_r[p];
            m0 = fmaxf(m0, pa.x); m1 = fmaxf(m1, pa.y); m2 = fmaxf(m2, pa.z); m3 = fmaxf(m3, pa.w);
            m4 = fmaxf(m4, pb.x); m5 = fmaxf(m5, pb.y);
        }
    }
    const int off = (py + 1) * kPlaneStride + px + 1;
    s.planes[0][off] = m0;
    s.planes[1][off] = m1;
    s.planes[2][off] = m2;
    s.planes[3][off] = m3;
    s.planes[4][off] = m4;
    s.planes[5][off] = m5;
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

}  // namespace carle

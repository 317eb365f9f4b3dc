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
// behind the tables: conv2 parameters (64 floats), an all-dead bit-packed grid (64 row words) and
// the bonus of that grid -- a function of the parameters alone, evaluated when the tables are built
static constexpr int kWsZeroGrid = kFusedWsFloats + 64;
static constexpr int kWsDeadBonus = kWsZeroGrid + 128;
static constexpr int kFusedWsTotal = kWsDeadBonus + 4;
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
    float w2[2][4][12];                // conv2 weights per net and input channel: 9 taps + 3 zeros;
                                       // rows [1][2], [1][3] are zero (the random net has two channels)
    float b2[4];                       // conv2 biases: [0] predictor, [1] random net
};

// tables -> LDS (coalesced copies), zero the plane borders and the two padding rows once; the
// caller synchronises
// Every global load is requested BEFORE the first LDS store: written as copy loops, each load was followed by its
// own s_waitcnt vmcnt(0) -- eight dependent round trips at the head of every workgroup (and the first of them also
// waited for the first grid's loads, which the fused CA + RND2D kernel requests ahead of this): round 2, seen in the
// disassembly.
__device__ __forceinline__ void rnd_block_init(RndBlockShared &s, const float *__restrict__ ws, int tid) {
    static_assert((2 * kLinFloats) % 4 == 0 && (6 * kPlaneFloats) % 4 == 0 && kLutFloats % 4 == 0, "float4 copies");
    constexpr int kLin4 = 2 * kLinFloats / 4, kLinIt = (kLin4 + 255) / 256;
    const float4 *src = reinterpret_cast<const float4 *>(ws);
    const float2 *src2 = reinterpret_cast<const float2 *>(ws + 2048);
    const float4 *lsrc = reinterpret_cast<const float4 *>(ws + kLutFloats);
    const float4 p0 = src[tid], p1 = src[tid + 256];
    const float2 r0 = src2[tid], r1 = src2[tid + 256];
    // tails without a branch (the compiler would sink the load into it and wait there): a thread beyond the end
    // repeats its previous element, load and store -- a benign duplicate
    float4 l[kLinIt];
    int li[kLinIt];
#pragma unroll
    for (int k = 0; k < kLinIt; ++k) { const int i = tid + 256 * k; li[k] = i < kLin4 ? i : i - 256; l[k] = lsrc[li[k]]; }
    // conv2 parameters (behind the tables in the workspace: pred w[36] b at +0, rand w[18] b at +40): every thread
    // handles entry tid % 96 of w2 and entry tid % 4 of b2
    const int e = tid % 96, net = e / 48, c = (e % 48) / 12, t = e % 12;
    const bool w2_in = t < 9 && (net == 0 || c < 2);
    const float w2v = ws[kFusedWsFloats + (w2_in ? (net == 0 ? 0 : 40) + c * 9 + t : 0)];
    const int be = tid & 3;
    const float b2v = ws[kFusedWsFloats + (be == 1 ? 58 : 36)];

    s.lut_p[tid] = p0; s.lut_p[tid + 256] = p1;
    s.lut_r[tid] = r0; s.lut_r[tid + 256] = r1;
    float4 *ldst = reinterpret_cast<float4 *>(s.lin);
#pragma unroll
    for (int k = 0; k < kLinIt; ++k) ldst[li[k]] = l[k];
    float4 *pl = reinterpret_cast<float4 *>(&s.planes[0][0]);
    for (int i = tid; i < 6 * kPlaneFloats / 4; i += 256) pl[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 66) s.rows[tid] = 0ull;
    (&s.w2[0][0][0])[e] = w2_in ? w2v : 0.0f;
    s.b2[be] = be < 2 ? b2v : 0.0f;
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
// (py = tid/16, px = tid%16), zero padding at the grid border (mcl.py:263).  The 16 lookups of a
// window are issued in batches of BATCH (1.5 LDS reads and 6 registers each): a dependent LDS round
// trip costs 250-500 cycles with 16 waves on the CU, so few round trips -- as many lookups in
// flight as the kernel's register budget allows.
#ifndef CARLE_RND_CONV1_BATCH
#define CARLE_RND_CONV1_BATCH 4
#endif
// Timing probe only (tools/ab_libs.sh builds, EXPERIMENTS.md 5.2): fewer than 16 lookups per window give WRONG planes --
// what halving the lookups (a 12-bit pair table) could save at most, with occupancy and table size as they are
#ifndef CARLE_RND_CONV1_LOOKUPS
#define CARLE_RND_CONV1_LOOKUPS 16
#endif
// VMEM_P / VMEM_R: of the window's 16 lookups, the first VMEM_P (predictor table) / VMEM_R (random-net table) read the
// table's copy in the workspace through the vector L1 instead of LDS (ws: the workspace the tables were copied from)
template <int BATCH = CARLE_RND_CONV1_BATCH, int VMEM_P = 0, int VMEM_R = 0>
__device__ __forceinline__ void rnd_block_conv1(RndBlockShared &s, int tid, const float *__restrict__ ws = nullptr) {
    static_assert(16 % BATCH == 0, "whole batches");
    const int py = tid >> 4, px = tid & 15;
    uint32_t seg[6];                                  // [west | 4 cells | east] of rows 4py-1..4py+4
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const uint64_t row = s.rows[4 * py + r];      // rows[1 + (4py - 1 + r)]
        seg[r] = (uint32_t)(px == 0 ? row << 1 : row >> (4 * px - 1)) & 0x3Fu;
    }
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f, m5 = 0.f;
#pragma unroll
    for (int b = 0; b < CARLE_RND_CONV1_LOOKUPS / BATCH; ++b) {
        float4 pa[BATCH];
        float2 pb[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            const int c = b * BATCH + j, dy = c >> 2, dx = c & 3;
            const uint32_t p = ((seg[dy] >> dx) & 7u) | (((seg[dy + 1] >> dx) & 7u) << 3) |
                               (((seg[dy + 2] >> dx) & 7u) << 6);
            pa[j] = c < VMEM_P ? reinterpret_cast<const float4 *>(ws)[p] : s.lut_p[p];
            pb[j] = c < VMEM_R ? reinterpret_cast<const float2 *>(ws + 2048)[p] : s.lut_r[p];
        }
        __builtin_amdgcn_sched_barrier(0);            // the whole batch is issued before its first max
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            m0 = fmaxf(m0, pa[j].x); m1 = fmaxf(m1, pa[j].y); m2 = fmaxf(m2, pa[j].z); m3 = fmaxf(m3, pa[j].w);
            m4 = fmaxf(m4, pb[j].x); m5 = fmaxf(m5, pb[j].y);
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

__device__ __forceinline__ float dpp_quad_xor1(float x) {      // value of lane ^ 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_quad_xor2(float x) {      // value of lane ^ 2
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_row_ror(float x, int ctrl) {   // 0x124 = row_ror:4, 0x128 = row_ror:8
    return ctrl == 0x124
        ? __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xF, 0xF, true))
        : __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xF, 0xF, true));
}

// Stage 4 on ALL 256 threads: conv2 (C->1, 3x3, zero pad) + bias + relu + 2x2 max-pool of both nets
// from the six zero-bordered 16x16 planes.  Thread = (pooling window f = tid / 4, input channel
// c = tid % 4): it reads the 4x4 patch of ITS channel under the window (predictor plane c and, for
// c < 2, random-net plane 4 + c; zero weights otherwise) plus its 9 + 9 weights -- every LDS read
// is issued before the first use, ONE round trip -- computes the window's four conv outputs for
// that channel, and the four channel lanes (a DPP quad) add up.  Summation order: taps (ky, kx)
// inside a channel, then (c0 + c1) + (c2 + c3).
__device__ __forceinline__ void rnd_block_conv2(const RndBlockShared &s, float (*__restrict__ feat)[64], int tid) {
    const int f = tid >> 2, c = tid & 3;
    const int fy = f >> 3, fx = f & 7;
    float patch[2][4][4], w[2][12];
    // the sixteen patch halves as single ds_read_b64, written out: the compiler merges pairs of 8-byte LDS loads into ds_read2_b64,
    // which occupies the LDS array for 8 cycles where two ds_read_b64 take 2 + 2 (MI355X_MICROARCH, LDS; ae_block64_kernel).
    // Fused CA + RND2D step 30.0 -> 28.7 us, rnd_block64_kernel 20.85 -> 20.4 us (same box, round 5)
    f32x2 ph[2][8];
#pragma unroll
    for (int net = 0; net < 2; ++net) {
        const uint32_t a = (uint32_t)reinterpret_cast<uintptr_t>(&s.planes[net == 0 ? c : 4 + (c & 1)][(2 * fy) * kPlaneStride + 2 * fx]);
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(ph[net][0]) : "v"(a), "n"(0 * kPlaneStride * 4) : "memory");
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(ph[net][1]) : "v"(a), "n"(0 * kPlaneStride * 4 + 8) : "memory");
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(ph[net][2]) : "v"(a), "n"(1 * kPlaneStride * 4) : "memory");
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(ph[net][3]) : "v"(a), "n"(1 * kPlaneStride * 4 + 8) : "memory");
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(ph[net][4]) : "v"(a), "n"(2 * kPlaneStride * 4) : "memory");
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(ph[net][5]) : "v"(a), "n"(2 * kPlaneStride * 4 + 8) : "memory");
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(ph[net][6]) : "v"(a), "n"(3 * kPlaneStride * 4) : "memory");
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(ph[net][7]) : "v"(a), "n"(3 * kPlaneStride * 4 + 8) : "memory");
    }
#pragma unroll
    for (int net = 0; net < 2; ++net) {
        const float4 *wp = reinterpret_cast<const float4 *>(&s.w2[net][c][0]);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float4 t = wp[i];
            w[net][4 * i] = t.x; w[net][4 * i + 1] = t.y; w[net][4 * i + 2] = t.z; w[net][4 * i + 3] = t.w;
        }
    }
    // (the weights' loads above are in flight with the patches: ONE round trip)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(ph[0][0]), "+v"(ph[0][1]), "+v"(ph[0][2]), "+v"(ph[0][3]), "+v"(ph[0][4]), "+v"(ph[0][5]), "+v"(ph[0][6]), "+v"(ph[0][7]),
                   "+v"(ph[1][0]), "+v"(ph[1][1]), "+v"(ph[1][2]), "+v"(ph[1][3]), "+v"(ph[1][4]), "+v"(ph[1][5]), "+v"(ph[1][6]), "+v"(ph[1][7])
                 : : "memory");
#pragma unroll
    for (int net = 0; net < 2; ++net)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            patch[net][r][0] = ph[net][2 * r].x; patch[net][r][1] = ph[net][2 * r].y;
            patch[net][r][2] = ph[net][2 * r + 1].x; patch[net][r][3] = ph[net][2 * r + 1].y;
        }
    const float bias0 = s.b2[0], bias1 = s.b2[1];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int net = 0; net < 2; ++net) {
        float o[2][2];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                float acc = 0.f;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) acc = fmaf(w[net][ky * 3 + kx], patch[net][dy + ky][dx + kx], acc);
                acc += dpp_quad_xor1(acc);            // channels (c0 + c1), (c2 + c3)
                acc += dpp_quad_xor2(acc);            // all four
                o[dy][dx] = acc;
            }
        const float mx = fmaxf(fmaxf(o[0][0], o[0][1]), fmaxf(o[1][0], o[1][1])) + (net == 0 ? bias0 : bias1);
        if (c == 0) feat[net][f] = fmaxf(mx, 0.0f);
    }
}

// Stage 5 on ONE wavefront (all 64 lanes): the two 64->16 linear layers + tanh and the bonus
// mean_k (target - prediction)^2 (carle/mcl.py:343-352).  lane = (net, k16, half): 32 inputs per
// lane, eight LDS reads in flight at a time; the two halves, the 16 outputs and the two nets meet
// through DPP / v_permlane32_swap / v_readlane (no LDS round trips).  Wave-uniform result.
__device__ __forceinline__ float rnd_linear_bonus(const float (*__restrict__ feat)[64], const float *__restrict__ lin,
                                                  int lane, int64_t g, float *__restrict__ pred_out,
                                                  float *__restrict__ target_out) {
    const int half = lane & 1, kk = (lane >> 1) & 15, net = lane >> 5;
    const float4 *f4 = reinterpret_cast<const float4 *>(&feat[net][32 * half]);
    const float4 *w4 = reinterpret_cast<const float4 *>(&lin[net * kLinFloats + kk * kLinStride + 32 * half]);
    const float bias = lin[net * kLinFloats + 16 * kLinStride + kk];
    float acc = 0.0f;
#pragma unroll
    for (int b = 0; b < 2; ++b) {                         // two batches of 4 + 4 reads (32 registers)
        float4 fv[4], wv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { fv[i] = f4[4 * b + i]; wv[i] = w4[4 * b + i]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc = fmaf(wv[i].x, fv[i].x, acc); acc = fmaf(wv[i].y, fv[i].y, acc);
            acc = fmaf(wv[i].z, fv[i].z, acc); acc = fmaf(wv[i].w, fv[i].w, acc);
        }
    }
    acc += dpp_quad_xor1(acc);                            // the other half of the inputs
    const float out = tanhf(acc + bias);
    if (half == 0) {
        if (net == 0 && pred_out) pred_out[g * 16 + kk] = out;
        if (net == 1 && target_out) target_out[g * 16 + kk] = out;
    }
    // the other net's output for the same k sits 32 lanes away: v_permlane32_swap exchanges the
    // upper half of its first operand with the lower half of its second
    const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, out),
                                                     __builtin_bit_cast(unsigned, out), false, false);
    const float other = __builtin_bit_cast(float, lane < 32 ? sw[1] : sw[0]);
    float sq = (out - other) * (out - other);
    sq += dpp_quad_xor2(sq);                              // k ^ 1
    sq += dpp_row_ror(sq, 0x124);                         // + the other k's of this 16-lane row: eight per
    sq += dpp_row_ror(sq, 0x128);                         //   row, each once (lane bits 1..3 are summed)
    const float total = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sq), 0)) +
                        __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sq), 16));
    return total * (1.0f / 16.0f);
}

}  // namespace carle

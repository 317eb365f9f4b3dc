// ae_device.hpp -- device side of the AE2D forward (carle/mcl.py:452-467, 504-512) for ONE 64x64 binary grid held by a
// workgroup of EIGHT wavefronts, as the phases of the one-launch AE2D(RND2D(CARLE)) step (ca_rnd_fused.hip:
// ca_step_rnd_ae64_kernel).  The arithmetic of ae_block64_kernel (nets_fused.hip) on twice the lanes, operation for
// operation in the same order, so the bonus is bit-identical to that kernel's:
//   conv1 (512-entry table) + relu + 2x2 pool -> A1 (4 x 32x32)   fused into the RND2D lookups: the 9-bit pattern of a
//                                                                  cell is formed ONCE for the ten channels of both nets
//   conv2 (4->2) + relu + 2x2 pool            -> A2 (2 x 16x16)   thread = (output channel, 4x4 block)
//   convT1 (2->1, k4 s2 p1) + relu            -> A3 (32x32)       thread = (row parity, 4x4 block); A3 overlays A1 plane 0
//   convT2 (1->1) + sigmoid, squared error     -> tsum (256)      lane pair = one 4x4 block, two rows each; the pair's
//                                                                  second lane continues the first one's fma chain
//   sum of the 256 block errors                -> bonus           ONE wavefront replays ae_block64_kernel's reduction
//                                                                  (four 64-lane xor trees, then (r0 + r1) + (r2 + r3))
#pragma once
#include "rnd_device.hpp"

namespace carle {

// workspace layout of ae2d_prepare_tables (nets_fused.hip): conv1 table, then conv2.w[72] b[2] | convT1.w[32] b[1] |
// convT2.w[16] b[1]
static constexpr int kAeX8LutFloats = 512 * 4;
static constexpr int kAeX8A1Stride = 36, kAeX8A1Plane = 34 * 36;
static constexpr int kAeX8A2Stride = 20, kAeX8A2Plane = 18 * 20;

struct AeBlockSharedX8 {
    float4 lut[512];                     // conv1 + relu of the AE2D predictor per 9-bit neighbourhood pattern
    float a1[4 * kAeX8A1Plane];          // A1, zero-bordered; A3 later overlays plane 0
    float a2[2 * kAeX8A2Plane];          // A2, zero-bordered
    float tsum[256];                     // squared error of every 4x4 block
};

__device__ __forceinline__ float ae_x8_sigmoid(float x) {       // = fast_sigmoid of nets_fused.hip
    return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}

// table -> LDS, planes zeroed (their borders stay zero); 512 threads; the caller synchronises
__device__ __forceinline__ void ae_x8_init(AeBlockSharedX8 &a, const float *__restrict__ ws_a, int tid) {
    const float4 e = reinterpret_cast<const float4 *>(ws_a)[tid];
    a.lut[tid] = e;
    for (int i = tid; i < 4 * kAeX8A1Plane; i += 512) a.a1[i] = 0.0f;
    for (int i = tid; i < 2 * kAeX8A2Plane; i += 512) a.a2[i] = 0.0f;
}

// conv1 of BOTH reward nets: thread = (4x4 window tid / 2, half tid % 2) -- rnd_block_conv1_x8 with the AE2D table read
// beside the two RND2D tables at the same address index.  RND2D: maxima over the window (the halves meet by DPP); AE2D:
// maxima over the 2x2 blocks, of which the half owns the two in its pair of rows (A1 rows 2 py + half).
__device__ __forceinline__ void rnd_ae_conv1_x8(RndBlockShared &s, AeBlockSharedX8 &a, int tid) {
    const int win = tid >> 1, half = tid & 1;
    const int py = win >> 4, px = win & 15;
    uint32_t seg[4];                                  // [west | 4 cells | east] of rows 4py-1+2half .. 4py+2+2half
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint64_t row = s.rows[4 * py + 2 * half + r];
        seg[r] = (uint32_t)(px == 0 ? row << 1 : row >> (4 * px - 1)) & 0x3Fu;
    }
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f, m5 = 0.f;
    float e[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};       // AE2D: [2x2 block column b][channel]
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {                  // two batches of four lookups (ten floats each)
        float4 pa[4], qa[4];
        float2 pb[4];
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
            const uint32_t p = ((seg[dy] >> dx) & 7u) | (((seg[dy + 1] >> dx) & 7u) << 3) | (((seg[dy + 2] >> dx) & 7u) << 6);
            pa[dx] = s.lut_p[p];
            pb[dx] = s.lut_r[p];
            qa[dx] = a.lut[p];
        }
        __builtin_amdgcn_sched_barrier(0);            // the whole batch is issued before its first max
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
            m0 = fmaxf(m0, pa[dx].x); m1 = fmaxf(m1, pa[dx].y); m2 = fmaxf(m2, pa[dx].z); m3 = fmaxf(m3, pa[dx].w);
            m4 = fmaxf(m4, pb[dx].x); m5 = fmaxf(m5, pb[dx].y);
            const int b = dx >> 1;
            e[b][0] = fmaxf(e[b][0], qa[dx].x); e[b][1] = fmaxf(e[b][1], qa[dx].y);
            e[b][2] = fmaxf(e[b][2], qa[dx].z); e[b][3] = fmaxf(e[b][3], qa[dx].w);
        }
    }
    m0 = fmaxf(m0, dpp_quad_xor1(m0)); m1 = fmaxf(m1, dpp_quad_xor1(m1)); m2 = fmaxf(m2, dpp_quad_xor1(m2));
    m3 = fmaxf(m3, dpp_quad_xor1(m3)); m4 = fmaxf(m4, dpp_quad_xor1(m4)); m5 = fmaxf(m5, dpp_quad_xor1(m5));
    const int off = (py + 1) * kPlaneStride + px + 1;
    s.planes[3 * half][off] = half ? m3 : m0;
    s.planes[3 * half + 1][off] = half ? m4 : m1;
    s.planes[3 * half + 2][off] = half ? m5 : m2;
    const int o1 = (2 * py + half + 1) * kAeX8A1Stride + 2 * px + 1;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {                  // (columns 2px + 1, 2px + 2 of the padded row: not an aligned pair)
        a.a1[ch * kAeX8A1Plane + o1] = e[0][ch];
        a.a1[ch * kAeX8A1Plane + o1 + 1] = e[1][ch];
    }
}

// conv2 (4->2, 3x3, zero pad) + relu + 2x2 pool -> A2: thread = (output channel tid / 256, block (py, px))
__device__ __forceinline__ void ae_conv2_x8(AeBlockSharedX8 &a, const float *__restrict__ prm, int tid) {
    const int oc = __builtin_amdgcn_readfirstlane(tid >> 8);
    const int py = (tid & 255) >> 4, px = tid & 15;
    kfloat *w2 = opaque_uniform(prm) + oc * 36;
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
        const float *pl = a.a1 + cc * kAeX8A1Plane + (2 * py) * kAeX8A1Stride + 2 * px;   // padded rows 2py.., cols 2px..
        float patch[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float2 lo = *reinterpret_cast<const float2 *>(pl + r * kAeX8A1Stride);
            const float2 hi = *reinterpret_cast<const float2 *>(pl + r * kAeX8A1Stride + 2);
            patch[r][0] = lo.x; patch[r][1] = lo.y; patch[r][2] = hi.x; patch[r][3] = hi.y;
        }
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
                        acc[dy][dx] = fmaf(w2[cc * 9 + ky * 3 + kx], patch[dy + ky][dx + kx], acc[dy][dx]);
    }
    const float bias = opaque_uniform(prm)[72 + oc];
    const float mx = fmaxf(fmaxf(acc[0][0], acc[0][1]), fmaxf(acc[1][0], acc[1][1])) + bias;
    a.a2[oc * kAeX8A2Plane + (py + 1) * kAeX8A2Stride + px + 1] = fmaxf(mx, 0.0f);
}

// convT1 (2->1, k4 s2 p1) + relu -> A3 rows 2 py + J of the block, overlaying A1 plane 0: thread = (row parity J, block)
template <int J>
__device__ __forceinline__ void ae_convT1_rows(AeBlockSharedX8 &a, const float *__restrict__ prm, int py, int px) {
    kfloat *wt1 = opaque_uniform(prm) + 74;
    const float bt1 = wt1[32];
    // output row y = 2py+J: y even: (iy=py, ky=1), (iy=py-1, ky=3);  y odd: (iy=py, ky=2), (iy=py+1, ky=0)
    constexpr int ry1 = J ? 2 : 0, ky0 = J ? 2 : 1, ky1 = J ? 0 : 3;
    float r2[2][2][3];                                // A2 rows (py, then py -+ 1), cols px-1..px+1 (padded)
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            r2[cc][0][x] = a.a2[cc * kAeX8A2Plane + (py + 1) * kAeX8A2Stride + px + x];
            r2[cc][1][x] = a.a2[cc * kAeX8A2Plane + (py + ry1) * kAeX8A2Stride + px + x];
        }
    float out[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int cx0 = 1, kx0 = i ? 2 : 1;
        const int cx1 = i ? 2 : 0, kx1 = i ? 0 : 3;
        float acc = 0.0f;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            acc = fmaf(r2[cc][0][cx0], wt1[cc * 16 + ky0 * 4 + kx0], acc);
            acc = fmaf(r2[cc][0][cx1], wt1[cc * 16 + ky0 * 4 + kx1], acc);
            acc = fmaf(r2[cc][1][cx0], wt1[cc * 16 + ky1 * 4 + kx0], acc);
            acc = fmaf(r2[cc][1][cx1], wt1[cc * 16 + ky1 * 4 + kx1], acc);
        }
        out[i] = fmaxf(acc + bt1, 0.0f);
    }
    a.a1[(2 * py + J + 1) * kAeX8A1Stride + 2 * px + 1] = out[0];
    a.a1[(2 * py + J + 1) * kAeX8A1Stride + 2 * px + 2] = out[1];
}

__device__ __forceinline__ void ae_convT1_x8(AeBlockSharedX8 &a, const float *__restrict__ prm, int tid) {
    const int j = __builtin_amdgcn_readfirstlane(tid >> 8);
    const int py = (tid & 255) >> 4, px = tid & 15;
    if (j) ae_convT1_rows<1>(a, prm, py, px);
    else ae_convT1_rows<0>(a, prm, py, px);
}

// convT2 (1->1) + sigmoid + squared error against the grid's own cells: lane pair (tid / 2 = block (py, px), h = tid % 2)
// with rows 4 py + 2 h, + 1 of the block.  ae_block64_kernel's thread runs ONE fma chain over its 16 cells, row by row:
// lane h = 1 continues lane h = 0's chain (both lanes run the eight fmas twice: from zero, and from the partner's sum).
__device__ __forceinline__ void ae_convT2_err_x8(AeBlockSharedX8 &a, const uint64_t *__restrict__ rows1,
                                                 const float *__restrict__ prm, int tid) {
    const int win = tid >> 1, h = tid & 1;
    const int py = win >> 4, px = win & 15;
    kfloat *wt2 = opaque_uniform(prm) + 107;
    const float bt2 = wt2[16];
    float t[3][4];                                    // A3 (padded) rows 2py + h .. 2py + h + 2, cols 2px .. 2px + 3
    const float *pl = a.a1 + (2 * py + h) * kAeX8A1Stride + 2 * px;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float2 lo = *reinterpret_cast<const float2 *>(pl + r * kAeX8A1Stride);
        const float2 hi = *reinterpret_cast<const float2 *>(pl + r * kAeX8A1Stride + 2);
        t[r][0] = lo.x; t[r][1] = lo.y; t[r][2] = hi.x; t[r][3] = hi.y;
    }
    float df[2][4];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {                  // block row r = 2 h + rr: r & 1 = rr, r >> 1 = h (t[] starts at row h)
        const uint32_t bits = (uint32_t)(rows1[4 * py + 2 * h + rr] >> (4 * px)) & 0xFu;
#pragma unroll
        for (int cx = 0; cx < 4; ++cx) {
            const int mx = cx >> 1;
            const int ry0 = 1, ky0 = rr ? 2 : 1;
            const int ry1 = rr ? 2 : 0, ky1 = rr ? 0 : 3;
            const int cx0 = mx + 1, kx0 = (cx & 1) ? 2 : 1;
            const int cx1 = (cx & 1) ? mx + 2 : mx, kx1 = (cx & 1) ? 0 : 3;
            float acc = 0.0f;
            acc = fmaf(t[ry0][cx0], wt2[ky0 * 4 + kx0], acc);
            acc = fmaf(t[ry0][cx1], wt2[ky0 * 4 + kx1], acc);
            acc = fmaf(t[ry1][cx0], wt2[ky1 * 4 + kx0], acc);
            acc = fmaf(t[ry1][cx1], wt2[ky1 * 4 + kx1], acc);
            const float pr = ae_x8_sigmoid(acc + bt2);
            df[rr][cx] = (float)((bits >> cx) & 1u) - pr;
        }
    }
    float sq = 0.0f;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int cx = 0; cx < 4; ++cx) sq = fmaf(df[rr][cx], df[rr][cx], sq);
    float sq2 = dpp_quad_xor1(sq);                    // lane h = 1: the chain of the block's first two rows
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int cx = 0; cx < 4; ++cx) sq2 = fmaf(df[rr][cx], df[rr][cx], sq2);
    if (h) a.tsum[win] = sq2;
}

// bonus = mean of the 4096 squared errors, summed as ae_block64_kernel sums them: per group of 64 blocks (its
// wavefront) the xor tree of wave_sum_f32, then (r0 + r1) + (r2 + r3).  All 64 lanes of ONE wavefront; wave-uniform result.
__device__ __forceinline__ float ae_x8_bonus(const AeBlockSharedX8 &a, int lane) {
    float r[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) r[w] = a.tsum[64 * w + lane];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) r[w] += __shfl_xor(r[w], sft, 64);
    }
    return ((r[0] + r[1]) + (r[2] + r[3])) * (1.0f / 4096.0f);
}

}  // namespace carle

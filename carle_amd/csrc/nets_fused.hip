// nets_fused.hip -- RND2D forward for 64x64 BINARY grids, fused into one kernel per batch.
//
// Same networks and numerics as nets_staged.hip (carle/mcl.py:262-296, 343-352; eval mode), but
// nothing leaves the CU between the observation read and the 4-byte bonus write:
//   * the observation is binary, so conv1 (3x3, 1->4 and 1->2 channels) + bias + relu is a
//     512-entry table per channel, indexed by the 9-bit neighbourhood pattern.  The tables are
//     built once per call by rnd_prepare_kernel (1.5k patterns x 9 fma) and sit in LDS;
//   * one workgroup per grid: the grid sits in LDS as 64 row words, thread = one 4x4 pooling
//     window reads its 6x6 neighbourhood from them (zero padding = two zero rows + shifts);
//   * pooled conv1 maps (6 x 16x16) go to a zero-bordered LDS image; conv2 + relu + 2x2 pool runs
//     from LDS on all four waves, the 64->16 linear layers + tanh and the bonus on one;
//   * HBM traffic per grid: 16 KiB in (fp32 obs) or 512 B (bit-packed obs), 4 B out.
// Workgroups are persistent: they stride over grids, so the 21 KiB table load is amortised, and 4
// of them per CU (33 KB of LDS each) are in different phases.  The training kernels further down
// (forward + backward) use the same workgroup-per-grid decomposition.
#include "rnd_device.hpp"

namespace carle {

// ---------------------------------------------------------------------------------------------
// per-call table build: conv1 LUTs (relu applied) + padded copy of the linear layers
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void rnd_prepare_kernel(const float *__restrict__ pp,
                                                          const float *__restrict__ rp,
                                                          float *__restrict__ ws) {
    const int p = threadIdx.x;                                  // neighbourhood pattern, bit ky*3+kx
    float *lut_p = ws, *lut_r = ws + 2048;
#pragma unroll
    for (int ch = 0; ch < 6; ++ch) {
        const float *wt = ch < 4 ? pp + ch * 9 : rp + (ch - 4) * 9;
        const float bias = ch < 4 ? pp[36 + ch] : rp[18 + (ch - 4)];
        float acc = 0.0f;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc = fmaf(wt[t], (float)((p >> t) & 1), acc);
        const float v = fmaxf(acc + bias, 0.0f);
        if (ch < 4) lut_p[p * 4 + ch] = v; else lut_r[p * 2 + (ch - 4)] = v;
    }
    float *lin = ws + kLutFloats;
    for (int i = p; i < 2 * kLinFloats; i += 512) {
        const int net = i / kLinFloats, j = i % kLinFloats;
        const float *src = (net == 0 ? pp + 77 : rp + 39);      // lin.w[16][64] then lin.b[16]
        float v;
        if (j < 16 * kLinStride) {
            const int k = j / kLinStride, d = j % kLinStride;
            v = d < 64 ? src[k * 64 + d] : 0.0f;
        } else {
            v = src[16 * 64 + (j - 16 * kLinStride)];
        }
        lin[i] = v;
    }
    // conv2 parameters behind the tables: pred w[36] b[1] at +0, rand w[18] b[1] at +40
    float *c2 = ws + kFusedWsFloats;
    if (p < 37) c2[p] = pp[40 + p];
    if (p >= 40 && p < 59) c2[p] = rp[20 + (p - 40)];
    if (p < 128) ws[kWsZeroGrid + p] = 0.0f;                    // the all-dead grid, bit-packed
}

// One workgroup (4 waves) per grid, persistent over grids; phases separated by barriers:
//   load   fp32: every lane loads 4 quads (rows 16w+4i+q, columns 4k..4k+3; the quads of the
//          workgroup's NEXT grid are requested before the conv1 phase), the 16 lanes of a row OR their
//          nibbles into the row's 64-bit word in LDS.  bit-packed: the 64 row words are copied.
//   conv1  thread = one 4x4 pooling window (rnd_block_conv1)
//   tail   all threads: conv2 + pooling (rnd_block_conv2); wave 0: linear layers, tanh, bonus
// 33 KB of LDS -> 4 workgroups (16 waves) per CU in different phases.
template <bool PACKED_OBS>
__global__ __launch_bounds__(256, 4) void rnd_block64_kernel(const void *__restrict__ obs_v,
                                                             const float *__restrict__ ws,
                                                             float *__restrict__ bonus,
                                                             float *__restrict__ pred_out,
                                                             float *__restrict__ target_out,
                                                             float *__restrict__ reward, float reward_scale,
                                                             int64_t n) {
    __shared__ RndBlockShared s;
    __shared__ float s_feat[2][64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    rnd_block_init(s, ws, tid);
    __syncthreads();

    const int q = lane >> 4, k = lane & 15;
    const int64_t first = blockIdx.x, stride = gridDim.x;
    uint4 v[4];
    uint64_t word = 0;
    auto issue = [&](int64_t g) {
        if (PACKED_OBS) {
            if (tid < 64) word = reinterpret_cast<const uint64_t *>(obs_v)[g * 64 + tid];
        } else {
            const uint4 *gin = reinterpret_cast<const uint4 *>(obs_v) + g * 1024 + 256 * wv + lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = gin[i * 64];
        }
    };
    if (first < n) issue(first);
    for (int64_t g = first; g < n; g += stride) {
        if (PACKED_OBS) {
            if (tid < 64) s.rows[1 + tid] = word;
        } else {
            uint32_t *rows32 = reinterpret_cast<uint32_t *>(&s.rows[1]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t nib = nz_bit(v[i].x) | (nz_bit(v[i].y) << 1) | (nz_bit(v[i].z) << 2) |
                                     (nz_bit(v[i].w) << 3);
                const uint32_t x = row_nibbles_to_dword(nib, k);
                if ((k & 7) == 7) rows32[2 * (16 * wv + 4 * i + q) + (k >> 3)] = x;
            }
        }
        if (g + stride < n) issue(g + stride);        // the next grid travels during conv1 + tail
        __syncthreads();
        rnd_block_conv1(s, tid);
        __syncthreads();
        rnd_block_conv2(s, s_feat, tid);
        __syncthreads();
        if (wv == 0) {
            const float b = rnd_linear_bonus(s_feat, s.lin, lane, g, pred_out, target_out);
            if (lane == 0) {
                bonus[g] = b;
                if (reward) reward[g] += reward_scale * b;       // mcl.py:403-405
            }
        }
        // rows are rewritten at the top of the next iteration: every wave has passed the barrier
        // behind conv1 (their last reader); the planes only behind the next iteration's first barrier
    }
}

}  // namespace carle

using namespace carle;

namespace carle {

// workspace: [tables + padded linear layers (kFusedWsFloats)] [conv2 parameters (64)] [all-dead
// packed grid (128)] [its bonus (4)]
int64_t rnd2d_fused_workspace_floats() { return kFusedWsTotal; }

// Tables, and the bonus of the all-dead grid (the observation of every instance after a master
// reset, env.py:208-216): it depends on the parameters only, so it is evaluated here, once per
// parameter change, by the ordinary kernel on a zero grid kept in the workspace.
void rnd2d_prepare_tables(const float *pp, const float *rp, float *ws, hipStream_t s) {
    hipLaunchKernelGGL(rnd_prepare_kernel, dim3(1), dim3(512), 0, s, pp, rp, ws);
    hipLaunchKernelGGL(rnd_block64_kernel<true>, dim3(1), dim3(256), 0, s, static_cast<const void *>(ws + kWsZeroGrid),
                       static_cast<const float *>(ws), ws + kWsDeadBonus, static_cast<float *>(nullptr),
                       static_cast<float *>(nullptr), static_cast<float *>(nullptr), 0.0f, (int64_t)1);
}

int rnd2d_fused64(const void *obs, int obs_mode, int64_t n, const float *pp, const float *rp,
                  float *ws, float *bonus, float *pred_out, float *target_out, float *reward,
                  float reward_scale, bool tables_valid, hipStream_t s) {
    if (!tables_valid) rnd2d_prepare_tables(pp, rp, ws, s);
    int64_t blocks = n < 1024 ? n : 1024;             // persistent: 4 workgroups are resident per CU
    if (blocks < 1) blocks = 1;
    if (obs_mode == 1)
        hipLaunchKernelGGL(rnd_block64_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, obs, ws,
                           bonus, pred_out, target_out, reward, reward_scale, n);
    else
        hipLaunchKernelGGL(rnd_block64_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, obs, ws,
                           bonus, pred_out, target_out, reward, reward_scale, n);
    return check_launch("carle_rnd2d_bonus(fused 64x64)");
}

}  // namespace carle

// =============================================================================================
// AE2D forward for 64x64 binary grids, fused (carle/mcl.py:452-467, 504-512; eval mode)
// =============================================================================================
// Same decomposition as the staged kernels (conv1+relu+pool2 -> conv2+relu+pool2 -> convT1+relu
// -> convT2+sigmoid -> mean squared error) but per workgroup and LDS-resident:
//   * the grid sits in LDS as 64 row words; conv1 is the 512-entry table (4 channels = one float4
//     per neighbourhood pattern);
//   * A1 (4 x 32x32), A2 (2 x 16x16) and A3 (32x32) live in zero-bordered LDS planes; A3 overlays
//     A1.  Row strides of 36 / 20 floats spread the 8-byte patch reads over the banks;
//   * every thread finally reconstructs its own 4x4 cells and accumulates (obs - sigmoid)^2.
namespace carle {

static constexpr int kAeLutFloats = 512 * 4;
static constexpr int kAeParamFloats = 128;                    // conv2, convT1, convT2 parameters
static constexpr int kAeWsFloats = kAeLutFloats + kAeParamFloats;
static constexpr int kA1Stride = 36, kA1Rows = 34, kA1Plane = kA1Rows * kA1Stride;   // also A3
static constexpr int kA2Stride = 20, kA2Rows = 18, kA2Plane = kA2Rows * kA2Stride;

__global__ __launch_bounds__(512) void ae_prepare_kernel(const float *__restrict__ ap,
                                                         float *__restrict__ ws) {
    const int p = threadIdx.x;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
        float acc = 0.0f;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc = fmaf(ap[ch * 9 + t], (float)((p >> t) & 1), acc);
        ws[p * 4 + ch] = fmaxf(acc + ap[36 + ch], 0.0f);
    }
    // conv2.w[72] b[2] | convT1.w[32] b[1] | convT2.w[16] b[1]  (= ap[40..164))
    if (p < 124) ws[kAeLutFloats + p] = ap[40 + p];
}

__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
    return v;
}

// One workgroup (4 waves) per grid, persistent over grids; thread (py = tid/16, px = tid%16) owns the
// 4x4 cell block rows 4py.., cols 4px.. in every layer (its 2x2 outputs of A1 / A3, its one output
// of A2), phases separated by barriers.  31 KB of LDS -> 4-5 workgroups (16-20 waves) per CU.
struct AeBlockShared {
    float4 lut[512];
    float a1[4 * kA1Plane];          // A1; A3 later overlays plane 0
    float a2[2 * kA2Plane];
    uint64_t rows[66];               // the grid, one 64-bit word per row at [1..64]; [0], [65] stay zero
    float red[4];
};

template <bool PACKED_OBS>
__global__ __launch_bounds__(256, 4) void ae_block64_kernel(
    const void *__restrict__ obs_v, const float *__restrict__ ws, float *__restrict__ bonus,
    float *__restrict__ recon, float *__restrict__ reward, float reward_scale, int64_t n) {
    __shared__ AeBlockShared s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    {
        const float4 *src = reinterpret_cast<const float4 *>(ws);
        for (int i = tid; i < 512; i += 256) s.lut[i] = src[i];
        for (int i = tid; i < 4 * kA1Plane; i += 256) s.a1[i] = 0.0f;
        for (int i = tid; i < 2 * kA2Plane; i += 256) s.a2[i] = 0.0f;
        if (tid < 66) s.rows[tid] = 0ull;
    }
    __syncthreads();
    const float *prm = ws + kAeLutFloats;            // uniform parameter block
    const float *w2 = prm, *b2 = prm + 72, *wt1 = prm + 74, *wt2 = prm + 107;
    const float bt1 = prm[106], bt2 = prm[123];

    const int q = lane >> 4, k = lane & 15;          // load mapping: rows 16w+4i+q, columns 4k..4k+3
    const int py = tid >> 4, px = tid & 15;          // compute mapping
    const int64_t first = blockIdx.x, stride = gridDim.x;
    uint4 v[4];
    uint64_t word = 0;
    auto issue = [&](int64_t g) {
        if (PACKED_OBS) {
            if (tid < 64) word = reinterpret_cast<const uint64_t *>(obs_v)[g * 64 + tid];
        } else {
            const uint4 *gin = reinterpret_cast<const uint4 *>(obs_v) + g * 1024 + 256 * wv + lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = gin[i * 64];
        }
    };
    if (first < n) issue(first);
    for (int64_t g = first; g < n; g += stride) {
        // ---- the grid as 64 row words in LDS
        if (PACKED_OBS) {
            if (tid < 64) s.rows[1 + tid] = word;
        } else {
            uint32_t *rows32 = reinterpret_cast<uint32_t *>(&s.rows[1]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t nib = nz_bit(v[i].x) | (nz_bit(v[i].y) << 1) | (nz_bit(v[i].z) << 2) |
                                     (nz_bit(v[i].w) << 3);
                const uint32_t x = row_nibbles_to_dword(nib, k);
                if ((k & 7) == 7) rows32[2 * (16 * wv + 4 * i + q) + (k >> 3)] = x;
            }
        }
        if (g + stride < n) issue(g + stride);        // the next grid travels during the layers
        __syncthreads();

        // ---- conv1 (table) + relu + 2x2 pool -> A1 (2py+a, 2px+b); zero padding at the border
        uint32_t seg[6];                              // [west | 4 cells | east] of rows 4py-1..4py+4
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const uint64_t row = s.rows[4 * py + r];
            seg[r] = (uint32_t)(px == 0 ? row << 1 : row >> (4 * px - 1)) & 0x3Fu;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        const int r = 2 * a + dy, cx = 2 * b + dx;
                        const uint32_t p = ((seg[r] >> cx) & 7u) | (((seg[r + 1] >> cx) & 7u) << 3) |
                                           (((seg[r + 2] >> cx) & 7u) << 6);
                        const float4 e = s.lut[p];
                        m0 = fmaxf(m0, e.x); m1 = fmaxf(m1, e.y); m2 = fmaxf(m2, e.z); m3 = fmaxf(m3, e.w);
                    }
                const int off = (2 * py + a + 1) * kA1Stride + 2 * px + b + 1;
                s.a1[0 * kA1Plane + off] = m0;
                s.a1[1 * kA1Plane + off] = m1;
                s.a1[2 * kA1Plane + off] = m2;
                s.a1[3 * kA1Plane + off] = m3;
            }
        __syncthreads();

        // ---- conv2 (4->2, 3x3, zero pad) + relu + 2x2 pool -> A2 (py, px)
        {
            float acc2[2][2][2];
#pragma unroll
            for (int oc = 0; oc < 2; ++oc) { acc2[oc][0][0] = 0.f; acc2[oc][0][1] = 0.f; acc2[oc][1][0] = 0.f; acc2[oc][1][1] = 0.f; }
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const float *pl = s.a1 + cc * kA1Plane + (2 * py) * kA1Stride + 2 * px;   // padded rows 2py.., cols 2px..
                float patch[4][4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float2 lo = *reinterpret_cast<const float2 *>(pl + r * kA1Stride);
                    const float2 hi = *reinterpret_cast<const float2 *>(pl + r * kA1Stride + 2);
                    patch[r][0] = lo.x; patch[r][1] = lo.y; patch[r][2] = hi.x; patch[r][3] = hi.y;
                }
#pragma unroll
                for (int oc = 0; oc < 2; ++oc)
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 2; ++dx)
#pragma unroll
                            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                                for (int kx = 0; kx < 3; ++kx)
                                    acc2[oc][dy][dx] = fmaf(w2[(oc * 4 + cc) * 9 + ky * 3 + kx],
                                                            patch[dy + ky][dx + kx], acc2[oc][dy][dx]);
            }
#pragma unroll
            for (int oc = 0; oc < 2; ++oc) {
                const float mx = fmaxf(fmaxf(acc2[oc][0][0], acc2[oc][0][1]),
                                       fmaxf(acc2[oc][1][0], acc2[oc][1][1])) + b2[oc];
                s.a2[oc * kA2Plane + (py + 1) * kA2Stride + px + 1] = fmaxf(mx, 0.0f);
            }
        }
        __syncthreads();                              // A1 reads done: A3 may overlay plane 0

        // ---- convT1 (2->1, k4 s2 p1) + relu -> A3 (2py+j, 2px+i), overlaying A1 plane 0
        {
            float r2[2][3][3];                        // A2 rows py-1..py+1, cols px-1..px+1 (padded)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int x = 0; x < 3; ++x)
                        r2[cc][r][x] = s.a2[cc * kA2Plane + (py + r) * kA2Stride + px + x];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    // output row y = 2py+j: y even: (iy=py, ky=1), (iy=py-1, ky=3);
                    //                       y odd:  (iy=py, ky=2), (iy=py+1, ky=0)
                    const int ry0 = 1, ky0 = j ? 2 : 1;
                    const int ry1 = j ? 2 : 0, ky1 = j ? 0 : 3;
                    const int cx0 = 1, kx0 = i ? 2 : 1;
                    const int cx1 = i ? 2 : 0, kx1 = i ? 0 : 3;
                    float acc = 0.0f;
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc) {
                        acc = fmaf(r2[cc][ry0][cx0], wt1[cc * 16 + ky0 * 4 + kx0], acc);
                        acc = fmaf(r2[cc][ry0][cx1], wt1[cc * 16 + ky0 * 4 + kx1], acc);
                        acc = fmaf(r2[cc][ry1][cx0], wt1[cc * 16 + ky1 * 4 + kx0], acc);
                        acc = fmaf(r2[cc][ry1][cx1], wt1[cc * 16 + ky1 * 4 + kx1], acc);
                    }
                    s.a1[(2 * py + j + 1) * kA1Stride + 2 * px + i + 1] = fmaxf(acc + bt1, 0.0f);
                }
        }
        __syncthreads();

        // ---- convT2 (1->1) + sigmoid + squared error on the thread's own 4x4 cells
        float sq = 0.0f;
        {
            float t3[4][4];                           // A3 rows 2py-1..2py+2, cols 2px-1..2px+2 (padded)
            const float *pl = s.a1 + (2 * py) * kA1Stride + 2 * px;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float2 lo = *reinterpret_cast<const float2 *>(pl + r * kA1Stride);
                const float2 hi = *reinterpret_cast<const float2 *>(pl + r * kA1Stride + 2);
                t3[r][0] = lo.x; t3[r][1] = lo.y; t3[r][2] = hi.x; t3[r][3] = hi.y;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t bits = (uint32_t)(s.rows[1 + 4 * py + r] >> (4 * px)) & 0xFu;
                float out4[4];
#pragma unroll
                for (int cx = 0; cx < 4; ++cx) {
                    const int my = r >> 1, mx = cx >> 1;              // local A3 index of iy=m is m+1
                    const int ry0 = my + 1, ky0 = (r & 1) ? 2 : 1;
                    const int ry1 = (r & 1) ? my + 2 : my, ky1 = (r & 1) ? 0 : 3;
                    const int cx0 = mx + 1, kx0 = (cx & 1) ? 2 : 1;
                    const int cx1 = (cx & 1) ? mx + 2 : mx, kx1 = (cx & 1) ? 0 : 3;
                    float acc = 0.0f;
                    acc = fmaf(t3[ry0][cx0], wt2[ky0 * 4 + kx0], acc);
                    acc = fmaf(t3[ry0][cx1], wt2[ky0 * 4 + kx1], acc);
                    acc = fmaf(t3[ry1][cx0], wt2[ky1 * 4 + kx0], acc);
                    acc = fmaf(t3[ry1][cx1], wt2[ky1 * 4 + kx1], acc);
                    const float pr = 1.0f / (1.0f + expf(-(acc + bt2)));
                    out4[cx] = pr;
                    const float df = (float)((bits >> cx) & 1u) - pr;
                    sq = fmaf(df, df, sq);
                }
                if (recon)
                    *reinterpret_cast<float4 *>(recon + g * 4096 + (4 * py + r) * 64 + 4 * px) =
                        make_float4(out4[0], out4[1], out4[2], out4[3]);
            }
        }
        sq = wave_sum_f32(sq);
        if (lane == 0) s.red[wv] = sq;
        __syncthreads();                              // also: rows / A3 reads done before the next grid
        if (tid == 0) {
            const float b = ((s.red[0] + s.red[1]) + (s.red[2] + s.red[3])) * (1.0f / 4096.0f);
            bonus[g] = b;
            if (reward) reward[g] += reward_scale * b;
        }
    }
}

int64_t ae2d_fused_workspace_floats() { return kAeWsFloats; }

int ae2d_fused64(const void *obs, bool packed, int64_t n, const float *ap, float *ws, float *bonus,
                 float *recon, float *reward, float reward_scale, bool tables_valid, hipStream_t s) {
    if (!tables_valid) hipLaunchKernelGGL(ae_prepare_kernel, dim3(1), dim3(512), 0, s, ap, ws);
    int64_t blocks = n < 1280 ? n : 1280;             // persistent: 5 workgroups are resident per CU
    if (blocks < 1) blocks = 1;
    if (packed)
        hipLaunchKernelGGL(ae_block64_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, obs, ws, bonus,
                           recon, reward, reward_scale, n);
    else
        hipLaunchKernelGGL(ae_block64_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, obs, ws, bonus,
                           recon, reward, reward_scale, n);
    return check_launch("carle_ae2d_bonus(fused 64x64)");
}

}  // namespace carle

// =============================================================================================
// RND2D forward + BACKWARD for 64x64 binary grids: the predictor's gradient
// of  sum_t mean_n mean_k (target - prediction)^2 / batch_size  (carle/mcl.py:343-386), computed
// in the same kernel that evaluates the bonus -- no observation window to retain, no autograd
// replay.  Eval-mode semantics (Dropout = identity).  Per grid:
//   forward as rnd_block64_kernel, additionally remembering, per 4x4 pooling window and predictor
//   channel, the 9-bit input pattern of the arg-max pixel (ties only occur between equal
//   patterns or at the relu floor, where the routing does not matter);
//   backward: dtanh -> linear -> arg-max of the 2x2 pool and relu gate -> conv2 -> relu gate at
//   the arg-max pixel of each 4x4 window -> conv1 weight gradients as gated sums of the
//   remembered pattern bits.
// =============================================================================================
namespace carle {

static constexpr int kRndPredParams = 36 + 4 + 36 + 1 + 16 * 64 + 16;   // 1117

// One WORKGROUP per grid (the decomposition of rnd_block64_kernel): thread (py,px) owns
// pooling window (py,px) in the forward conv1 AND in the backward pass through it, so the arg-max
// patterns stay in its registers; the gradient w.r.t. the conv2 outputs goes through a sparse
// 16x16 map in LDS (one entry per 2x2 pooling block) and is GATHERED (no LDS atomics):
//   d plane[c](P)      = sum_k  w2[c][k] * map(P - k + 1)
//   d w2[c][k]        += plane[c](P) * map(P - k + 1)        (the same nine map reads)
// conv1 / conv2 weight gradients are channel-split (wave c walks the 256 windows of channel c, 4 per
// lane): accumulators per thread 9+1 (conv1) + 9 (conv2) + its 4 entries of the 16x64 linear layer.
// One partial row per workgroup at the end; rnd_grad_reduce_kernel adds the rows in a fixed order
// (deterministic).  2 workgroups (8 waves) per CU; the first version (one WAVE per grid, all 1117
// accumulators spread over 64 lanes, ds_add_f32 scatters, 4 waves per CU) took twice as long.
template <bool PACKED_OBS>
__global__ __launch_bounds__(256, 3) void rnd_train_block64_kernel(
    const void *__restrict__ obs_v, const float *__restrict__ ws, float *__restrict__ bonus,
    float *__restrict__ reward, float reward_scale, float grad_scale, float *__restrict__ partial,
    int64_t n) {
    __shared__ RndBlockShared s;
    __shared__ float s_map[kPlaneFloats];            // d loss / d conv2 output, zero-bordered 16x16
    __shared__ float s_feat[2][64];
    __shared__ float s_dy[16];
    __shared__ uint16_t s_pat[4][256];               // arg-max input pattern per predictor channel and window
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    rnd_block_init(s, ws, tid);
    for (int i = tid; i < kPlaneFloats; i += 256) s_map[i] = 0.0f;
    __syncthreads();
    const float *cw2p = ws + kFusedWsFloats;          // predictor conv2.w[36], b at [36]
    const float *cw2r = ws + kFusedWsFloats + 40;     // random conv2.w[18], b at [18]

    const int q = lane >> 4, k = lane & 15;           // load mapping
    const int py = tid >> 4, px = tid & 15;           // window mapping
    const int fy = lane >> 3, fx = lane & 7;          // wave 0: pooled conv2 output
    const int half = lane & 1, kk = (lane >> 1) & 15, net = lane >> 5;
    const int wk = tid >> 4, wj = 4 * (tid & 15);     // this thread's entries W[wk][wj..wj+3] of lin.w

    // conv1 / conv2 gradients are CHANNEL-SPLIT: wave c accumulates those of predictor channel c
    float gw1c[9], gb1c = 0.0f, gw2c[9], gW[4], gB = 0.0f, gb2 = 0.0f;
#pragma unroll
    for (int i = 0; i < 9; ++i) { gw1c[i] = 0.0f; gw2c[i] = 0.0f; }
#pragma unroll
    for (int i = 0; i < 4; ++i) gW[i] = 0.0f;
    const int wvs = __builtin_amdgcn_readfirstlane(wv);
    float wc[9];                                      // conv2.w of input channel wv (scalar registers)
#pragma unroll
    for (int i = 0; i < 9; ++i) wc[i] = cw2p[wvs * 9 + i];

    const int64_t first = blockIdx.x, stride = gridDim.x;
    uint4 v[4];
    uint64_t word = 0;
    auto issue = [&](int64_t g) {
        if (PACKED_OBS) {
            if (tid < 64) word = reinterpret_cast<const uint64_t *>(obs_v)[g * 64 + tid];
        } else {
            const uint4 *gin = reinterpret_cast<const uint4 *>(obs_v) + g * 1024 + 256 * wv + lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = gin[i * 64];
        }
    };
    if (first < n) issue(first);
    for (int64_t g = first; g < n; g += stride) {
        // ---- the grid as 64 row words in LDS
        if (PACKED_OBS) {
            if (tid < 64) s.rows[1 + tid] = word;
        } else {
            uint32_t *rows32 = reinterpret_cast<uint32_t *>(&s.rows[1]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t nib = nz_bit(v[i].x) | (nz_bit(v[i].y) << 1) | (nz_bit(v[i].z) << 2) |
                                     (nz_bit(v[i].w) << 3);
                const uint32_t x = row_nibbles_to_dword(nib, k);
                if ((k & 7) == 7) rows32[2 * (16 * wv + 4 * i + q) + (k >> 3)] = x;
            }
        }
        if (g + stride < n) issue(g + stride);
        __syncthreads();

        // ---- conv1 + relu + 4x4 pool of window (py,px), remembering the arg-max pattern per
        //      predictor channel
        {
            uint32_t seg[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const uint64_t row = s.rows[4 * py + r];
                seg[r] = (uint32_t)(px == 0 ? row << 1 : row >> (4 * px - 1)) & 0x3Fu;
            }
            float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f, m5 = 0.f;
            uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
            for (int dy = 0; dy < 4; ++dy)
#pragma unroll
                for (int dx = 0; dx < 4; ++dx) {
                    const uint32_t p = ((seg[dy] >> dx) & 7u) | (((seg[dy + 1] >> dx) & 7u) << 3) |
                                       (((seg[dy + 2] >> dx) & 7u) << 6);
                    const float4 a = s.lut_p[p];
                    const float2 b = s.lut_r[p];
                    if (a.x > m0) { m0 = a.x; a0 = p; }
                    if (a.y > m1) { m1 = a.y; a1 = p; }
                    if (a.z > m2) { m2 = a.z; a2 = p; }
                    if (a.w > m3) { m3 = a.w; a3 = p; }
                    m4 = fmaxf(m4, b.x); m5 = fmaxf(m5, b.y);
                }
            s_pat[0][tid] = (uint16_t)a0;
            s_pat[1][tid] = (uint16_t)a1;
            s_pat[2][tid] = (uint16_t)a2;
            s_pat[3][tid] = (uint16_t)a3;
            const int off = (py + 1) * kPlaneStride + px + 1;
            s.planes[0][off] = m0;
            s.planes[1][off] = m1;
            s.planes[2][off] = m2;
            s.planes[3][off] = m3;
            s.planes[4][off] = m4;
            s.planes[5][off] = m5;
        }
        __syncthreads();

        // ---- wave 0: conv2 + relu + 2x2 pool (both nets), linear + tanh, bonus, dL/dy
        int amax = 0;
        bool active = false;
        if (wv == 0) {
            const float *planes = &s.planes[0][0];
#pragma unroll
            for (int nn = 0; nn < 2; ++nn) {
                const int nch = nn == 0 ? 4 : 2, chbase = nn == 0 ? 0 : 4;
                const float *cw2 = nn == 0 ? cw2p : cw2r;
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
                float mx = o00; int am = 0;
                if (o01 > mx) { mx = o01; am = 1; }
                if (o10 > mx) { mx = o10; am = 2; }
                if (o11 > mx) { mx = o11; am = 3; }
                mx += cw2[nn == 0 ? 36 : 18];
                s_feat[nn][fy * 8 + fx] = fmaxf(mx, 0.0f);
                if (nn == 0) { amax = am; active = mx > 0.0f; }
            }
            __builtin_amdgcn_wave_barrier();
            const float4 *f4 = reinterpret_cast<const float4 *>(&s_feat[net][32 * half]);
            const float4 *w4 = reinterpret_cast<const float4 *>(&s.lin[net * kLinFloats + kk * kLinStride + 32 * half]);
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4 f = f4[i], w = w4[i];
                acc = fmaf(w.x, f.x, acc); acc = fmaf(w.y, f.y, acc);
                acc = fmaf(w.z, f.z, acc); acc = fmaf(w.w, f.w, acc);
            }
            acc += __shfl_xor(acc, 1, 64);
            const float out = tanhf(acc + s.lin[net * kLinFloats + 16 * kLinStride + kk]);
            const float other = __shfl_xor(out, 32, 64);
            const float diff = out - other;                      // net 0 lanes: prediction - target
            float sq = diff * diff;
#pragma unroll
            for (int sh = 2; sh < 32; sh <<= 1) sq += __shfl_xor(sq, sh, 64);
            if (lane == 0) {
                const float b = sq * (1.0f / 16.0f);
                bonus[g] = b;
                if (reward) reward[g] += reward_scale * b;
            }
            if (net == 0 && half == 0) s_dy[kk] = grad_scale * diff * (1.0f - out * out);   // d loss / d y_k
        }
        __syncthreads();

        // ---- linear layer gradients (every thread owns 4 weights); wave 0: back through the
        //      linear layer, the 2x2 pool and relu -> the conv2-output gradient map
        {
            const float dyk = s_dy[wk];
#pragma unroll
            for (int i = 0; i < 4; ++i) gW[i] = fmaf(dyk, s_feat[0][wj + i], gW[i]);
            if (tid < 16) gB += s_dy[tid];
        }
        if (wv == 0) {
            float dp2 = 0.0f;
#pragma unroll
            for (int o = 0; o < 16; ++o) dp2 = fmaf(s.lin[o * kLinStride + lane], s_dy[o], dp2);
            const float dz2 = active ? dp2 : 0.0f;
            gb2 += dz2;
            // the four conv2 outputs of this pooling block: gradient at the arg-max, zero elsewhere
#pragma unroll
            for (int e = 0; e < 4; ++e)
                s_map[(2 * fy + (e >> 1) + 1) * kPlaneStride + 2 * fx + (e & 1) + 1] = e == amax ? dz2 : 0.0f;
        }
        __syncthreads();

        // ---- back through conv2 (gather), the 4x4 pool, relu and conv1 of channel wv: lane = 4 windows
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int wdx = lane + 64 * j, wy = wdx >> 4, wx = wdx & 15;
            float m[3][3];                            // map rows wy..wy+2, cols wx..wx+2 (padded coords)
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) m[r][c] = s_map[(wy + r) * kPlaneStride + wx + c];
            const float pmv = s.planes[wvs][(wy + 1) * kPlaneStride + wx + 1];
            const uint32_t pt = s_pat[wvs][wdx];
            float d = 0.0f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    d = fmaf(wc[ky * 3 + kx], m[2 - ky][2 - kx], d);
                    gw2c[ky * 3 + kx] = fmaf(pmv, m[2 - ky][2 - kx], gw2c[ky * 3 + kx]);
                }
            const float gz = pmv > 0.0f ? d : 0.0f;
            gb1c += gz;
#pragma unroll
            for (int tt = 0; tt < 9; ++tt) gw1c[tt] = fmaf(gz, (float)((pt >> tt) & 1u), gw1c[tt]);
        }
        // the next iteration's barriers order its writes to rows / planes / patterns / the map behind
        // these reads
    }

    // ---- one partial gradient row per workgroup, parameter order of the predictor blob
    float *row = partial + (int64_t)blockIdx.x * kRndPredParams;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        float a = gw1c[i], b = gw2c[i];
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1) { a += __shfl_xor(a, sh, 64); b += __shfl_xor(b, sh, 64); }
        if (lane == 0) { row[wvs * 9 + i] = a; row[40 + wvs * 9 + i] = b; }
    }
    {
        float a = gb1c, b = gb2;                      // gb2 is non-zero on wave 0 only
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1) { a += __shfl_xor(a, sh, 64); b += __shfl_xor(b, sh, 64); }
        if (lane == 0) {
            row[36 + wvs] = a;
            if (wvs == 0) row[76] = b;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) row[77 + wk * 64 + wj + i] = gW[i];
    if (tid < 16) row[77 + 1024 + tid] = gB;
}

// grad[p] += sum over the partial rows, fixed order: 32 parameters x 32 row groups per workgroup
// (128-byte coalesced row segments, four independent loads in flight per thread), the 32 group sums
// combined in a fixed order -> deterministic.  (8 row groups per workgroup took 14-20 us for 768 rows:
// a third of the RND2D forward+backward kernel.)
__global__ __launch_bounds__(1024) void rnd_grad_reduce_kernel(const float *__restrict__ partial, int rows,
                                                               float *__restrict__ grad, int nparams) {
    __shared__ float part[32][33];
    const int pl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int p = blockIdx.x * 32 + pl;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    if (p < nparams) {
        int r = rl;
        for (; r + 96 < rows; r += 128) {
            s0 += partial[(int64_t)r * nparams + p];
            s1 += partial[(int64_t)(r + 32) * nparams + p];
            s2 += partial[(int64_t)(r + 64) * nparams + p];
            s3 += partial[(int64_t)(r + 96) * nparams + p];
        }
        for (; r < rows; r += 32) s0 += partial[(int64_t)r * nparams + p];
    }
    part[rl][pl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rl == 0 && p < nparams) {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < 32; ++i) s += part[i][pl];
        grad[p] += s;
    }
}

int rnd2d_train64_rows(int64_t n) {                // = workgroups of rnd_train_block64_kernel
    int64_t blocks = n < 768 ? n : 768;              // persistent: 3 workgroups are resident per CU (a fourth needs <= 128 VGPRs: spills, slower)
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

int rnd2d_train64(const void *obs, int obs_mode, int64_t n, const float *pp, const float *rp, float *ws,
                  float *bonus, float *reward, float reward_scale, float grad_scale, float *partial,
                  float *grad, bool tables_valid, hipStream_t s) {
    if (!tables_valid) hipLaunchKernelGGL(rnd_prepare_kernel, dim3(1), dim3(512), 0, s, pp, rp, ws);
    const int rows = rnd2d_train64_rows(n);
    if (obs_mode == 1)
        hipLaunchKernelGGL(rnd_train_block64_kernel<true>, dim3((unsigned)rows), dim3(256), 0, s, obs, ws, bonus,
                           reward, reward_scale, grad_scale, partial, n);
    else
        hipLaunchKernelGGL(rnd_train_block64_kernel<false>, dim3((unsigned)rows), dim3(256), 0, s, obs, ws, bonus,
                           reward, reward_scale, grad_scale, partial, n);
    hipLaunchKernelGGL(rnd_grad_reduce_kernel, dim3((kRndPredParams + 31) / 32), dim3(1024), 0, s, partial,
                       rows, grad, kRndPredParams);
    return check_launch("carle_rnd2d_train(fused 64x64)");
}

}  // namespace carle

// =============================================================================================
// AE2D forward + BACKWARD for 64x64 binary grids: gradient of
//   sum_t mean_n mean_hw (obs - decoder(obs))^2 / batch_size   (carle/mcl.py:504-547)
// with respect to the 164 predictor parameters, in the pass that evaluates the bonus.
// =============================================================================================
namespace carle {

static constexpr int kAeParams = 164;

// ONE WORKGROUP per grid.  Forward as ae_block64_kernel (thread (py,px) owns the 4x4
// cell block at (4py,4px) in every layer), A3 in its own plane (A1 is needed again on the way
// back).  Backward:
//   convT2 / convT1: each thread accumulates the contributions of its own outputs to the 4x4
//     (3x3) input neighbourhood in registers, then ds_add_f32's them into the d3 / d2 planes;
//   conv2: the gradient w.r.t. its outputs goes to a sparse map (one entry per 2x2 pooling block);
//   conv2 + conv1 weights: CHANNEL-SPLIT -- wave c walks all 32x32 positions of A1 plane c (16 per
//     lane), gathers  d A1[c](P) = sum_{o,k} w2[o][c][k] map[o](P-k+1),  accumulates
//     d w2[o][c][k] += A1[c](P) map[o](P-k+1)  and, through the relu gate and the arg-max pattern
//     of P, the conv1 gradients of channel c: 18 + 9 + 1 accumulators per lane instead of 72 + 36 + 4.
// 80 accumulators per thread in all (the first version, one wave per grid: 164 per lane, 512
// registers, 1.6 KB of scratch, 3 waves per CU, twice the time); 62 KB of LDS -> 2 workgroups
// (8 waves) per CU.
struct AeTrainShared {
    float4 lut[512];
    float a1[4 * kA1Plane];
    float a2[2 * kA2Plane];
    float a3[kA1Plane];
    float d3[kA1Plane];              // d loss / d A3 (before its relu gate)
    float d2[2 * kA2Plane];          // d loss / d A2 (after the pool)
    float map2[2 * kA1Plane];        // d loss / d conv2 output: one non-zero per 2x2 pooling block
    uint16_t pat[4][1024];           // arg-max input pattern of every A1 position (32x32, row-major)
    uint64_t rows[66];
    float red[4][52];
};

template <bool PACKED_OBS>
__global__ __launch_bounds__(256, 2) void ae_train_block64_kernel(
    const void *__restrict__ obs_v, const float *__restrict__ ws, float *__restrict__ bonus,
    float *__restrict__ reward, float reward_scale, float grad_scale, float *__restrict__ partial,
    int64_t n) {
    __shared__ AeTrainShared s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    {
        const float4 *src = reinterpret_cast<const float4 *>(ws);
        for (int i = tid; i < 512; i += 256) s.lut[i] = src[i];
        for (int i = tid; i < 4 * kA1Plane; i += 256) s.a1[i] = 0.0f;
        for (int i = tid; i < 2 * kA2Plane; i += 256) { s.a2[i] = 0.0f; s.d2[i] = 0.0f; }
        for (int i = tid; i < kA1Plane; i += 256) { s.a3[i] = 0.0f; s.d3[i] = 0.0f; }
        for (int i = tid; i < 2 * kA1Plane; i += 256) s.map2[i] = 0.0f;
        if (tid < 66) s.rows[tid] = 0ull;
    }
    __syncthreads();
    const float *prm = ws + kAeLutFloats;            // uniform parameter block
    const float *w2 = prm, *b2 = prm + 72, *wt1 = prm + 74, *wt2 = prm + 107;
    const float bt1 = prm[106], bt2 = prm[123];

    const int q = lane >> 4, k = lane & 15;          // load mapping
    const int py = tid >> 4, px = tid & 15;          // cell-block mapping

    float gwt2[16], gbt2 = 0.0f, gwt1[32], gbt1 = 0.0f, gb2[2] = {0.0f, 0.0f};
    float gw2c[18], gw1c[9], gb1c = 0.0f;            // conv2 / conv1 gradients of input channel wv
#pragma unroll
    for (int i = 0; i < 16; ++i) gwt2[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; ++i) gwt1[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 18; ++i) gw2c[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 9; ++i) gw1c[i] = 0.0f;
    const int wvs = __builtin_amdgcn_readfirstlane(wv);   // provably wave-uniform: scalar loads below
    float wc[18];                                    // w2[oc][wv][k], oc-major (SGPRs)
#pragma unroll
    for (int oc = 0; oc < 2; ++oc)
#pragma unroll
        for (int kq = 0; kq < 9; ++kq) wc[oc * 9 + kq] = w2[(oc * 4 + wvs) * 9 + kq];

    const int64_t first = blockIdx.x, stride = gridDim.x;
    uint4 v[4];
    uint64_t word = 0;
    auto issue = [&](int64_t g) {
        if (PACKED_OBS) {
            if (tid < 64) word = reinterpret_cast<const uint64_t *>(obs_v)[g * 64 + tid];
        } else {
            const uint4 *gin = reinterpret_cast<const uint4 *>(obs_v) + g * 1024 + 256 * wv + lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = gin[i * 64];
        }
    };
    if (first < n) issue(first);
    for (int64_t g = first; g < n; g += stride) {
        // ---- the grid as 64 row words in LDS
        if (PACKED_OBS) {
            if (tid < 64) s.rows[1 + tid] = word;
        } else {
            uint32_t *rows32 = reinterpret_cast<uint32_t *>(&s.rows[1]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t nib = nz_bit(v[i].x) | (nz_bit(v[i].y) << 1) | (nz_bit(v[i].z) << 2) |
                                     (nz_bit(v[i].w) << 3);
                const uint32_t x = row_nibbles_to_dword(nib, k);
                if ((k & 7) == 7) rows32[2 * (16 * wv + 4 * i + q) + (k >> 3)] = x;
            }
        }
        if (g + stride < n) issue(g + stride);
        __syncthreads();

        // ---- conv1 (table) + relu + 2x2 pool -> A1 (2py+a, 2px+b) with arg-max patterns
        {
            uint32_t seg[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const uint64_t row = s.rows[4 * py + r];
                seg[r] = (uint32_t)(px == 0 ? row << 1 : row >> (4 * px - 1)) & 0x3Fu;
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
                    uint32_t p0 = 0, p1 = 0, p2 = 0, p3 = 0;
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 2; ++dx) {
                            const int r = 2 * a + dy, cx = 2 * b + dx;
                            const uint32_t p = ((seg[r] >> cx) & 7u) | (((seg[r + 1] >> cx) & 7u) << 3) |
                                               (((seg[r + 2] >> cx) & 7u) << 6);
                            const float4 e = s.lut[p];
                            if (e.x > m0) { m0 = e.x; p0 = p; }
                            if (e.y > m1) { m1 = e.y; p1 = p; }
                            if (e.z > m2) { m2 = e.z; p2 = p; }
                            if (e.w > m3) { m3 = e.w; p3 = p; }
                        }
                    const int off = (2 * py + a + 1) * kA1Stride + 2 * px + b + 1;
                    s.a1[0 * kA1Plane + off] = m0;
                    s.a1[1 * kA1Plane + off] = m1;
                    s.a1[2 * kA1Plane + off] = m2;
                    s.a1[3 * kA1Plane + off] = m3;
                    const int pos = (2 * py + a) * 32 + 2 * px + b;
                    s.pat[0][pos] = (uint16_t)p0;
                    s.pat[1][pos] = (uint16_t)p1;
                    s.pat[2][pos] = (uint16_t)p2;
                    s.pat[3][pos] = (uint16_t)p3;
                }
        }
        __syncthreads();

        // ---- conv2 (4->2) + relu + 2x2 pool -> A2 (py, px); arg-max position + gate
        int am2[2];
        bool on2[2];
        {
            float acc2[2][2][2];
#pragma unroll
            for (int oc = 0; oc < 2; ++oc) { acc2[oc][0][0] = 0.f; acc2[oc][0][1] = 0.f; acc2[oc][1][0] = 0.f; acc2[oc][1][1] = 0.f; }
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const float *pl = s.a1 + cc * kA1Plane + (2 * py) * kA1Stride + 2 * px;
                float patch[4][4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float2 lo = *reinterpret_cast<const float2 *>(pl + r * kA1Stride);
                    const float2 hi = *reinterpret_cast<const float2 *>(pl + r * kA1Stride + 2);
                    patch[r][0] = lo.x; patch[r][1] = lo.y; patch[r][2] = hi.x; patch[r][3] = hi.y;
                }
#pragma unroll
                for (int oc = 0; oc < 2; ++oc)
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 2; ++dx)
#pragma unroll
                            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                                for (int kx = 0; kx < 3; ++kx)
                                    acc2[oc][dy][dx] = fmaf(w2[(oc * 4 + cc) * 9 + ky * 3 + kx],
                                                            patch[dy + ky][dx + kx], acc2[oc][dy][dx]);
            }
#pragma unroll
            for (int oc = 0; oc < 2; ++oc) {
                float mx = acc2[oc][0][0]; int am = 0;
                if (acc2[oc][0][1] > mx) { mx = acc2[oc][0][1]; am = 1; }
                if (acc2[oc][1][0] > mx) { mx = acc2[oc][1][0]; am = 2; }
                if (acc2[oc][1][1] > mx) { mx = acc2[oc][1][1]; am = 3; }
                mx += b2[oc];
                am2[oc] = am;
                on2[oc] = mx > 0.0f;
                s.a2[oc * kA2Plane + (py + 1) * kA2Stride + px + 1] = fmaxf(mx, 0.0f);
            }
        }
        __syncthreads();

        // ---- convT1 (2->1, k4 s2 p1) + relu -> A3 (2py+j, 2px+i)
        bool on3[2][2];
        {
        float r2[2][3][3];                            // A2 rows py-1..py+1, cols px-1..px+1 (padded)
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int x = 0; x < 3; ++x)
                    r2[cc][r][x] = s.a2[cc * kA2Plane + (py + r) * kA2Stride + px + x];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ry0 = 1, ky0 = j ? 2 : 1;
                const int ry1 = j ? 2 : 0, ky1 = j ? 0 : 3;
                const int cx0 = 1, kx0 = i ? 2 : 1;
                const int cx1 = i ? 2 : 0, kx1 = i ? 0 : 3;
                float acc = 0.0f;
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    acc = fmaf(r2[cc][ry0][cx0], wt1[cc * 16 + ky0 * 4 + kx0], acc);
                    acc = fmaf(r2[cc][ry0][cx1], wt1[cc * 16 + ky0 * 4 + kx1], acc);
                    acc = fmaf(r2[cc][ry1][cx0], wt1[cc * 16 + ky1 * 4 + kx0], acc);
                    acc = fmaf(r2[cc][ry1][cx1], wt1[cc * 16 + ky1 * 4 + kx1], acc);
                }
                const float a3 = fmaxf(acc + bt1, 0.0f);
                on3[j][i] = a3 > 0.0f;
                s.a3[(2 * py + j + 1) * kA1Stride + 2 * px + i + 1] = a3;
            }
        }
        __syncthreads();

        // ---- convT2 + sigmoid + loss on the thread's own 4x4 cells; backward of convT2
        float sq = 0.0f;
        {
            float t3[4][4], dt3[4][4];                // A3 rows 2py-1..2py+2, cols 2px-1..2px+2 (padded)
            const float *pl = s.a3 + (2 * py) * kA1Stride + 2 * px;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float2 lo = *reinterpret_cast<const float2 *>(pl + r * kA1Stride);
                const float2 hi = *reinterpret_cast<const float2 *>(pl + r * kA1Stride + 2);
                t3[r][0] = lo.x; t3[r][1] = lo.y; t3[r][2] = hi.x; t3[r][3] = hi.y;
                dt3[r][0] = 0.f; dt3[r][1] = 0.f; dt3[r][2] = 0.f; dt3[r][3] = 0.f;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t bits = (uint32_t)(s.rows[1 + 4 * py + r] >> (4 * px)) & 0xFu;
#pragma unroll
                for (int cx = 0; cx < 4; ++cx) {
                    const int my = r >> 1, mx = cx >> 1;
                    const int ry0 = my + 1, ky0 = (r & 1) ? 2 : 1;
                    const int ry1 = (r & 1) ? my + 2 : my, ky1 = (r & 1) ? 0 : 3;
                    const int cx0 = mx + 1, kx0 = (cx & 1) ? 2 : 1;
                    const int cx1 = (cx & 1) ? mx + 2 : mx, kx1 = (cx & 1) ? 0 : 3;
                    float acc = 0.0f;
                    acc = fmaf(t3[ry0][cx0], wt2[ky0 * 4 + kx0], acc);
                    acc = fmaf(t3[ry0][cx1], wt2[ky0 * 4 + kx1], acc);
                    acc = fmaf(t3[ry1][cx0], wt2[ky1 * 4 + kx0], acc);
                    acc = fmaf(t3[ry1][cx1], wt2[ky1 * 4 + kx1], acc);
                    const float pr = 1.0f / (1.0f + expf(-(acc + bt2)));
                    const float ob = (float)((bits >> cx) & 1u);
                    const float df = ob - pr;
                    sq = fmaf(df, df, sq);
                    const float dz = grad_scale * (pr - ob) * pr * (1.0f - pr);     // d loss / d (pre-sigmoid)
                    gbt2 += dz;
                    gwt2[ky0 * 4 + kx0] = fmaf(dz, t3[ry0][cx0], gwt2[ky0 * 4 + kx0]);
                    gwt2[ky0 * 4 + kx1] = fmaf(dz, t3[ry0][cx1], gwt2[ky0 * 4 + kx1]);
                    gwt2[ky1 * 4 + kx0] = fmaf(dz, t3[ry1][cx0], gwt2[ky1 * 4 + kx0]);
                    gwt2[ky1 * 4 + kx1] = fmaf(dz, t3[ry1][cx1], gwt2[ky1 * 4 + kx1]);
                    dt3[ry0][cx0] = fmaf(dz, wt2[ky0 * 4 + kx0], dt3[ry0][cx0]);
                    dt3[ry0][cx1] = fmaf(dz, wt2[ky0 * 4 + kx1], dt3[ry0][cx1]);
                    dt3[ry1][cx0] = fmaf(dz, wt2[ky1 * 4 + kx0], dt3[ry1][cx0]);
                    dt3[ry1][cx1] = fmaf(dz, wt2[ky1 * 4 + kx1], dt3[ry1][cx1]);
                }
            }
            // scatter into d3 (interior positions only: the border is never read)
            float *dp = s.d3 + (2 * py) * kA1Stride + 2 * px;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const int Y = 2 * py + r, X = 2 * px + x;                 // padded coordinates
                    if (Y >= 1 && Y <= 32 && X >= 1 && X <= 32) atomicAdd(dp + r * kA1Stride + x, dt3[r][x]);
                }
        }
        sq = wave_sum_f32(sq);
        if (lane == 0) s.red[wv][0] = sq;
        __syncthreads();
        if (tid == 0) {
            const float b = ((s.red[0][0] + s.red[1][0]) + (s.red[2][0] + s.red[3][0])) * (1.0f / 4096.0f);
            bonus[g] = b;
            if (reward) reward[g] += reward_scale * b;
        }

        // ---- relu gate of A3, convT1 backward (thread's own 2x2 outputs)
        {
            float r2[2][3][3], dr2[2][3][3];          // A2 neighbourhood re-read: cheaper than 18 live registers
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int x = 0; x < 3; ++x)
                        r2[cc][r][x] = s.a2[cc * kA2Plane + (py + r) * kA2Stride + px + x];
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int r = 0; r < 3; ++r) { dr2[cc][r][0] = 0.f; dr2[cc][r][1] = 0.f; dr2[cc][r][2] = 0.f; }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    float *own = s.d3 + (2 * py + j + 1) * kA1Stride + 2 * px + i + 1;
                    const float gz = on3[j][i] ? *own : 0.0f;
                    *own = 0.0f;                                   // ready for the next grid
                    gbt1 += gz;
                    const int ry0 = 1, ky0 = j ? 2 : 1;
                    const int ry1 = j ? 2 : 0, ky1 = j ? 0 : 3;
                    const int cx0 = 1, kx0 = i ? 2 : 1;
                    const int cx1 = i ? 2 : 0, kx1 = i ? 0 : 3;
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc) {
                        gwt1[cc * 16 + ky0 * 4 + kx0] = fmaf(gz, r2[cc][ry0][cx0], gwt1[cc * 16 + ky0 * 4 + kx0]);
                        gwt1[cc * 16 + ky0 * 4 + kx1] = fmaf(gz, r2[cc][ry0][cx1], gwt1[cc * 16 + ky0 * 4 + kx1]);
                        gwt1[cc * 16 + ky1 * 4 + kx0] = fmaf(gz, r2[cc][ry1][cx0], gwt1[cc * 16 + ky1 * 4 + kx0]);
                        gwt1[cc * 16 + ky1 * 4 + kx1] = fmaf(gz, r2[cc][ry1][cx1], gwt1[cc * 16 + ky1 * 4 + kx1]);
                        dr2[cc][ry0][cx0] = fmaf(gz, wt1[cc * 16 + ky0 * 4 + kx0], dr2[cc][ry0][cx0]);
                        dr2[cc][ry0][cx1] = fmaf(gz, wt1[cc * 16 + ky0 * 4 + kx1], dr2[cc][ry0][cx1]);
                        dr2[cc][ry1][cx0] = fmaf(gz, wt1[cc * 16 + ky1 * 4 + kx0], dr2[cc][ry1][cx0]);
                        dr2[cc][ry1][cx1] = fmaf(gz, wt1[cc * 16 + ky1 * 4 + kx1], dr2[cc][ry1][cx1]);
                    }
                }
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int x = 0; x < 3; ++x) {
                        const int Y = py + r, X = px + x;                    // padded coordinates
                        if (Y >= 1 && Y <= 16 && X >= 1 && X <= 16)
                            atomicAdd(&s.d2[cc * kA2Plane + Y * kA2Stride + X], dr2[cc][r][x]);
                    }
        }
        __syncthreads();

        // ---- 2x2 pool arg-max + relu gate of A2 -> the conv2-output gradient map
#pragma unroll
        for (int oc = 0; oc < 2; ++oc) {
            float *own = &s.d2[oc * kA2Plane + (py + 1) * kA2Stride + px + 1];
            const float gz = on2[oc] ? *own : 0.0f;
            *own = 0.0f;
            gb2[oc] += gz;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                s.map2[oc * kA1Plane + (2 * py + (e >> 1) + 1) * kA1Stride + 2 * px + (e & 1) + 1] =
                    e == am2[oc] ? gz : 0.0f;
        }
        __syncthreads();

        // ---- conv2 and conv1 weight gradients of input channel wv: lane = 16 positions of A1
        //      row Y = lane / 2, columns 16 * (lane & 1) ..
        {
            const int Y = lane >> 1, X0 = 16 * (lane & 1);
            const float *ap = s.a1 + wvs * kA1Plane + (Y + 1) * kA1Stride + X0 + 1;
            const uint16_t *pp = &s.pat[wvs][Y * 32 + X0];
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                float mp[2][3][6];                    // map rows Y..Y+2, cols X0+4ch .. +5 (padded)
#pragma unroll
                for (int oc = 0; oc < 2; ++oc)
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const float *mr = s.map2 + oc * kA1Plane + (Y + r) * kA1Stride + X0 + 4 * ch;
                        const float2 m01 = *reinterpret_cast<const float2 *>(mr);
                        const float2 m23 = *reinterpret_cast<const float2 *>(mr + 2);
                        const float2 m45 = *reinterpret_cast<const float2 *>(mr + 4);
                        mp[oc][r][0] = m01.x; mp[oc][r][1] = m01.y; mp[oc][r][2] = m23.x;
                        mp[oc][r][3] = m23.y; mp[oc][r][4] = m45.x; mp[oc][r][5] = m45.y;
                    }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float av = ap[4 * ch + i];
                    const uint32_t pt = pp[4 * ch + i];
                    float d = 0.0f;
#pragma unroll
                    for (int oc = 0; oc < 2; ++oc)
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx) {
                                const float mv = mp[oc][2 - ky][i + 2 - kx];
                                gw2c[oc * 9 + ky * 3 + kx] = fmaf(av, mv, gw2c[oc * 9 + ky * 3 + kx]);
                                d = fmaf(wc[oc * 9 + ky * 3 + kx], mv, d);
                            }
                    const float gz = av > 0.0f ? d : 0.0f;
                    gb1c += gz;
#pragma unroll
                    for (int tt = 0; tt < 9; ++tt)
                        gw1c[tt] = fmaf(gz, (float)((pt >> tt) & 1u), gw1c[tt]);
                }
            }
        }
        // the next iteration's barriers order its writes to rows / A1 / pat / map2 behind these reads
    }

    // ---- one partial gradient row per workgroup, parameter order of the AE blob
    float *row = partial + (int64_t)blockIdx.x * kAeParams;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 52; ++i) {
        // 0..15 convT2.w, 16 convT2.b, 17..48 convT1.w, 49 convT1.b, 50..51 conv2.b
        float a = i < 16 ? gwt2[i] : i == 16 ? gbt2 : i < 49 ? gwt1[i - 17] : i == 49 ? gbt1 : gb2[i - 50];
        a = wave_sum_f32(a);
        if (lane == 0) s.red[wv][i] = a;
    }
    __syncthreads();
    if (tid < 52) {
        const float a = (s.red[0][tid] + s.red[1][tid]) + (s.red[2][tid] + s.red[3][tid]);
        const int dst = tid < 16 ? 147 + tid : tid == 16 ? 163 : tid < 49 ? 114 + (tid - 17) : tid == 49 ? 146 : 112 + (tid - 50);
        row[dst] = a;
    }
    // channel-split parts: wave wv owns conv1 channel wv and conv2 input channel wv
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const float a = wave_sum_f32(gw1c[i]);
        if (lane == 0) row[wv * 9 + i] = a;
    }
    {
        const float a = wave_sum_f32(gb1c);
        if (lane == 0) row[36 + wv] = a;
    }
#pragma unroll
    for (int oc = 0; oc < 2; ++oc)
#pragma unroll
        for (int kq = 0; kq < 9; ++kq) {
            const float a = wave_sum_f32(gw2c[oc * 9 + kq]);
            if (lane == 0) row[40 + (oc * 4 + wv) * 9 + kq] = a;
        }
}

int ae2d_train64_rows(int64_t n) {                 // = workgroups of ae_train_block64_kernel
    int64_t rows = n < 512 ? n : 512;                  // persistent: 2 workgroups are resident per CU
    return (int)(rows < 1 ? 1 : rows);
}

int ae2d_train64(const void *obs, bool packed, int64_t n, const float *ap, float *ws, float *bonus,
                 float *reward, float reward_scale, float grad_scale, float *partial, float *grad,
                 bool tables_valid, hipStream_t s) {
    if (!tables_valid) hipLaunchKernelGGL(ae_prepare_kernel, dim3(1), dim3(512), 0, s, ap, ws);
    const int rows = ae2d_train64_rows(n);
    if (packed)
        hipLaunchKernelGGL(ae_train_block64_kernel<true>, dim3((unsigned)rows), dim3(256), 0, s, obs, ws, bonus,
                           reward, reward_scale, grad_scale, partial, n);
    else
        hipLaunchKernelGGL(ae_train_block64_kernel<false>, dim3((unsigned)rows), dim3(256), 0, s, obs, ws, bonus,
                           reward, reward_scale, grad_scale, partial, n);
    hipLaunchKernelGGL(rnd_grad_reduce_kernel, dim3((kAeParams + 31) / 32), dim3(1024), 0, s, partial, rows, grad,
                       kAeParams);
    return check_launch("carle_ae2d_train(fused 64x64)");
}

}  // namespace carle

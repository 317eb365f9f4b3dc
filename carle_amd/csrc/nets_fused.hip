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
#include <stdlib.h>
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
    if (first < n) issue(first);                     // the first grid travels while the tables are laid down
    rnd_block_init(s, ws, tid);
    __syncthreads();
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
        __syncthreads();
        // (the random net's table reads go through the vector L1, the predictor's through LDS: 22.9 -> 21.2 us at N=4096; not in
        // the fused CA + RND2D step, whose vector memory path carries the next grid -- loads return in order: 36.5 -> 41.4 us)
        rnd_block_conv1<CARLE_RND_CONV1_BATCH, 0, 16>(s, tid, ws);
        __syncthreads();
        if (g + stride < n) issue(g + stride);        // the next grid travels during the tail (behind conv1's loads: in-order returns)
        rnd_block_conv2(s, s_feat, tid);
        __syncthreads();
        if (wv == 0) {
            const float b = rnd_linear_bonus(s_feat, s.lin, lane, g, pred_out, target_out);
            if (lane == 0) {
                bonus[g] = b;
                if (reward) reward[g] = reward_accumulate(reward[g], reward_scale, b);       // mcl.py:403-405
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
    if (const int64_t v = knobs().net_blocks; v > 0) blocks = v < n ? v : n;   // experiment knob, read once
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
#ifndef CARLE_AE_LUT_VMEM
#define CARLE_AE_LUT_VMEM 2
#endif
static constexpr int kAeLutVmem = CARLE_AE_LUT_VMEM;          // ae_block64_kernel: pooled outputs (of 4) whose table reads go through L1
static constexpr int kAePairFloats = 72;                      // conv2.w interleaved by output channel (64x64 kernels)
static constexpr int kAeWsFloats = kAeLutFloats + kAeParamFloats + kAePairFloats;
static constexpr int kA1Stride = 36, kA1Rows = 34, kA1Plane = kA1Rows * kA1Stride;   // also A3
static constexpr int kA2Stride = 20, kA2Rows = 18, kA2Plane = kA2Rows * kA2Stride;

// PAIRS (the 64x64 kernels' workspace only): conv2.w once more behind the parameter block, the two output channels of
// every (input channel, tap) side by side -- operand pairs of v_pk_fma_f32 (ae_block64_kernel)
template <bool PAIRS>
__global__ __launch_bounds__(512) void ae_prepare_kernel(const float *__restrict__ ap,
                                                         float *__restrict__ ws) {
    const int p = threadIdx.x;
    if (PAIRS && p >= 256 && p < 256 + 72) {
        const int i = p - 256;                        // = (cc * 9 + tap) * 2 + oc
        ws[kAeLutFloats + kAeParamFloats + i] = ap[40 + (i & 1) * 36 + (i >> 1)];
    }
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

// sigmoid through v_exp_f32 + v_rcp_f32 (~2 ulp each; |error| < 5e-7 absolute): the libm expf + IEEE division
// pair costs ~25 instructions per cell, a third of the AE2D forward kernel
// (round 4: __frcp_rn is the correctly ROUNDED reciprocal -- v_div_scale x 2, v_rcp, four fma, v_div_fmas, v_div_fixup: ten
// instructions per cell, 16 % of the forward kernel's VALU instructions; the bare v_rcp_f32 is within 1 ulp)
__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}

// sum over each 16-lane DPP row, in every lane of the row: quad_perm xor 1, xor 2, row_half_mirror, row_mirror
__device__ __forceinline__ float row16_sum_f32(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
    return v;
}

// ds_read_b64 at an immediate byte offset from an LDS address; the value is there after lds_wait_patch names it
template <int OFF>
__device__ __forceinline__ f32x2 lds_read_b64_issue(uint32_t addr) {
    f32x2 v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
// the eight halves of the 4x4 patch of A1 plane CC (rows of kA1Stride floats, planes of kA1Plane)
template <int CC>
__device__ __forceinline__ void lds_issue_patch(uint32_t addr, f32x2 (&p)[8]) {
    p[0] = lds_read_b64_issue<(CC * kA1Plane + 0 * kA1Stride) * 4>(addr);
    p[1] = lds_read_b64_issue<(CC * kA1Plane + 0 * kA1Stride + 2) * 4>(addr);
    p[2] = lds_read_b64_issue<(CC * kA1Plane + 1 * kA1Stride) * 4>(addr);
    p[3] = lds_read_b64_issue<(CC * kA1Plane + 1 * kA1Stride + 2) * 4>(addr);
    p[4] = lds_read_b64_issue<(CC * kA1Plane + 2 * kA1Stride) * 4>(addr);
    p[5] = lds_read_b64_issue<(CC * kA1Plane + 2 * kA1Stride + 2) * 4>(addr);
    p[6] = lds_read_b64_issue<(CC * kA1Plane + 3 * kA1Stride) * 4>(addr);
    p[7] = lds_read_b64_issue<(CC * kA1Plane + 3 * kA1Stride + 2) * 4>(addr);
}
// all but the N youngest LDS operations of this wave have returned (they return in order); p is what that releases
template <int N>
__device__ __forceinline__ void lds_wait_patch(f32x2 (&p)[8]) {
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7])
                 : "n"(N) : "memory");
}

// One workgroup (4 waves) per grid, persistent over grids; thread (py = tid/16, px = tid%16) owns the
// 4x4 cell block rows 4py.., cols 4px.. in every layer (its 2x2 outputs of A1 / A3, its one output
// of A2), phases separated by barriers.  31 KB of LDS -> 4-5 workgroups (16-20 waves) per CU.
struct AeBlockShared {
    float4 lut[512];
    float a1[4 * kA1Plane];          // A1; A3 later overlays plane 0
    float a2[2 * kA2Plane];
    uint64_t rows[66];               // the grid, one 64-bit word per row at [1..64]; [0], [65] stay zero
    uint64_t trows[64];              // TGT: the frame the reconstruction is compared with (PredictionBonus)
    float red[16];                   // squared error per 16-lane row of the workgroup
};

// (kfloat / opaque_uniform: common.hpp)

// TGT (PredictionBonus, mcl.py:581-605): the squared error is taken against another bit-packed frame `tgt` of the same
// universes instead of the input itself
// STATS (round 5; an AE2D that is the OUTERMOST wrapper of a stack whose rewards feed a RunningRewardStats): the rewards this
// kernel finishes are the step's final ones, so every workgroup leaves (sum, sum of squares) of the rewards it wrote in
// `partials` and workgroup 0 registers them in the environment's status block (CARLE_ST_MOM_*): the next launch on the block
// merges them -- the stand-alone moments launch (a dependent 5 us dispatch per step) is gone, as for the tiled pair.
template <bool PACKED_OBS, bool TGT = false, bool STATS = false>
__global__ __launch_bounds__(256, 4) void ae_block64_kernel(
    const void *__restrict__ obs_v, const float *__restrict__ ws, float *__restrict__ bonus,
    float *__restrict__ recon, float *__restrict__ reward, float reward_scale, int64_t n,
    const uint64_t *__restrict__ tgt, double *__restrict__ partials = nullptr, int32_t *__restrict__ status = nullptr) {
    __shared__ AeBlockShared s;
    double rsum = 0.0, rsq = 0.0;                    // STATS: thread 0's sums over this workgroup's grids
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float *prm = ws + kAeLutFloats;            // uniform parameter block
    // (the 124 uniform weights are scalar-loaded inside the phase that uses them: opaque_uniform)

    const int q = lane >> 4, k = lane & 15;          // load mapping: rows 16w+4i+q, columns 4k..4k+3
    // compute mapping: px = the lane of a DPP row; py with bits 0 and 2 of tid / 16 exchanged -- lanes 16-31 of a 32-lane LDS group
    // then sit four block rows below lanes 0-15 (8 plane rows x 36 floats = 32 banks on): their ds_read_b64 of A1 / A3 do not collide
    const int pyr = tid >> 4, py = (pyr & 10) | ((pyr & 1) << 2) | ((pyr >> 2) & 1), px = tid & 15;
    const int64_t first = blockIdx.x, stride = gridDim.x;
    uint4 v[4];
    uint64_t word = 0, tword = 0;
    auto issue = [&](int64_t g) {
        if (TGT) {
            if (tid < 64) tword = tgt[g * 64 + tid];
        }
        if (PACKED_OBS) {
            if (tid < 64) word = reinterpret_cast<const uint64_t *>(obs_v)[g * 64 + tid];
        } else {
            const uint4 *gin = reinterpret_cast<const uint4 *>(obs_v) + g * 1024 + 256 * wv + lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = gin[i * 64];
        }
    };
    if (first < n) issue(first);                     // the first grid travels while the table is laid down
    {
        const float4 *src = reinterpret_cast<const float4 *>(ws);
        lds_copy_f4<512, 256>(s.lut, src, tid);
        for (int i = tid; i < 4 * kA1Plane; i += 256) s.a1[i] = 0.0f;
        for (int i = tid; i < 2 * kA2Plane; i += 256) s.a2[i] = 0.0f;
        if (tid < 66) s.rows[tid] = 0ull;
    }
    __syncthreads();
    for (int64_t g = first; g < n; g += stride) {
        // ---- the grid as 64 row words in LDS
        if (TGT) {
            if (tid < 64) s.trows[tid] = tword;
        }
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
// the lookups of kAeLutVmem of the window's four pooled outputs go to the table's copy in global memory
                        // (8 KB, resident in the CU's vector L1) instead of LDS: the LDS array is this kernel's busiest unit
                        // (63 % of the cycles, a third of them bank conflicts of exactly these reads), the vector memory path
                        // is idle.  Same box, N=4096: 25.5 (0 of 4) / 25.1 (1) / 24.2 (2) / 24.2 (3) / 29.4 us (4)
                        const float4 e = (2 * a + b < kAeLutVmem) ? reinterpret_cast<const float4 *>(ws)[p] : s.lut[p];
                        m0 = fmaxf(m0, e.x); m1 = fmaxf(m1, e.y); m2 = fmaxf(m2, e.z); m3 = fmaxf(m3, e.w);
                    }
                const int off = (2 * py + a + 1) * kA1Stride + 2 * px + b + 1;
                s.a1[0 * kA1Plane + off] = m0;
                s.a1[1 * kA1Plane + off] = m1;
                s.a1[2 * kA1Plane + off] = m2;
                s.a1[3 * kA1Plane + off] = m3;
            }
        __syncthreads();
        // the next grid travels during the remaining layers (requested BEHIND conv1: vector loads return in order, and the
        // table reads that conv1 sends through L1 would wait for this one's trip to HBM)
        if (g + stride < n) issue(g + stride);

        // ---- conv2 (4->2, 3x3, zero pad) + relu + 2x2 pool -> A2 (py, px)
        // both output channels of a tap in ONE v_pk_fma_f32 (weights: the interleaved copy, an SGPR pair; the input value
        // broadcast by op_sel): 144 packed instead of 288 scalar FMAs per thread and grid, the same order of summation
        {
            kfloat *b2 = opaque_uniform(prm) + 72;
            const f32x2 __attribute__((address_space(4))) *w2p =
                reinterpret_cast<const f32x2 __attribute__((address_space(4))) *>(opaque_uniform(prm + kAeParamFloats));
            f32x2 acc2[2][2];
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) { acc2[dy][0] = f32x2{0.f, 0.f}; acc2[dy][1] = f32x2{0.f, 0.f}; }
            // The 4x4 patches: EIGHT ds_read_b64 per input plane, written out -- the compiler merges any two of them into a
            // ds_read2_b64, which occupies the LDS array for 8 cycles where two ds_read_b64 take 2 + 2 (MI355X_MICROARCH,
            // LDS).  Plane cc + 1 is in flight while plane cc is consumed; the waits name the registers they release.
            const uint32_t a1_addr = (uint32_t)reinterpret_cast<uintptr_t>(s.a1 + (2 * py) * kA1Stride + 2 * px);   // padded rows 2py.., cols 2px..
            f32x2 pa[8], pb[8];
            lds_issue_patch<0>(a1_addr, pa);
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                f32x2 *cur = (cc & 1) ? pb : pa;
                if (cc == 0) { lds_issue_patch<1>(a1_addr, pb); lds_wait_patch<8>(pa); }
                if (cc == 1) { lds_issue_patch<2>(a1_addr, pa); lds_wait_patch<8>(pb); }
                if (cc == 2) { lds_issue_patch<3>(a1_addr, pb); lds_wait_patch<8>(pa); }
                if (cc == 3) lds_wait_patch<0>(pb);
                float patch[4][4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    patch[r][0] = cur[2 * r].x; patch[r][1] = cur[2 * r].y; patch[r][2] = cur[2 * r + 1].x; patch[r][3] = cur[2 * r + 1].y;
                }
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx)
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx) {
                                const float pv = patch[dy + ky][dx + kx];
                                acc2[dy][dx] = __builtin_elementwise_fma(w2p[cc * 9 + ky * 3 + kx], f32x2{pv, pv}, acc2[dy][dx]);
                            }
            }
            const float mx0 = fmaxf(fmaxf(acc2[0][0].x, acc2[0][1].x), fmaxf(acc2[1][0].x, acc2[1][1].x)) + b2[0];
            const float mx1 = fmaxf(fmaxf(acc2[0][0].y, acc2[0][1].y), fmaxf(acc2[1][0].y, acc2[1][1].y)) + b2[1];
            s.a2[0 * kA2Plane + (py + 1) * kA2Stride + px + 1] = fmaxf(mx0, 0.0f);
            s.a2[1 * kA2Plane + (py + 1) * kA2Stride + px + 1] = fmaxf(mx1, 0.0f);
        }
        __syncthreads();                              // A1 reads done: A3 may overlay plane 0

        // ---- convT1 (2->1, k4 s2 p1) + relu -> A3 (2py+j, 2px+i), overlaying A1 plane 0
        {
            kfloat *wt1 = opaque_uniform(prm) + 74;
            const float bt1 = wt1[32];
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
        // columns cx and cx + 2 of the block take the same taps: one v_pk_fma_f32 / v_pk_add_f32 serves both (the A3 values
        // arrive as horizontal pairs (0,1), (1,2), (2,3) of each row); per cell the order of summation of the scalar form.
        // The exponent of the sigmoid in one more packed fma: -log2(e) * (acc + bias) = fma(acc, -log2 e, -log2 e * bias)
        float sq = 0.0f;
        {
            kfloat *wt2 = opaque_uniform(prm) + 107;
            constexpr float kNegLog2e = -1.4426950408889634f;
            const float bt2s = kNegLog2e * wt2[16];
            typedef f32x2 __attribute__((aligned(4))) f32x2_a4;
            f32x2 h01[4], h12[4], h23[4];             // A3 rows 2py-1..2py+2 (padded), column pairs from 2px-1
            const float *pl = s.a1 + (2 * py) * kA1Stride + 2 * px;
            // (the aligned pairs as single ds_read_b64, as the conv2 patches above; the unaligned ones are ds_read2_b32 anyway)
            {
                const uint32_t a3 = (uint32_t)reinterpret_cast<uintptr_t>(pl);
                h01[0] = lds_read_b64_issue<(0 * kA1Stride) * 4>(a3); h23[0] = lds_read_b64_issue<(0 * kA1Stride + 2) * 4>(a3);
                h01[1] = lds_read_b64_issue<(1 * kA1Stride) * 4>(a3); h23[1] = lds_read_b64_issue<(1 * kA1Stride + 2) * 4>(a3);
                h01[2] = lds_read_b64_issue<(2 * kA1Stride) * 4>(a3); h23[2] = lds_read_b64_issue<(2 * kA1Stride + 2) * 4>(a3);
                h01[3] = lds_read_b64_issue<(3 * kA1Stride) * 4>(a3); h23[3] = lds_read_b64_issue<(3 * kA1Stride + 2) * 4>(a3);
#pragma unroll
                for (int r = 0; r < 4; ++r) h12[r] = *reinterpret_cast<const f32x2_a4 *>(pl + r * kA1Stride + 1);
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(h01[0]), "+v"(h01[1]), "+v"(h01[2]), "+v"(h01[3]), "+v"(h23[0]), "+v"(h23[1]), "+v"(h23[2]), "+v"(h23[3])
                             : : "memory");
            }
            f32x2 sq2 = {0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t bits = (uint32_t)((TGT ? s.trows[4 * py + r] : s.rows[1 + 4 * py + r]) >> (4 * px)) & 0xFu;
                const int my = r >> 1;                                // local A3 index of iy = m is m + 1
                const int ry0 = my + 1, ky0 = (r & 1) ? 2 : 1;
                const int ry1 = (r & 1) ? my + 2 : my, ky1 = (r & 1) ? 0 : 3;
                float out4[4];
#pragma unroll
                for (int cx = 0; cx < 2; ++cx) {      // cells cx and cx + 2: inputs (mx + 1) = (1, 2); (mx) = (0, 1) or (mx + 2) = (2, 3)
                    const int kx0 = cx ? 2 : 1, kx1 = cx ? 0 : 3;
                    const f32x2 in0a = h12[ry0], in0b = cx ? h23[ry0] : h01[ry0];
                    const f32x2 in1a = h12[ry1], in1b = cx ? h23[ry1] : h01[ry1];
                    f32x2 acc = {0.f, 0.f};
                    acc = __builtin_elementwise_fma(in0a, f32x2{wt2[ky0 * 4 + kx0], wt2[ky0 * 4 + kx0]}, acc);
                    acc = __builtin_elementwise_fma(in0b, f32x2{wt2[ky0 * 4 + kx1], wt2[ky0 * 4 + kx1]}, acc);
                    acc = __builtin_elementwise_fma(in1a, f32x2{wt2[ky1 * 4 + kx0], wt2[ky1 * 4 + kx0]}, acc);
                    acc = __builtin_elementwise_fma(in1b, f32x2{wt2[ky1 * 4 + kx1], wt2[ky1 * 4 + kx1]}, acc);
                    const f32x2 t = __builtin_elementwise_fma(acc, f32x2{kNegLog2e, kNegLog2e}, f32x2{bt2s, bt2s});
                    const f32x2 den = f32x2{1.0f, 1.0f} + f32x2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
                    const f32x2 pr = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};   // sigmoid, twice
                    out4[cx] = pr.x; out4[cx + 2] = pr.y;
                    const f32x2 df = f32x2{(float)((bits >> cx) & 1u), (float)((bits >> (cx + 2)) & 1u)} - pr;
                    sq2 = __builtin_elementwise_fma(df, df, sq2);
                }
                if (recon)
                    *reinterpret_cast<float4 *>(recon + g * 4096 + (4 * py + r) * 64 + 4 * px) =
                        make_float4(out4[0], out4[1], out4[2], out4[3]);
            }
            sq = sq2.x + sq2.y;
        }
        sq = row16_sum_f32(sq);                       // (four DPP adds; the wave-wide butterfly was six ds_bpermute)
        if ((lane & 15) == 0) s.red[tid >> 4] = sq;
        __syncthreads();                              // also: rows / A3 reads done before the next grid
        if (tid == 0) {
            const float4 *r4 = reinterpret_cast<const float4 *>(s.red);
            float b = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) b += (r4[i].x + r4[i].y) + (r4[i].z + r4[i].w);
            b *= 1.0f / 4096.0f;
            bonus[g] = b;
            if (reward) {
                const float r = reward_accumulate(reward[g], reward_scale, b);
                reward[g] = r;
                if (STATS) { rsum += (double)r; rsq = fma((double)r, (double)r, rsq); }
            }
        }
    }
    if (STATS && tid == 0) {
        reinterpret_cast<double2 *>(partials)[blockIdx.x] = make_double2(rsum, rsq);
        if (blockIdx.x == 0) {                       // pending reward moments of this step: merged by the next call on the block
            uint4 w5;
            w5.x = (uint32_t)reinterpret_cast<uint64_t>(partials);
            w5.y = (uint32_t)(reinterpret_cast<uint64_t>(partials) >> 32);
            w5.z = gridDim.x;
            w5.w = (uint32_t)n;
            reinterpret_cast<uint4 *>(status)[5] = w5;
        }
    }
}

int64_t ae2d_fused_workspace_floats() { return kAeWsFloats; }
int64_t ae2d_fused64_blocks(int64_t n) {
    int64_t blocks = n < 1280 ? n : 1280;             // persistent: 5 workgroups are resident per CU
    if (const int64_t v = knobs().net_blocks; v > 0) blocks = v < n ? v : n;   // experiment knob, read once
    return blocks < 1 ? 1 : blocks;
}

// conv1 table + the uniform parameter block, ws[0 .. kAeWsFloats): shared with the tiled kernel (nets_tiled.hip)
void ae2d_prepare_tables(const float *ap, float *ws, hipStream_t s) {
    hipLaunchKernelGGL(ae_prepare_kernel<false>, dim3(1), dim3(512), 0, s, ap, ws);
}

int ae2d_fused64(const void *obs, bool packed, int64_t n, const float *ap, float *ws, float *bonus,
                 float *recon, float *reward, float reward_scale, bool tables_valid, hipStream_t s,
                 const uint64_t *tgt, double *partials, int32_t *status) {
    if (!tables_valid) hipLaunchKernelGGL(ae_prepare_kernel<true>, dim3(1), dim3(512), 0, s, ap, ws);
    const int64_t blocks = ae2d_fused64_blocks(n);
    const uint64_t *no_tgt = nullptr;
    if (partials && status && packed && !tgt && reward) {
        hipLaunchKernelGGL((ae_block64_kernel<true, false, true>), dim3((unsigned)blocks), dim3(256), 0, s, obs, ws, bonus,
                           recon, reward, reward_scale, n, no_tgt, partials, status);
        return check_launch("carle_ae2d_bonus(fused 64x64, reward moments)");
    }
    if (tgt)                                          // (bit-packed frames only)
        hipLaunchKernelGGL((ae_block64_kernel<true, true>), dim3((unsigned)blocks), dim3(256), 0, s, obs, ws, bonus,
                           recon, reward, reward_scale, n, tgt);
    else if (packed)
        hipLaunchKernelGGL((ae_block64_kernel<true, false>), dim3((unsigned)blocks), dim3(256), 0, s, obs, ws, bonus,
                           recon, reward, reward_scale, n, no_tgt);
    else
        hipLaunchKernelGGL((ae_block64_kernel<false, false>), dim3((unsigned)blocks), dim3(256), 0, s, obs, ws, bonus,
                           recon, reward, reward_scale, n, no_tgt);
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
                if (reward) reward[g] = reward_accumulate(reward[g], reward_scale, b);
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

// ONE WORKGROUP per grid, persistent, 3 workgroups per CU (<= 168 VGPRs, 49 KB of LDS).  Forward as
// ae_block64_kernel (thread (py,px) owns the 4x4 cell block at (4py,4px) in every layer).  Backward:
//   convT2 / convT1: a thread accumulates the contributions of its own outputs to the 4x4 (3x3) input
//     neighbourhood in registers; the columns that belong to the threads px-1 / px+1 travel by DPP
//     row shifts (px = lane % 16 is a DPP row; the shift's zero fill IS the plane border), the rows that
//     belong to py-1 / py+1 through a halo buffer in LDS that the owner reads behind the next barrier:
//     no atomics, a fixed summation order;
//   conv2: the gradient w.r.t. its outputs goes to a sparse map (one entry per 2x2 pooling block);
//   conv2 + conv1 weights: CHANNEL-SPLIT -- wave c walks all 32x32 positions of A1 plane c (16 per
//     lane), gathers  d A1[c](P) = sum_{o,k} w2[o][c][k] map[o](P-k+1),  accumulates
//     d w2[o][c][k] += A1[c](P) map[o](P-k+1)  and, through the relu gate and the arg-max pattern
//     of P, the conv1 gradients of channel c: 18 + 9 + 1 accumulators per lane instead of 72 + 36 + 4.
//     The arg-max patterns are not stored: per A1 position the 4x4 cell neighbourhood (16 bits, shared
//     by the four channels) + a 2-bit arg-max index per channel.
// LDS: the map overlays A3 + the convT2 halo (both dead by then; the 132 border entries of its
// second plane that the gather reads are re-zeroed every grid).
// History: round 1 (62 KB of LDS, ds_add_f32 scatters, 256 VGPRs + 72 spilled, 2 workgroups per CU):
// 288-304 us per 4096 grids.
struct AeTrainShared {
    float4 lut[512];
    float a1[4 * kA1Plane];
    float a2[2 * kA2Plane];
    union {
        struct {
            float a3[kA1Plane];
            float2 halo3[2][256];    // [0]: for the thread above (its row 2), [1]: for the thread below (its row 1)
            float pad[(kA1Plane * 4 - 4096) / 4];
        } f;
        float map2[2 * kA1Plane];    // d loss / d conv2 output: one non-zero per 2x2 pooling block
    } u;
    float2 halo2[2][256];            // convT1 backward: (cc0, cc1) for the thread above / below
    uint16_t nb[1024];               // 4x4 cell neighbourhood of every A1 position (32x32, row-major)
    uint8_t am[1024];                // 2-bit arg-max index (dy * 2 + dx) of the four conv1 channels
    uint64_t rows[66];
    uint64_t trows[64];              // TGT: the frame the reconstruction is compared with (PredictionBonus)
    float red[4][52];
};
static_assert(sizeof(AeTrainShared) <= 54272, "three workgroups per CU");

template <int CTRL>
__device__ __forceinline__ float dpp_mov0(float x) {   // DPP move, lanes without a source read 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
static constexpr int kRowShr1 = 0x111, kRowShl1 = 0x101;   // value of lane - 1 / lane + 1 of the 16-lane row

// sum over the 64 lanes without LDS round trips; the result is valid in every lane
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_quad_xor1(v);
    v += dpp_quad_xor2(v);
    v += dpp_row_ror(v, 0x124);
    v += dpp_row_ror(v, 0x128);
    const int b = __builtin_bit_cast(int, v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) +
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) +
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48)));
}

// Reduce per-thread contributions c[0..N) over the four lanes of a DPP quad while transposing them:
// afterwards lane l of the quad has added the quad's sum of entry 4m + (l & 3) to acc[m] -- N / 4
// accumulators per thread instead of N, 2.25 N extra VALU operations, a fixed summation order.
template <int N>
__device__ __forceinline__ void quad_transpose_add(const float (&c)[N], float (&acc)[N / 4], int lane) {
    const bool b0 = lane & 1, b1 = lane & 2;
    float h[N / 2];
#pragma unroll
    for (int m = 0; m < N / 2; ++m) {
        const float mine = b0 ? c[2 * m + 1] : c[2 * m], send = b0 ? c[2 * m] : c[2 * m + 1];
        h[m] = mine + dpp_quad_xor1(send);
    }
#pragma unroll
    for (int m = 0; m < N / 4; ++m) {
        const float mine = b1 ? h[2 * m + 1] : h[2 * m], send = b1 ? h[2 * m] : h[2 * m + 1];
        acc[m] += mine + dpp_quad_xor2(send);
    }
}

// total over the wave of a quad-transposed accumulator: every lane ends with the sum of the lanes
// that share its (lane & 3)
__device__ __forceinline__ float quad_class_sum(float v) {
    v += dpp_row_ror(v, 0x124);
    v += dpp_row_ror(v, 0x128);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}



template <bool PACKED_OBS, bool TGT = false>
__global__ __launch_bounds__(256, 3) void ae_train_block64_kernel(
    const void *__restrict__ obs_v, const float *__restrict__ ws, float *__restrict__ bonus,
    float *__restrict__ reward, float reward_scale, float grad_scale, float *__restrict__ partial,
    int64_t n, const uint64_t *__restrict__ tgt) {
    __shared__ AeTrainShared s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    {
        const float4 *src = reinterpret_cast<const float4 *>(ws);
        lds_copy_f4<512, 256>(s.lut, src, tid);
        for (int i = tid; i < 4 * kA1Plane; i += 256) s.a1[i] = 0.0f;
        for (int i = tid; i < 2 * kA2Plane; i += 256) s.a2[i] = 0.0f;
        for (int i = tid; i < 2 * kA1Plane; i += 256) s.u.map2[i] = 0.0f;       // A3's border with it
        if (tid < 66) s.rows[tid] = 0ull;
    }
    __syncthreads();
    const float *prm = ws + kAeLutFloats;            // uniform parameter block:
    // conv2.w[72] b[2] | convT1.w[32] b[1] | convT2.w[16] b[1]

    const int q = lane >> 4, k = lane & 15;          // load mapping
    const int py = tid >> 4, px = tid & 15;          // cell-block mapping

    // convT2 / convT1 weight gradients, quad-transposed: lane l holds entries 4m + (l & 3)
    float gwt2q[4] = {0.f, 0.f, 0.f, 0.f}, gbt2 = 0.0f, gwt1q[8], gbt1 = 0.0f, gb2[2] = {0.0f, 0.0f};
    float gw2c[18], gw1c[9], gb1c = 0.0f;            // conv2 / conv1 gradients of input channel wv
#pragma unroll
    for (int i = 0; i < 8; ++i) gwt1q[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 18; ++i) gw2c[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 9; ++i) gw1c[i] = 0.0f;
    const int wvs = __builtin_amdgcn_readfirstlane(wv);   // provably wave-uniform: scalar loads below

    const int64_t first = blockIdx.x, stride = gridDim.x;
    uint4 v[4];
    uint32_t xr[4];                                  // fp32 observations: this lane's dword of rows 16wv+4i+q
    uint64_t word = 0, tword = 0;
    auto convert = [&]() {                            // waits for the loads; 4 live registers instead of 16
        if (!PACKED_OBS) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t nib = nz_bit(v[i].x) | (nz_bit(v[i].y) << 1) | (nz_bit(v[i].z) << 2) |
                                     (nz_bit(v[i].w) << 3);
                xr[i] = row_nibbles_to_dword(nib, k);
            }
        }
    };
    auto issue = [&](int64_t g) {
        if (TGT) {
            if (tid < 64) tword = tgt[g * 64 + tid];
        }
        if (PACKED_OBS) {
            if (tid < 64) word = reinterpret_cast<const uint64_t *>(obs_v)[g * 64 + tid];
        } else {
            const uint4 *gin = reinterpret_cast<const uint4 *>(obs_v) + g * 1024 + 256 * wv + lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = gin[i * 64];
        }
    };
    if (first < n) { issue(first); convert(); }
    for (int64_t g = first; g < n; g += stride) {
        // ---- the grid as 64 row words in LDS
        if (TGT) {
            if (tid < 64) s.trows[tid] = tword;
        }
        if (PACKED_OBS) {
            if (tid < 64) s.rows[1 + tid] = word;
        } else {
            uint32_t *rows32 = reinterpret_cast<uint32_t *>(&s.rows[1]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if ((k & 7) == 7) rows32[2 * (16 * wv + 4 * i + q) + (k >> 3)] = xr[i];
        }
        __syncthreads();

        // ---- conv1 (table) + relu + 2x2 pool -> A1 (2py+a, 2px+b); arg-max index per channel and
        //      the 4x4 cell neighbourhood of every A1 position
        uint32_t own_bits;                            // the thread's own 4x4 cells, row-major
        {
            uint32_t seg[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const uint64_t row = s.rows[4 * py + r];
                seg[r] = (uint32_t)(px == 0 ? row << 1 : row >> (4 * px - 1)) & 0x3Fu;
            }
            own_bits = ((seg[1] >> 1) & 15u) | (((seg[2] >> 1) & 15u) << 4) | (((seg[3] >> 1) & 15u) << 8) |
                       (((seg[4] >> 1) & 15u) << 12);
            if (TGT) {                                // the error is taken against the target frame's cells
                own_bits = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) own_bits |= ((uint32_t)(s.trows[4 * py + r] >> (4 * px)) & 15u) << (4 * r);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                float4 e[2][4];                       // the eight lookups of A1 row 2py+a in flight together
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const int r = 2 * a + (d >> 1), cx = 2 * b + (d & 1);
                        const uint32_t p = ((seg[r] >> cx) & 7u) | (((seg[r + 1] >> cx) & 7u) << 3) |
                                           (((seg[r + 2] >> cx) & 7u) << 6);
                        e[b][d] = s.lut[p];
                    }
                __builtin_amdgcn_sched_barrier(0);
                float m[2][4];
                uint32_t amx[2], nbx[2];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
                    uint32_t i0 = 0, i1 = 0, i2 = 0, i3 = 0;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {     // strict >: the first maximum in scan order (ATen)
                        if (e[b][d].x > m0) { m0 = e[b][d].x; i0 = d; }
                        if (e[b][d].y > m1) { m1 = e[b][d].y; i1 = d; }
                        if (e[b][d].z > m2) { m2 = e[b][d].z; i2 = d; }
                        if (e[b][d].w > m3) { m3 = e[b][d].w; i3 = d; }
                    }
                    m[b][0] = m0; m[b][1] = m1; m[b][2] = m2; m[b][3] = m3;
                    amx[b] = i0 | (i1 << 2) | (i2 << 4) | (i3 << 6);
                    nbx[b] = ((seg[2 * a] >> (2 * b)) & 15u) | (((seg[2 * a + 1] >> (2 * b)) & 15u) << 4) |
                             (((seg[2 * a + 2] >> (2 * b)) & 15u) << 8) | (((seg[2 * a + 3] >> (2 * b)) & 15u) << 12);
                }
                const int off = (2 * py + a + 1) * kA1Stride + 2 * px + 1;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    s.a1[c * kA1Plane + off] = m[0][c];
                    s.a1[c * kA1Plane + off + 1] = m[1][c];
                }
                const int pos = (2 * py + a) * 32 + 2 * px;
                *reinterpret_cast<uint32_t *>(&s.nb[pos]) = nbx[0] | (nbx[1] << 16);
                *reinterpret_cast<uint16_t *>(&s.am[pos]) = (uint16_t)(amx[0] | (amx[1] << 8));
            }
        }
        __syncthreads();

        // ---- conv2 (4->2) + relu + 2x2 pool -> A2 (py, px); arg-max position + gate
        uint32_t flags = 0;                           // bits 0-1 / 2-3: arg-max of oc 0 / 1; 4, 5: gates; 8-11: A3 gates
        {
            kfloat *w2 = opaque_uniform(prm), *b2 = w2 + 72;
            float acc2[2][2][2];
#pragma unroll
            for (int oc = 0; oc < 2; ++oc) { acc2[oc][0][0] = 0.f; acc2[oc][0][1] = 0.f; acc2[oc][1][0] = 0.f; acc2[oc][1][1] = 0.f; }
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {          // two input channels per LDS round trip
                float patch[2][4][4];
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    const float *pl = s.a1 + (2 * ch + c2) * kA1Plane + (2 * py) * kA1Stride + 2 * px;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float2 lo = *reinterpret_cast<const float2 *>(pl + r * kA1Stride);
                        const float2 hi = *reinterpret_cast<const float2 *>(pl + r * kA1Stride + 2);
                        patch[c2][r][0] = lo.x; patch[c2][r][1] = lo.y; patch[c2][r][2] = hi.x; patch[c2][r][3] = hi.y;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2)
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
                                        acc2[oc][dy][dx] = fmaf(w2[(oc * 4 + 2 * ch + c2) * 9 + ky * 3 + kx],
                                                                patch[c2][dy + ky][dx + kx], acc2[oc][dy][dx]);
            }
#pragma unroll
            for (int oc = 0; oc < 2; ++oc) {
                float mx = acc2[oc][0][0]; uint32_t am = 0;
                if (acc2[oc][0][1] > mx) { mx = acc2[oc][0][1]; am = 1; }
                if (acc2[oc][1][0] > mx) { mx = acc2[oc][1][0]; am = 2; }
                if (acc2[oc][1][1] > mx) { mx = acc2[oc][1][1]; am = 3; }
                mx += b2[oc];
                flags |= am << (2 * oc);
                if (mx > 0.0f) flags |= 16u << oc;
                s.a2[oc * kA2Plane + (py + 1) * kA2Stride + px + 1] = fmaxf(mx, 0.0f);
            }
        }
        __syncthreads();

        // ---- convT1 (2->1, k4 s2 p1) + relu -> A3 (2py+j, 2px+i)
        {
            kfloat *wt1 = opaque_uniform(prm) + 74;
            const float bt1 = wt1[32];
            float r2[2][3][3];                        // A2 rows py-1..py+1, cols px-1..px+1 (padded)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int x = 0; x < 3; ++x)
                        r2[cc][r][x] = s.a2[cc * kA2Plane + (py + r) * kA2Stride + px + x];
            __builtin_amdgcn_sched_barrier(0);
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
                    if (a3 > 0.0f) flags |= 256u << (2 * j + i);
                    s.u.f.a3[(2 * py + j + 1) * kA1Stride + 2 * px + i + 1] = a3;
                }
        }
        __syncthreads();

        // ---- convT2 + sigmoid + loss on the thread's own 4x4 cells; backward of convT2
        float sq = 0.0f;
        float g3[2][2];                               // d loss / d A3 of the own 2x2 outputs, before the halo rows
        {
            kfloat *wt2 = opaque_uniform(prm) + 107;
            const float bt2 = wt2[16];
            float t3[4][4], dt3[4][4];                // A3 rows 2py-1..2py+2, cols 2px-1..2px+2 (padded)
            float gwt2[16];                           // this grid's convT2 weight-gradient contributions
#pragma unroll
            for (int i = 0; i < 16; ++i) gwt2[i] = 0.0f;
            const float *pl = s.u.f.a3 + (2 * py) * kA1Stride + 2 * px;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float2 lo = *reinterpret_cast<const float2 *>(pl + r * kA1Stride);
                const float2 hi = *reinterpret_cast<const float2 *>(pl + r * kA1Stride + 2);
                t3[r][0] = lo.x; t3[r][1] = lo.y; t3[r][2] = hi.x; t3[r][3] = hi.y;
                dt3[r][0] = 0.f; dt3[r][1] = 0.f; dt3[r][2] = 0.f; dt3[r][3] = 0.f;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
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
                    const float pr = fast_sigmoid(acc + bt2);
                    const float ob = (float)((own_bits >> (4 * r + cx)) & 1u);
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
            // columns: local column x is A3 column 2px-1+x; x = 1, 2 are the thread's own, x = 3 belongs to
            // px+1 (its x = 1), x = 0 to px-1 (its x = 2).  Rows likewise: r = 1, 2 own, r = 0 -> py-1 (its
            // r = 2), r = 3 -> py+1 (its r = 1).
            float c1[4], c2[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                c1[r] = dt3[r][1] + dpp_mov0<kRowShr1>(dt3[r][3]);
                c2[r] = dt3[r][2] + dpp_mov0<kRowShl1>(dt3[r][0]);
            }
            s.u.f.halo3[0][tid] = make_float2(c1[0], c2[0]);
            s.u.f.halo3[1][tid] = make_float2(c1[3], c2[3]);
            g3[0][0] = c1[1]; g3[0][1] = c2[1]; g3[1][0] = c1[2]; g3[1][1] = c2[2];
            quad_transpose_add<16>(gwt2, gwt2q, lane);
        }
        sq = wave_sum_dpp(sq);
        if (lane == 0) s.red[wv][0] = sq;
        __syncthreads();
        if (tid == 0) {
            const float b = ((s.red[0][0] + s.red[1][0]) + (s.red[2][0] + s.red[3][0])) * (1.0f / 4096.0f);
            bonus[g] = b;
            if (reward) reward[g] = reward_accumulate(reward[g], reward_scale, b);
        }

        // ---- relu gate of A3, convT1 backward (thread's own 2x2 outputs)
        // fp32 observations: the next grid's 16 registers travel during this phase and the next (few live
        // registers) and are folded into 4 before the gather phase (the register peak)
        if (!PACKED_OBS && g + stride < n) issue(g + stride);
        float g2[2];                                  // d loss / d A2 of the own entry, before the halo rows
        {
            kfloat *wt1 = opaque_uniform(prm) + 74;
            float r2[2][3][3], dr2[2][3][3];          // A2 neighbourhood re-read: cheaper than 18 live registers
            const float2 from_above = py > 0 ? s.u.f.halo3[1][tid - 16] : make_float2(0.f, 0.f);
            const float2 from_below = py < 15 ? s.u.f.halo3[0][tid + 16] : make_float2(0.f, 0.f);
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int x = 0; x < 3; ++x)
                        r2[cc][r][x] = s.a2[cc * kA2Plane + (py + r) * kA2Stride + px + x];
            __builtin_amdgcn_sched_barrier(0);
            g3[0][0] += from_above.x; g3[0][1] += from_above.y;
            g3[1][0] += from_below.x; g3[1][1] += from_below.y;
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int r = 0; r < 3; ++r) { dr2[cc][r][0] = 0.f; dr2[cc][r][1] = 0.f; dr2[cc][r][2] = 0.f; }
            float gwt1[32];                           // every entry receives exactly one product per grid
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float gz = (flags >> (8 + 2 * j + i)) & 1u ? g3[j][i] : 0.0f;
                    gbt1 += gz;
                    const int ry0 = 1, ky0 = j ? 2 : 1;
                    const int ry1 = j ? 2 : 0, ky1 = j ? 0 : 3;
                    const int cx0 = 1, kx0 = i ? 2 : 1;
                    const int cx1 = i ? 2 : 0, kx1 = i ? 0 : 3;
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc) {
                        gwt1[cc * 16 + ky0 * 4 + kx0] = gz * r2[cc][ry0][cx0];
                        gwt1[cc * 16 + ky0 * 4 + kx1] = gz * r2[cc][ry0][cx1];
                        gwt1[cc * 16 + ky1 * 4 + kx0] = gz * r2[cc][ry1][cx0];
                        gwt1[cc * 16 + ky1 * 4 + kx1] = gz * r2[cc][ry1][cx1];
                        dr2[cc][ry0][cx0] = fmaf(gz, wt1[cc * 16 + ky0 * 4 + kx0], dr2[cc][ry0][cx0]);
                        dr2[cc][ry0][cx1] = fmaf(gz, wt1[cc * 16 + ky0 * 4 + kx1], dr2[cc][ry0][cx1]);
                        dr2[cc][ry1][cx0] = fmaf(gz, wt1[cc * 16 + ky1 * 4 + kx0], dr2[cc][ry1][cx0]);
                        dr2[cc][ry1][cx1] = fmaf(gz, wt1[cc * 16 + ky1 * 4 + kx1], dr2[cc][ry1][cx1]);
                    }
                }
            // local column x is A2 column px-1+x: x = 1 own, x = 2 -> px+1 (its x = 1 ... seen from there
            // as the value of px-1), x = 0 -> px-1; rows: r = 1 own, r = 0 -> py-1, r = 2 -> py+1
            float col[2][3];
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int r = 0; r < 3; ++r)
                    col[cc][r] = (dr2[cc][r][1] + dpp_mov0<kRowShr1>(dr2[cc][r][2])) + dpp_mov0<kRowShl1>(dr2[cc][r][0]);
            s.halo2[0][tid] = make_float2(col[0][0], col[1][0]);
            s.halo2[1][tid] = make_float2(col[0][2], col[1][2]);
            g2[0] = col[0][1]; g2[1] = col[1][1];
            quad_transpose_add<32>(gwt1, gwt1q, lane);
        }
        __syncthreads();

        // ---- 2x2 pool arg-max + relu gate of A2 -> the conv2-output gradient map (overlays A3 / halo3)
        {
            const float2 from_above = py > 0 ? s.halo2[1][tid - 16] : make_float2(0.f, 0.f);
            const float2 from_below = py < 15 ? s.halo2[0][tid + 16] : make_float2(0.f, 0.f);
            g2[0] = (g2[0] + from_above.x) + from_below.x;
            g2[1] = (g2[1] + from_above.y) + from_below.y;
#pragma unroll
            for (int oc = 0; oc < 2; ++oc) {
                const float gz = (flags >> (4 + oc)) & 1u ? g2[oc] : 0.0f;
                gb2[oc] += gz;
                const uint32_t am = (flags >> (2 * oc)) & 3u;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    s.u.map2[oc * kA1Plane + (2 * py + (e >> 1) + 1) * kA1Stride + 2 * px + (e & 1) + 1] =
                        e == (int)am ? gz : 0.0f;
            }
            // the border of plane 1 that the gather reads (rows 0, 33 and columns 0, 33) held halo3 values
            if (tid < 132) {
                const int o = tid < 34 ? tid : tid < 68 ? 33 * kA1Stride + (tid - 34)
                            : tid < 100 ? (tid - 68 + 1) * kA1Stride : (tid - 100 + 1) * kA1Stride + 33;
                s.u.map2[kA1Plane + o] = 0.0f;
            }
        }
        if (!PACKED_OBS && g + stride < n) convert();
        __syncthreads();

        // ---- conv2 and conv1 weight gradients of input channel wv: lane = 16 positions of A1
        //      row Y = lane / 2, columns 16 * (lane & 1) ..
        if (PACKED_OBS && g + stride < n) issue(g + stride);
        {
            kfloat *w2 = opaque_uniform(prm);
            float wc[18];                             // w2[oc][wv][k], oc-major (SGPRs)
#pragma unroll
            for (int oc = 0; oc < 2; ++oc)
#pragma unroll
                for (int kq = 0; kq < 9; ++kq) wc[oc * 9 + kq] = w2[(oc * 4 + wvs) * 9 + kq];
            const int Y = lane >> 1, X0 = 16 * (lane & 1);
            const float *ap = s.a1 + wvs * kA1Plane + (Y + 1) * kA1Stride + X0 + 1;
            const uint16_t *nbp = &s.nb[Y * 32 + X0];
            const uint8_t *amp = &s.am[Y * 32 + X0];
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                float mp[2][3][6];                    // map rows Y..Y+2, cols X0+4ch .. +5 (padded)
#pragma unroll
                for (int oc = 0; oc < 2; ++oc)
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const float *mr = s.u.map2 + oc * kA1Plane + (Y + r) * kA1Stride + X0 + 4 * ch;
                        const float2 m01 = *reinterpret_cast<const float2 *>(mr);
                        const float2 m23 = *reinterpret_cast<const float2 *>(mr + 2);
                        const float2 m45 = *reinterpret_cast<const float2 *>(mr + 4);
                        mp[oc][r][0] = m01.x; mp[oc][r][1] = m01.y; mp[oc][r][2] = m23.x;
                        mp[oc][r][3] = m23.y; mp[oc][r][4] = m45.x; mp[oc][r][5] = m45.y;
                    }
                float av[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) av[i] = ap[4 * ch + i];
                const uint2 nb4 = *reinterpret_cast<const uint2 *>(nbp + 4 * ch);
                const uint32_t am4 = *reinterpret_cast<const uint32_t *>(amp + 4 * ch);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float d = 0.0f;
#pragma unroll
                    for (int oc = 0; oc < 2; ++oc)
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx) {
                                const float mv = mp[oc][2 - ky][i + 2 - kx];
                                gw2c[oc * 9 + ky * 3 + kx] = fmaf(av[i], mv, gw2c[oc * 9 + ky * 3 + kx]);
                                d = fmaf(wc[oc * 9 + ky * 3 + kx], mv, d);
                            }
                    const float gz = av[i] > 0.0f ? d : 0.0f;
                    gb1c += gz;
                    // the arg-max cell (dy, dx) of channel wv inside the position's 4x4 neighbourhood
                    const uint32_t idx = (am4 >> (8 * i + 2 * wvs)) & 3u;
                    const uint32_t nbw = ((i & 2) ? nb4.y : nb4.x) >> (16 * (i & 1));
                    const uint32_t t = (nbw & 0xFFFFu) >> (((idx & 2u) << 1) | (idx & 1u));
#pragma unroll
                    for (int tt = 0; tt < 9; ++tt)
                        gw1c[tt] = fmaf(gz, (float)((t >> (4 * (tt / 3) + tt % 3)) & 1u), gw1c[tt]);
                }
            }
        }
        // the next iteration's barriers order its writes to rows / A1 / nb / am / A3 / halo3 behind these reads
    }

    // ---- one partial gradient row per workgroup, parameter order of the AE blob
    float *row = partial + (int64_t)blockIdx.x * kAeParams;
    __syncthreads();
    // red[wv][.]: 0..15 convT2.w, 16 convT2.b, 17..48 convT1.w, 49 convT1.b, 50..51 conv2.b
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const float a = quad_class_sum(gwt2q[m]);
        if (lane < 4) s.red[wv][4 * m + lane] = a;
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const float a = quad_class_sum(gwt1q[m]);
        if (lane < 4) s.red[wv][17 + 4 * m + lane] = a;
    }
    {
        const float a0 = wave_sum_dpp(gbt2), a1 = wave_sum_dpp(gbt1), a2 = wave_sum_dpp(gb2[0]), a3 = wave_sum_dpp(gb2[1]);
        if (lane == 0) { s.red[wv][16] = a0; s.red[wv][49] = a1; s.red[wv][50] = a2; s.red[wv][51] = a3; }
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
        const float a = wave_sum_dpp(gw1c[i]);
        if (lane == 0) row[wv * 9 + i] = a;
    }
    {
        const float a = wave_sum_dpp(gb1c);
        if (lane == 0) row[36 + wv] = a;
    }
#pragma unroll
    for (int oc = 0; oc < 2; ++oc)
#pragma unroll
        for (int kq = 0; kq < 9; ++kq) {
            const float a = wave_sum_dpp(gw2c[oc * 9 + kq]);
            if (lane == 0) row[40 + (oc * 4 + wv) * 9 + kq] = a;
        }
}

int ae2d_train64_rows(int64_t n) {                 // = workgroups of ae_train_block64_kernel
    int64_t rows = n < 768 ? n : 768;                  // persistent: 3 workgroups are resident per CU
    return (int)(rows < 1 ? 1 : rows);
}

int ae2d_train64(const void *obs, bool packed, int64_t n, const float *ap, float *ws, float *bonus,
                 float *reward, float reward_scale, float grad_scale, float *partial, float *grad,
                 bool tables_valid, hipStream_t s, const uint64_t *tgt) {
    if (!tables_valid) hipLaunchKernelGGL(ae_prepare_kernel<true>, dim3(1), dim3(512), 0, s, ap, ws);
    const int rows = ae2d_train64_rows(n);
    const uint64_t *no_tgt = nullptr;
    if (tgt)                                          // (bit-packed frames only)
        hipLaunchKernelGGL((ae_train_block64_kernel<true, true>), dim3((unsigned)rows), dim3(256), 0, s, obs, ws, bonus,
                           reward, reward_scale, grad_scale, partial, n, tgt);
    else if (packed)
        hipLaunchKernelGGL((ae_train_block64_kernel<true, false>), dim3((unsigned)rows), dim3(256), 0, s, obs, ws, bonus,
                           reward, reward_scale, grad_scale, partial, n, no_tgt);
    else
        hipLaunchKernelGGL((ae_train_block64_kernel<false, false>), dim3((unsigned)rows), dim3(256), 0, s, obs, ws, bonus,
                           reward, reward_scale, grad_scale, partial, n, no_tgt);
    hipLaunchKernelGGL(rnd_grad_reduce_kernel, dim3((kAeParams + 31) / 32), dim3(1024), 0, s, partial, rows, grad,
                       kAeParams);
    return check_launch("carle_ae2d_train(fused 64x64)");
}

}  // namespace carle

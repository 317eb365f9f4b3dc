// ca_rnd_rows.hip -- CARLE.step AND the RND2D bonus of the fresh observation in one kernel, ONE WAVEFRONT PER GRID
// (carle/env.py:188-242 + carle/mcl.py:343-352, 398-407; 64x64 fp32 grids, no action or the full 64x64 action window).
//
// Round 4 form of the fused kernel for large batches.  The workgroup-per-grid kernel (ca_rnd_fused.hip) spends ~3 000
// wave instructions per grid, a third of them turning fp32 cells into bits and back, and walks five barrier-separated
// phases per grid; its launch ends with one grid's latency through those phases after the last byte has arrived (36 us at
// N = 4096 against 31 us of memory time).  Here lane = ROW from load to store:
//   load     a grid row is ONE load instruction (64 lanes x 4 bytes = 256 contiguous bytes); `v_cmp_neq_f32 x, 0` turns the
//            64 cells into the row's 64-bit word in an SGPR pair (NaN counts, -0.0 does not: torch.logical_xor's
//            truthiness, env.py:182) -- one VALU instruction per 64 cells instead of ~12 per 16; the action row the same
//            way, s_xor_b64 is the toggle, two v_writelane put the word into lane = row.  Chunks of 16 rows, the next
//            chunk's loads in flight while a chunk is converted, chunks taken in a per-grid rotated order (wavefronts in
//            lockstep then touch all four 4-KiB slices of their grids at any time: tools/kbench/rowwave_io.hip,
//            profiles/r04_rowwave_io.txt)
//   step     the generation with lane = row (64-bit rotates = x seam, DPP wave rotates = y seam, ca_bits.hpp)
//   store    v_readlane x 2 -> v_cndmask_b32 from the SGPR pair -> one 256-byte store per row
//   bonus    the new rows go through 528 bytes of wave-private LDS; lane (cg, ry) = (lane / 16, lane % 16) owns the four
//            4x4 pooling windows (ry, 4cg .. 4cg+3): 64 table lookups (predictor float4 + random-net float2 per 9-bit
//            pattern), maxima in registers, pooled maps through wave-private LDS planes, conv2 + pool
//            of row ry, columns 4cg .. 4cg+3 with the weights in SGPRs, the two 64 -> 16 linear layers by
//            rnd_linear_bonus (rnd_device.hpp).  No workgroup barrier after the table set-up; every wavefront runs on
//            its own.
// ~1 600 VALU instructions per grid.  One workgroup of 16 wavefronts per CU shares the tables (25 KB) and gives every
// wavefront 6.7 KB of planes: at N = 4096 every grid is resident at once, all loads are requested up front and the reward
// network runs under the drain of the stores.  Arithmetic and summation order are those of rnd_block_conv2 /
// rnd_linear_bonus: bonuses are bit-identical to the stand-alone kernel (nets_fused.hip).
#include <stdlib.h>
#include "ca_cells.hpp"
#include "rnd_device.hpp"

namespace carle {

// v_writelane_b32 through the LLVM intrinsic (this clang has no __builtin_amdgcn_writelane): the compiler sees the
// instruction and pads its hazards, which it would not do for inline asm
extern "C" __device__ int carle_llvm_writelane(int val, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

static constexpr int kRowsWaves = 16;                    // wavefronts (= grids in flight) per workgroup
static constexpr int kPRow = 20;                         // floats per plane row: 16 pooled values + 4 zeros
static constexpr int kPFloats = 18 * kPRow;              // zero row, 16 rows, zero row

struct RowsWaveShared {
    uint64_t rows[66];             // the new state, one 64-bit word per row at [1..64]; [0], [65] stay zero (conv1 padding)
    float pad[4];                  // keeps the planes 16-byte aligned
    float planes[4][kPFloats];     // pooled conv1 maps of ONE net at a time (predictor: 4 channels, random net: 2)
    float feat[2][64];             // conv2 + pool output of both nets
};
struct RowsShared {
    float4 lut_p[512];             // conv1 + relu of the predictor per 9-bit neighbourhood pattern (4 channels)
    float2 lut_r[512];             // ... of the random network (2 channels).  (One 32-byte entry per pattern would save an
                                   // address per lookup but leaves half of the LDS banks unused by the 16-byte reads:
                                   // bank conflicts doubled, SQ_LDS_BANK_CONFLICT 12.0M against 5.6M per launch at N = 4096)
    float lin[2 * kLinFloats];
    double racc[kRowsWaves][2];    // ENV: per wavefront, sum / sum of squares of its rewards
    int turn;                      // staggered start: wavefronts whose first grid's loads are all requested
    int pad_[3];
    RowsWaveShared w[kRowsWaves];
};
static_assert(offsetof(RowsWaveShared, planes) % 16 == 0 && sizeof(RowsWaveShared) % 16 == 0, "16-byte LDS accesses");
static_assert(offsetof(RowsShared, w) % 16 == 0 && offsetof(RowsShared, lin) % 16 == 0, "16-byte LDS accesses");
static_assert(sizeof(RowsShared) <= 160 * 1024, "one workgroup per CU");

struct RowRegs { float s[16], a[16]; };                  // one chunk's loads of a lane: 16 state rows + 16 action rows

// conv2 (C -> 1, 3x3, zero padding) of plane row ry, columns 4cg .. 4cg+3, for the C channels in `planes`, + 2x2 pool.
// Same arithmetic as rnd_block_conv2: per channel a chain of nine fma (ky, kx order) from zero, channels summed as
// (c0 + c1) + (c2 + c3), maximum of the window, + bias, relu.  Returns the two pooled features of pooling row ry / 2,
// columns 2cg, 2cg + 1 (valid in every lane; lanes ry and ry ^ 1 hold the same pair).
template <int C>
__device__ __forceinline__ void rows_conv2(const float (*__restrict__ planes)[kPFloats], kfloat *__restrict__ cw, float bias,
                                           int cg, int ry, float &f0, float &f1) {
    static_assert(C == 2 || C == 4, "channel pairs");
    float acc[4][4] = {};                                             // [channel][x]; C == 2: channels 2, 3 stay zero
#pragma unroll
    for (int c0 = 0; c0 < C; c0 += 2) {                               // two channels' reads in flight (36 registers)
        float in[2][3][6];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float *pl = &planes[c0 + c][ry * kPRow + 4 * cg];   // row ry - 1 (plane rows are shifted by one)
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float4 mid = *reinterpret_cast<const float4 *>(pl + r * kPRow);
                in[c][r][0] = pl[r * kPRow - 1];
                in[c][r][1] = mid.x; in[c][r][2] = mid.y; in[c][r][3] = mid.z; in[c][r][4] = mid.w;
                in[c][r][5] = pl[r * kPRow + 4];
            }
        }
        __builtin_amdgcn_sched_barrier(0);                            // the pair's LDS reads are issued before its first fma
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
                        acc[c0 + c][x] = fmaf(cw[(c0 + c) * 9 + ky * 3 + kx], in[c][ky][x + kx], acc[c0 + c][x]);
    }
    float o[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) o[x] = (acc[0][x] + acc[1][x]) + (acc[2][x] + acc[3][x]);     // C == 2: (c0 + c1) + (0 + 0)
    float h0 = fmaxf(o[0], o[1]), h1 = fmaxf(o[2], o[3]);
    // the other row of the pooling window sits in lane ^ 1
    h0 = fmaxf(h0, dpp_quad_xor1(h0));
    h1 = fmaxf(h1, dpp_quad_xor1(h1));
    f0 = fmaxf(h0 + bias, 0.0f);
    f1 = fmaxf(h1 + bias, 0.0f);
}

template <bool HAS_ACT, bool NT_ST, bool ENV>
__global__ __launch_bounds__(64 * kRowsWaves) void ca_rnd_rows_kernel(
    const float *__restrict__ u_in, const float *__restrict__ act, float *__restrict__ u_out, int64_t n,
    uint32_t rule, int32_t *__restrict__ status, int slot, const float *__restrict__ ws,
    float *__restrict__ bonus, RndEnvArgs ea, int stagger) {
    __shared__ RowsShared s;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);          // uniform: descriptors stay in SGPRs
    RowsWaveShared &w = s.w[wv];
    const int64_t stride = (int64_t)gridDim.x * kRowsWaves;
    int64_t g = (int64_t)blockIdx.x * kRowsWaves + wv;
    const uint32_t voff = (uint32_t)lane * 4u;

    // chunk c of grid `grid` (rows 16cc .. 16cc+15, cc = (c + grid) % 4): 16 + 16 loads.  `valid` false (no such grid): the
    // descriptors get a zero range, nothing is read -- the loads stay unconditional (a load inside a branch is waited for
    // inside the branch)
    auto issue = [&](int64_t grid, int c, RowRegs &r, bool valid) __attribute__((always_inline)) {
        if (!valid) grid = 0;
        const int cc = (c + (int)(grid & 3)) & 3;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char *>(reinterpret_cast<const char *>(u_in) + grid * 16384), 0, valid ? 16384 : 0, 0x00020000);
#pragma unroll
        for (int i = 0; i < 16; ++i)
            r.s[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff + 256u * i, cc * 4096, 2));
        if (HAS_ACT) {
            const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char *>(reinterpret_cast<const char *>(act) + grid * 16384), 0, valid ? 16384 : 0, 0x00020000);
#pragma unroll
            for (int i = 0; i < 16; ++i)
                r.a[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, voff + 256u * i, cc * 4096, 2));
        }
    };

    // Staggered start (stagger = 1, 2): the 16 wavefronts of a CU form four groups of four (one wavefront per SIMD each);
    // group k requests its first grid only when the groups before it have requested theirs completely (1) / received it
    // (2).  All at once, every wavefront's data would arrive together at the end of the load stream and the reward
    // networks of all 16 grids would run behind it, bound by LDS bank conflicts; staggered, the networks of the early
    // groups run under the loads of the late ones and only four grids per CU are left behind the last byte.
    const int grp = wv >> 2;
    const bool early = stagger == 0 || grp == 0;
    RowRegs r0, r1;
    issue(g, 0, r0, early && g < n);                   // the first grid travels while the tables are laid down
    issue(g, 1, r1, early && g < n);

    // ---- tables -> LDS, planes / padding rows zeroed; ONE barrier, the only one of the kernel's steady state
    {
        const float4 *lp = reinterpret_cast<const float4 *>(ws);
        const float2 *lr = reinterpret_cast<const float2 *>(ws + 2048);
        const float4 *ls = reinterpret_cast<const float4 *>(ws + kLutFloats);
        constexpr int kLin4 = 2 * kLinFloats / 4;                      // 552
        static_assert(kLin4 <= 1024 && (2 * kLinFloats) % 4 == 0, "one float4 per thread");
        const int e = tid & 511;
        const float4 p = lp[e];
        const float2 q = lr[e];
        const float4 l = ls[tid < kLin4 ? tid : 0];
        if (tid < 512) s.lut_p[e] = p; else s.lut_r[e] = q;
        if (tid < kLin4) reinterpret_cast<float4 *>(s.lin)[tid] = l;
        float4 *pz = reinterpret_cast<float4 *>(&w.planes[0][0]);
#pragma unroll
        for (int i = 0; i < (4 * kPFloats / 4 + 63) / 64; ++i)
            if (lane + 64 * i < 4 * kPFloats / 4) pz[lane + 64 * i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane == 0) { w.rows[0] = 0ull; w.rows[65] = 0ull; }
        if (lane < 4) w.pad[lane] = 0.0f;             // planes[0][-1]: the left neighbour of the zero row above row 0
        if (ENV && lane == 0) { s.racc[wv][0] = 0.0; s.racc[wv][1] = 0.0; }
        if (tid == 0) s.turn = 0;
    }
    __syncthreads();
    bool first = true;                                 // this wavefront has not passed its turn on yet
    if (!early) {                                      // wave-uniform
        for (;;) {
            int t = lane == 0 ? __hip_atomic_load(&s.turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0;
            t = __builtin_amdgcn_readfirstlane(t);
            if (t >= 4 * grp) break;
            __builtin_amdgcn_s_sleep(4);
        }
        issue(g, 0, r0, g < n);
        issue(g, 1, r1, g < n);
    }

    // ENV with pre-stored action flags (publish == 0): the reset is known before the launch
    const bool reset_all = ENV && HAS_ACT && !ea.publish && status[CARLE_ST_NOT_ALL_ONES(slot)] == 0;
    unsigned long long any_acc = 0ull, notone_acc = 0ull;             // the wavefront's action predicates over all its grids
    double rsum = 0.0, rsq = 0.0;                                      // ENV: this wavefront's rewards

    if (g >= n && stagger != 0 && lane == 0)                           // a wavefront without a grid passes its turn on
        __hip_atomic_fetch_add(&s.turn, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (g < n) {                                                    // wave-uniform
        const int rot = (int)(g & 3);
        int lo = 0, hi = 0;
        unsigned long long notone_g = 0ull;
        // cells -> row words.  (An opaque float: with the bit cast visible, hipcc rewrites x != 0.0f into v_and +
        // v_cmp_ne_u32; the predicate folds are pinned, or the scheduler parks the masks in VGPR lanes.)
        auto convert = [&](int c, RowRegs &r) __attribute__((always_inline)) {
            const int base = ((c + rot) & 3) * 16;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                asm volatile("" : "+v"(r.s[i]));
                unsigned long long x = __ballot(r.s[i] != 0.0f);
                if (HAS_ACT) {
                    asm volatile("" : "+v"(r.a[i]));
                    const unsigned long long ma = __ballot(r.a[i] != 0.0f);
                    any_acc |= ma;
                    notone_g |= __ballot(r.a[i] != 1.0f);
                    asm volatile("" : "+s"(any_acc), "+s"(notone_g));
                    x ^= ma;                                           // toggle, env.py:182
                }
                lo = carle_llvm_writelane((int)(uint32_t)x, base + i, lo);
                hi = carle_llvm_writelane((int)(uint32_t)(x >> 32), base + i, hi);
            }
        };
        convert(0, r0);
        issue(g, 2, r0, true);
        convert(1, r1);
        issue(g, 3, r1, true);
        if (first && stagger == 1) {                   // every load of the first grid is requested: the next group's turn
            if (lane == 0) __hip_atomic_fetch_add(&s.turn, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            first = false;
        }
        convert(2, r0);
        convert(3, r1);
        if (first && stagger == 2) {                   // ... received
            if (lane == 0) __hip_atomic_fetch_add(&s.turn, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            first = false;
        }
        const int64_t gn = g + stride;

        // ---- one generation, lane = row
        const uint64_t c = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
        uint64_t nx;
        {
            // (the 18 rule bits expand to 18 SGPR-pair select masks: formed here, per grid, from an opaque copy of the
            // rule -- hoisted out of the grid loop they would occupy 36 SGPRs through every other phase and spill)
            uint32_t rule_g = rule;
            asm volatile("" : "+s"(rule_g));
            const uint64_t wst = rotl64(c, 1), est = rotr64(c, 1);     // torus seam in x
            uint64_t h0, h1, p0, p1;
            row_sums(wst, c, est, h0, h1, p0, p1);
            const uint64_t a0 = dpp64_ror1(h0), a1 = dpp64_ror1(h1);   // row above (lane - 1, wraps: y seam)
            const uint64_t b0 = dpp64_rol1(h0), b1 = dpp64_rol1(h1);   // row below
            nx = apply_rule(neighbour_count(a0, a1, b0, b1, p0, p1), c, rule_g);
        }
        if (ENV && HAS_ACT) {
            // master reset (env.py:208-216): every action element of the WHOLE batch is one.  A wavefront whose own grid's
            // action is not all ones (every wavefront, under an agent) knows the answer; else the batch is asked
            bool reset = reset_all;
            if (ea.publish && __builtin_expect(notone_g == 0ull, 0))
                reset = batch_action_all_ones<F32Cells>(act, n, 64, status, slot);
            if (reset) nx = 0ull;
        }
        notone_acc |= notone_g;
        w.rows[1 + lane] = nx;

        // ---- fp32 store of the new state: row word -> SGPR pair -> 0.0f / 1.0f per lane
        {
            const int nlo = (int)(uint32_t)nx, nhi = (int)(uint32_t)(nx >> 32);
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<char *>(u_out) + g * 16384, 0, 16384, 0x00020000);
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const int cc = (c4 + rot) & 3;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const unsigned long long m =
                        ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(nhi, cc * 16 + i) << 32) |
                        (uint32_t)__builtin_amdgcn_readlane(nlo, cc * 16 + i);
                    const float o = __builtin_amdgcn_inverse_ballot_w64(m) ? 1.0f : 0.0f;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, o), ro, voff + 256u * i, cc * 4096,
                                                          NT_ST ? 2 : 0);
                }
            }
        }

        // ---- conv1 (table lookup) + relu + 4x4 max-pool: lane (cg, ry) owns windows (ry, 4cg .. 4cg+3)
        // (an opaque copy of the lane id: every per-lane address and shift of the reward network is formed here, per grid;
        // hoisted out of the grid loop they would hold ~30 VGPRs through the load phase, which needs 64 of its own)
        int lane_g = lane;
        asm volatile("" : "+v"(lane_g));
        const int cg = lane_g >> 4, ry = lane_g & 15;
        float mp[4][4], mr[4][2];                                      // [window][channel]: predictor, random net
        {
            __builtin_amdgcn_wave_barrier();                           // rows: written above by the lanes of this wavefront
            uint32_t seg[6];                                           // columns 16cg-1 .. 16cg+16 of rows 4ry-1 .. 4ry+4
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const uint64_t row = w.rows[4 * ry + k];               // rows[1 + (4ry - 1 + k)]
                const uint32_t x = cg == 0 ? (uint32_t)row << 1 : (uint32_t)(row >> (16 * cg - 1));
                seg[k] = x & 0x3FFFFu;
            }
#pragma unroll
            for (int wx = 0; wx < 4; ++wx) {
                float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f, m5 = 0.f;
#pragma unroll
                for (int b = 0; b < 2; ++b) {                          // two batches of eight lookups (48 registers)
                    float4 pa[8];
                    float2 pb[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int cidx = b * 8 + j, dy = cidx >> 2, dx = cidx & 3, sh = 4 * wx + dx;
                        const uint32_t p = ((seg[dy] >> sh) & 7u) | (((seg[dy + 1] >> sh) & 7u) << 3) |
                                           (((seg[dy + 2] >> sh) & 7u) << 6);
                        pa[j] = s.lut_p[p];
                        pb[j] = s.lut_r[p];
                    }
                    __builtin_amdgcn_sched_barrier(0);                 // the whole batch is issued before its first max
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        m0 = fmaxf(m0, pa[j].x); m1 = fmaxf(m1, pa[j].y); m2 = fmaxf(m2, pa[j].z); m3 = fmaxf(m3, pa[j].w);
                        m4 = fmaxf(m4, pb[j].x); m5 = fmaxf(m5, pb[j].y);
                    }
                }
                mp[wx][0] = m0; mp[wx][1] = m1; mp[wx][2] = m2; mp[wx][3] = m3;
                mr[wx][0] = m4; mr[wx][1] = m5;
            }
        }

        // ---- conv2 + pool, one net at a time through the wave-private planes; features -> LDS
        {
            const int off = (ry + 1) * kPRow + 4 * cg;
#pragma unroll
            for (int ch = 0; ch < 4; ++ch)
                *reinterpret_cast<float4 *>(&w.planes[ch][off]) = make_float4(mp[0][ch], mp[1][ch], mp[2][ch], mp[3][ch]);
            __builtin_amdgcn_wave_barrier();
            // conv2 parameters: pred w[36] b at +0, rand w[18] b at +40 -- scalar loads behind a pointer laundered HERE,
            // per grid (hoisted out of the grid loop the 56 values would occupy the SGPR file through every phase)
            kfloat *cw = opaque_uniform(ws + kFusedWsFloats);
            float f0, f1;
            rows_conv2<4>(w.planes, cw, cw[36], cg, ry, f0, f1);
            __builtin_amdgcn_wave_barrier();                           // planes 0, 1 are rewritten: every lane has read
#pragma unroll
            for (int ch = 0; ch < 2; ++ch)
                *reinterpret_cast<float4 *>(&w.planes[ch][off]) = make_float4(mr[0][ch], mr[1][ch], mr[2][ch], mr[3][ch]);
            __builtin_amdgcn_wave_barrier();
            float t0, t1;
            rows_conv2<2>(w.planes, cw + 40, cw[58], cg, ry, t0, t1);
            if ((ry & 1) == 0) {
                const int f = (ry >> 1) * 8 + 2 * cg;
                *reinterpret_cast<float2 *>(&w.feat[0][f]) = make_float2(f0, f1);
                *reinterpret_cast<float2 *>(&w.feat[1][f]) = make_float2(t0, t1);
            }
            __builtin_amdgcn_wave_barrier();
        }

        // the NEXT grid of this wavefront sets out (its 64 registers would not fit beside the table lookups; the other
        // wavefronts of the CU, in other phases, keep the memory pipeline busy meanwhile)
        issue(gn, 0, r0, gn < n);
        issue(gn, 1, r1, gn < n);

        // ---- the two 64 -> 16 linear layers, tanh, bonus (wave-uniform result)
        const float b = rnd_linear_bonus(w.feat, s.lin, lane_g, g, nullptr, nullptr);
        if (lane == 0) bonus[g] = b;
        if (ENV) {
            const float r = ea.reward_scale * b;                       // reward = 0 + scale * bonus (mcl.py:403-405)
            if (lane == 0 && ea.reward) ea.reward[g] = r;
            rsum += (double)r; rsq = fma((double)r, (double)r, rsq);
        }
        __builtin_amdgcn_wave_barrier();                               // feat / planes / rows are rewritten by the next grid
        g = gn;
    }

    // the bonus of the all-dead grid (used by the epilogue for every instance after a master reset) was evaluated with
    // the tables: rnd2d_prepare_tables
    if (blockIdx.x == 0 && tid == 0) bonus[n] = ws[kWsDeadBonus];
    if (HAS_ACT && (!ENV || ea.publish) && status != nullptr && lane == 0) {      // every writer stores the same value: benign
        if (any_acc != 0ull) status[CARLE_ST_ANY_NONZERO(slot)] = 1;
        if (notone_acc != 0ull) status[CARLE_ST_NOT_ALL_ONES(slot)] = 1;
    }
    if (ENV) {
        // reward moments: one (sum, sum of squares) pair per workgroup, wavefronts added in a fixed order
        if (ea.merge_moments && ea.partials) {
            if (lane == 0) { s.racc[wv][0] = rsum; s.racc[wv][1] = rsq; }
            __syncthreads();
            if (tid == 0) {
                double a = 0.0, q = 0.0;
#pragma unroll
                for (int i = 0; i < kRowsWaves; ++i) { a += s.racc[i][0]; q += s.racc[i][1]; }
                reinterpret_cast<double2 *>(ea.partials)[blockIdx.x] = make_double2(a, q);
            }
        }
        if (blockIdx.x == 0 && wv == 1) {
            // bookkeeping by ONE wavefront of the launch (see ca_step_64_kernel<..., ENV>): the previous step's pending
            // moments and counters are applied, its flag slot recycled, this step left pending
            settle_moments_wave(status, lane);
            if (lane == 0) {
                uint4 *s4 = reinterpret_cast<uint4 *>(status);
                uint4 w0 = s4[0], w1 = s4[1], w2 = s4[2];
                settle_pending(w0, w1, w2);
                if (!HAS_ACT) {
                    int64_t step_number = (int64_t)(((uint64_t)w1.w << 32) | w1.z) + 1;
                    int64_t since = (int64_t)(((uint64_t)w2.y << 32) | w2.x) + 1;
                    w1.x = 0u;
                    w1.z = (uint32_t)step_number; w1.w = (uint32_t)((uint64_t)step_number >> 32);
                    w2.x = (uint32_t)since; w2.y = (uint32_t)((uint64_t)since >> 32);
                } else {
                    w1.y = 1u | 2u | (slot ? 4u : 0u);                 // CARLE_ST_PENDING: this step
                }
                s4[1] = w1; s4[2] = w2;
                const int other = slot ^ 1;
                status[CARLE_ST_ANY_NONZERO(other)] = 0;
                status[CARLE_ST_NOT_ALL_ONES(other)] = 0;
                status[CARLE_ST_SCAN_TICKET(other)] = 0;
                status[CARLE_ST_SCAN_DONE(other)] = 0;
                if (ea.merge_moments && ea.partials && ea.reward) {
                    uint4 w5;
                    w5.x = (uint32_t)reinterpret_cast<uint64_t>(ea.partials);
                    w5.y = (uint32_t)(reinterpret_cast<uint64_t>(ea.partials) >> 32);
                    w5.z = gridDim.x;
                    w5.w = (uint32_t)n;
                    s4[5] = w5;                                        // CARLE_ST_MOM_*: merged by the next call
                }
            }
        }
    }
}

// workgroups of the wavefront-per-grid launch: one per CU (the LDS image admits one), each wavefront strides over grids
int64_t rnd_rows_blocks(int64_t n) {
    int64_t blocks = (n + kRowsWaves - 1) / kRowsWaves;
    const int64_t cap = 256;
    if (blocks > cap) blocks = cap;
    return blocks < 1 ? 1 : blocks;
}

// same contract as ca_step_rnd64 (ca_rnd_fused.hip); action == nullptr or the full 64x64 window
int ca_step_rnd_rows(const float *u_in, const float *action, float *u_out, int64_t n, uint32_t rule, int32_t *status, int slot,
                     const float *ws, float *bonus, bool nt_st, hipStream_t s, const RndEnvArgs *env) {
    const dim3 grid((unsigned)rnd_rows_blocks(n)), block(64 * kRowsWaves);
    const RndEnvArgs ea = env ? *env : RndEnvArgs();
    const int64_t sk = knobs().rnd_rows_stagger;       // experiment knob, read once
    const int stagger = sk >= 0 ? (int)sk : 1;
#define CARLE_ROWS_LAUNCH(ACT, NTS, ENVF) \
    hipLaunchKernelGGL((ca_rnd_rows_kernel<ACT, NTS, ENVF>), grid, block, 0, s, u_in, action, u_out, n, rule, status, slot, ws, \
                       bonus, ea, stagger)
#define CARLE_ROWS_PICK(ENVF) \
    do { \
        if (action && nt_st) CARLE_ROWS_LAUNCH(true, true, ENVF); \
        else if (action) CARLE_ROWS_LAUNCH(true, false, ENVF); \
        else if (nt_st) CARLE_ROWS_LAUNCH(false, true, ENVF); \
        else CARLE_ROWS_LAUNCH(false, false, ENVF); \
    } while (0)
    if (env) CARLE_ROWS_PICK(true);
    else CARLE_ROWS_PICK(false);
#undef CARLE_ROWS_PICK
#undef CARLE_ROWS_LAUNCH
    return check_launch("carle_env_step_f32_rnd(CA + RND2D, wavefront per grid)");
}

}  // namespace carle

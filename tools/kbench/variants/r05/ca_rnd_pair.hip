// ca_rnd_pair.hip -- the one-launch CA + RND2D step (carle/env.py:188-242 + carle/mcl.py:343-352, 398-407; see
// ca_rnd_fused.hip) with conv1 read from a PAIR table: ONE workgroup of sixteen wavefronts per CU, four grids in flight.
//
// Why (round 5, EXPERIMENTS.md 5.2): at N = 4096 the four-wave kernel spends 2 500 LDS cycles per grid, 1 370 of them bank
// conflicts of its 4 096 random 16 + 8 byte table reads, and ~900 VALU instructions on the reads' addresses and maxima.  A
// timing probe that issues half the lookups (wrong planes) runs 36.7 -> 34.4 us.  Half the lookups with RIGHT planes is a
// table indexed by the 3 x 4 neighbourhood of TWO horizontally adjacent cells -- 4 096 entries holding the channel-wise
// maximum of the two cells' conv1 + relu outputs (the 4 x 4 max-pool that follows takes the maximum anyway; max is exact, so
// the planes are bit-identical): eight lookups per pooling window instead of sixteen.  96 KB -- not four times per CU.
// Hence one 1024-thread workgroup per CU that holds the tables ONCE and runs four independent GROUPS of four wavefronts,
// each the four-wave kernel's workgroup: a grid from load to bonus, three barriers per grid.  s_barrier spans the whole
// workgroup, so a group synchronises through a counter in LDS (group_barrier below): the four groups stay in different
// phases, which is what lets the memory phase of one overlap the LDS / VALU phases of the others.
#include <stdlib.h>
#include "ca_cells.hpp"
#include "rnd_device.hpp"

namespace carle {

// the pair tables behind the fused workspace (rnd2d_prepare_tables builds them with the 512-entry tables)
static constexpr int kPairEntries = 4096;
static constexpr int kWsPairP = kFusedWsTotal;                       // float4 per entry (predictor, 4 channels)
static constexpr int kWsPairR = kWsPairP + 4 * kPairEntries;         // float2 per entry (random network, 2 channels)
static constexpr int kPairWsTotal = kWsPairR + 2 * kPairEntries;
static_assert(kWsPairP % 4 == 0 && kWsPairR % 4 == 0, "16-byte aligned tables");

int64_t rnd2d_pair_workspace_floats() { return kPairWsTotal; }

// entry q = a | b << 4 | c << 8 (a, b, c: columns x-1 .. x+2 of the rows above, of and below the two cells x, x + 1):
// the channel-wise maximum of the two cells' 512-entry table rows
__global__ __launch_bounds__(256) void rnd_pair_prepare_kernel(float *__restrict__ ws) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int a = q & 15, b = (q >> 4) & 15, c = (q >> 8) & 15;
    const int pa = (a & 7) | ((b & 7) << 3) | ((c & 7) << 6);
    const int pb = ((a >> 1) & 7) | (((b >> 1) & 7) << 3) | (((c >> 1) & 7) << 6);
    const float4 x = reinterpret_cast<const float4 *>(ws)[pa], y = reinterpret_cast<const float4 *>(ws)[pb];
    const float2 u = reinterpret_cast<const float2 *>(ws + 2048)[pa], v = reinterpret_cast<const float2 *>(ws + 2048)[pb];
    reinterpret_cast<float4 *>(ws + kWsPairP)[q] = make_float4(fmaxf(x.x, y.x), fmaxf(x.y, y.y), fmaxf(x.z, y.z), fmaxf(x.w, y.w));
    reinterpret_cast<float2 *>(ws + kWsPairR)[q] = make_float2(fmaxf(u.x, v.x), fmaxf(u.y, v.y));
}

void rnd2d_prepare_pair_tables(float *ws, hipStream_t s) {
    hipLaunchKernelGGL(rnd_pair_prepare_kernel, dim3(kPairEntries / 256), dim3(256), 0, s, ws);
}

struct PairTables {                      // one copy per workgroup = per CU
    float4 pair_p[kPairEntries];         // 64 KB
    float2 pair_r[kPairEntries];         // 32 KB
    float lin[2 * kLinFloats];
    float w2[2][4][12];
    float b2[4];
    uint32_t bar[4];                     // arrivals at the group barriers, one counter per group (only ever grows)
    uint32_t stuck;                      // a group barrier gave up (never, unless a wavefront was lost): results are void
};

struct PairGroup {                       // one grid in flight (four wavefronts)
    float planes[6][kPlaneFloats];
    uint64_t rows[66];
    uint32_t rows_in[64][2];
    float feat[2][64];
    uint32_t notone[4];
    double racc[2];
};

struct PairShared {
    PairTables t;
    PairGroup g[4];
};
static_assert(sizeof(PairShared) <= 160 * 1024, "one workgroup per CU");

// conv1 (pair table) + relu + 4x4 max-pool of all six channels; thread = pooling window (py = ltid / 16, px = ltid % 16):
// rnd_block_conv1 with two cells per lookup
__device__ __forceinline__ void rnd_pair_conv1(const PairTables &t, PairGroup &g, int ltid) {
    const int py = ltid >> 4, px = ltid & 15;
    uint32_t seg[6];                                  // [west | 4 cells | east] of rows 4py-1..4py+4
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const uint64_t row = g.rows[4 * py + r];
        seg[r] = (uint32_t)(px == 0 ? row << 1 : row >> (4 * px - 1)) & 0x3Fu;
    }
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f, m5 = 0.f;
#pragma unroll
    for (int b = 0; b < 2; ++b) {                     // two batches of four lookups
        float4 pa[4];
        float2 pb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int dy = 2 * b + (j >> 1), sh = 2 * (j & 1);          // cells (dy, 0..1) or (dy, 2..3)
            const uint32_t q = ((seg[dy] >> sh) & 15u) | (((seg[dy + 1] >> sh) & 15u) << 4) | (((seg[dy + 2] >> sh) & 15u) << 8);
            pa[j] = t.pair_p[q];
            pb[j] = t.pair_r[q];
        }
        __builtin_amdgcn_sched_barrier(0);            // the whole batch is issued before its first max
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            m0 = fmaxf(m0, pa[j].x); m1 = fmaxf(m1, pa[j].y); m2 = fmaxf(m2, pa[j].z); m3 = fmaxf(m3, pa[j].w);
            m4 = fmaxf(m4, pb[j].x); m5 = fmaxf(m5, pb[j].y);
        }
    }
    const int off = (py + 1) * kPlaneStride + px + 1;
    g.planes[0][off] = m0;
    g.planes[1][off] = m1;
    g.planes[2][off] = m2;
    g.planes[3][off] = m3;
    g.planes[4][off] = m4;
    g.planes[5][off] = m5;
}

// rnd_block_conv2 on a group's planes (same arithmetic, same order)
__device__ __forceinline__ void rnd_pair_conv2(const PairTables &t, PairGroup &g, int ltid) {
    const int f = ltid >> 2, c = ltid & 3;
    const int fy = f >> 3, fx = f & 7;
    float patch[2][4][4], w[2][12];
#pragma unroll
    for (int net = 0; net < 2; ++net) {
        const float *pl = &g.planes[net == 0 ? c : 4 + (c & 1)][(2 * fy) * kPlaneStride + 2 * fx];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float2 lo = *reinterpret_cast<const float2 *>(pl + r * kPlaneStride);
            const float2 hi = *reinterpret_cast<const float2 *>(pl + r * kPlaneStride + 2);
            patch[net][r][0] = lo.x; patch[net][r][1] = lo.y; patch[net][r][2] = hi.x; patch[net][r][3] = hi.y;
        }
        const float4 *wp = reinterpret_cast<const float4 *>(&t.w2[net][c][0]);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float4 x = wp[i];
            w[net][4 * i] = x.x; w[net][4 * i + 1] = x.y; w[net][4 * i + 2] = x.z; w[net][4 * i + 3] = x.w;
        }
    }
    const float bias0 = t.b2[0], bias1 = t.b2[1];
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
                acc += dpp_quad_xor1(acc);
                acc += dpp_quad_xor2(acc);
                o[dy][dx] = acc;
            }
        const float mx = fmaxf(fmaxf(o[0][0], o[0][1]), fmaxf(o[1][0], o[1][1])) + (net == 0 ? bias0 : bias1);
        if (c == 0) g.feat[net][f] = fmaxf(mx, 0.0f);
    }
}

// Barrier of ONE group (four wavefronts of the sixteen): every wavefront adds one to the group's counter and waits until
// the counter has reached four times the number of barriers the group has passed.  LDS executes a wavefront's operations
// in order, so the add follows the wavefront's earlier LDS writes and reads; the others' reads behind their wait see them.
// Only LDS traffic is ordered (lgkmcnt), the global loads in flight for the next grid are left alone.  All sixteen
// wavefronts are resident, so a waiting wavefront only ever waits for running ones; the wait gives up after ~2^22 polls
// (a lost wavefront would otherwise hang the GPU) and flags the launch.
__device__ __forceinline__ void group_barrier(PairTables &t, int grp, uint32_t &target, int lane) {
    target += 4u;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(&t.bar[grp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (int it = 0; it < (1 << 22); ++it) {
        const uint32_t v = __builtin_amdgcn_readfirstlane(
            __hip_atomic_load(&t.bar[grp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if ((int32_t)(v - target) >= 0) {
            asm volatile("" ::: "memory");
            return;
        }
#ifndef CARLE_PAIR_NOSLEEP
        __builtin_amdgcn_s_sleep(1);
#endif
    }
    if (lane == 0) t.stuck = 1u;
    asm volatile("" ::: "memory");
}

struct Quads4 { uint4 v[4], a[4]; };     // one grid's loads of a lane: 4 state + 4 action quads

// the ENV instantiation only (the whole RND2D.step in one launch): see ca_step_rnd64_kernel<..., ENV>
template <bool HAS_ACT, bool NT_ST = true, bool NT_LD = true>
__global__ __launch_bounds__(1024, 4) void ca_step_rnd64_pair_kernel(
    const float *__restrict__ u_in, const float *__restrict__ act, float *__restrict__ u_out, int64_t n,
    int aw, uint32_t rule, int32_t *__restrict__ status, int slot, const float *__restrict__ ws,
    float *__restrict__ bonus, RndEnvArgs ea) {
    __shared__ PairShared s;
    const int tid = threadIdx.x;
    const int grp = __builtin_amdgcn_readfirstlane(tid >> 8);      // group of this wavefront (uniform)
    const int ltid = tid & 255, lane = tid & 63, wv = (tid >> 6) & 3;
    PairTables &T = s.t;
    PairGroup &G = s.g[grp];
    const bool reset_all = HAS_ACT && !ea.publish && status[CARLE_ST_NOT_ALL_ONES(slot)] == 0;
    bool any_acc = false, notone_acc = false;
    const int q = lane >> 4, k = lane & 15;
    const int o = (64 - aw) >> 1;
    const int ax = 4 * k - o;
    uint32_t rows_in_win = 0;
    if (HAS_ACT && ax >= 0 && ax < aw) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ay = 16 * wv + 4 * i + q - o;
            if (ay >= 0 && ay < aw) rows_in_win |= 1u << i;
        }
    }
    const uint32_t voff = (uint32_t)(256 * wv + lane) * 16u;
    const uint32_t abase = (HAS_ACT && ax >= 0 && ax < aw) ? (uint32_t)((16 * wv + q - o) * aw + ax) * 4u : 0x40000000u;
    uint32_t arow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ay0 = __builtin_amdgcn_readfirstlane(16 * wv) + 4 * i - o;
        arow[i] = (ay0 >= 0 && ay0 < aw) ? (uint32_t)(4 * i * aw) * 4u : 0x40000000u;
    }
    const uint32_t act_bytes = (uint32_t)(aw * aw) * 4u;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    auto issue = [&](int64_t g, Quads4 &r, bool valid) __attribute__((always_inline)) {
        if (!valid) g = 0;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char *>(reinterpret_cast<const char *>(u_in) + g * 16384), 0, valid ? 16384 : 0, 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 1024u * i, 0, NT_LD ? 2 : 0);
            r.v[i] = make_uint4(t.x, t.y, t.z, t.w);
        }
        if (HAS_ACT) {
            const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char *>(reinterpret_cast<const char *>(act) + g * (int64_t)act_bytes), 0,
                valid ? act_bytes : 0u, 0x00020000);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(ra, abase + arow[i], 0, 2);
                r.a[i] = make_uint4(t.x, t.y, t.z, t.w);
            }
        }
    };

    const int64_t first = (int64_t)blockIdx.x * 4 + grp, stride = (int64_t)gridDim.x * 4;
    Quads4 r0, r1;
    issue(first, r0, first < n);                      // the first grid travels while the tables are laid down

    // ---- tables -> LDS by all 1024 threads, every load requested before the first LDS store
    {
        const float4 *srcp = reinterpret_cast<const float4 *>(ws + kWsPairP);
        const float4 *srcr = reinterpret_cast<const float4 *>(ws + kWsPairR);     // 4096 float2 = 2048 float4
        const float4 *lsrc = reinterpret_cast<const float4 *>(ws + kLutFloats);
        constexpr int kLin4 = 2 * kLinFloats / 4;                                 // 552
        float4 p[4], r[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) p[i] = srcp[tid + 1024 * i];
#pragma unroll
        for (int i = 0; i < 2; ++i) r[i] = srcr[tid + 1024 * i];
        const int li = tid < kLin4 ? tid : tid - 512;                             // tail: a benign duplicate
        const float4 l = lsrc[li];
        const int e = tid % 96, net = e / 48, c = (e % 48) / 12, t12 = e % 12;
        const bool w2_in = t12 < 9 && (net == 0 || c < 2);
        const float w2v = ws[kFusedWsFloats + (w2_in ? (net == 0 ? 0 : 40) + c * 9 + t12 : 0)];
        const int be = tid & 3;
        const float b2v = ws[kFusedWsFloats + (be == 1 ? 58 : 36)];
#pragma unroll
        for (int i = 0; i < 4; ++i) T.pair_p[tid + 1024 * i] = p[i];
#pragma unroll
        for (int i = 0; i < 2; ++i) reinterpret_cast<float4 *>(T.pair_r)[tid + 1024 * i] = r[i];
        reinterpret_cast<float4 *>(T.lin)[li] = l;
        (&T.w2[0][0][0])[e] = w2_in ? w2v : 0.0f;
        T.b2[be] = be < 2 ? b2v : 0.0f;
        if (tid < 4) T.bar[tid] = 0u;
        if (tid == 0) T.stuck = 0u;
        float4 *pl = reinterpret_cast<float4 *>(&G.planes[0][0]);
        for (int i = ltid; i < 6 * kPlaneFloats / 4; i += 256) pl[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ltid < 66) G.rows[ltid] = 0ull;
        if (ltid == 64) { G.racc[0] = 0.0; G.racc[1] = 0.0; }
    }
    __syncthreads();                                  // the only workgroup-wide barrier
    uint32_t bar_target = 0u;
    int64_t pending = -1;

    auto step = [&](int64_t g, Quads4 &cur, Quads4 &nxt) __attribute__((always_inline)) {
        issue(g + stride, nxt, g + stride < n);
        // ---- A: nibbles of this lane's four rows, toggled; row words by OR over the 16 lanes
        uint32_t x[4];
        ActionSeen seen_g;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t nib = nibble_fast(cur.v[i]);
            if (HAS_ACT) {
                if ((rows_in_win >> i) & 1u) seen_g.add(cur.a[i]);
                nib ^= nibble_fast(cur.a[i]);         // toggle, env.py:182
            }
            x[i] = nib << (4 * (k & 7));
        }
        bool w_notone = true;
        if (HAS_ACT && ea.publish) {
            w_notone = __ballot(seen_g.not_all_ones<F32Cells>()) != 0ull;
            any_acc |= __ballot(seen_g.any_nonzero<F32Cells>()) != 0ull;
            notone_acc |= w_notone;
            if (lane == 0 && wv != 0) G.notone[wv] = w_notone ? 1u : 0u;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] |= __builtin_amdgcn_update_dpp(0u, x[i], 0x111, 0xF, 0xF, true);   // row_shr:1
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] |= __builtin_amdgcn_update_dpp(0u, x[i], 0x112, 0xF, 0xF, true);   // row_shr:2
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] |= __builtin_amdgcn_update_dpp(0u, x[i], 0x114, 0xF, 0xF, true);   // row_shr:4
        if ((k & 7) == 7) {
#pragma unroll
            for (int i = 0; i < 4; ++i) G.rows_in[16 * wv + 4 * i + q][k >> 3] = x[i];
        }
        group_barrier(T, grp, bar_target, lane);

        // ---- B: one generation, lane = row (wave 0); the previous grid's linear layers + bonus (wave 1)
        if (wv == 0) {
            const uint64_t c = ((uint64_t)G.rows_in[lane][1] << 32) | G.rows_in[lane][0];
            const uint64_t wst = rotl64(c, 1), est = rotr64(c, 1);
            uint64_t h0, h1, p0, p1;
            row_sums(wst, c, est, h0, h1, p0, p1);
            const uint64_t a0 = dpp64_ror1(h0), a1 = dpp64_ror1(h1);
            const uint64_t b0 = dpp64_rol1(h0), b1 = dpp64_rol1(h1);
            uint32_t rule_g = rule;
            asm volatile("" : "+s"(rule_g));
            uint64_t nx = apply_rule(neighbour_count(a0, a1, b0, b1, p0, p1), c, rule_g);
            if (HAS_ACT) {
                bool reset = reset_all;
                if (ea.publish && __builtin_expect(!w_notone, 0)) {
                    const bool own_notone = (G.notone[1] | G.notone[2] | G.notone[3]) != 0u;
                    if (!own_notone) reset = batch_action_all_ones<F32Cells>(act, n, aw, status, slot);
                }
                if (reset) nx = 0ull;
            }
            G.rows[1 + lane] = nx;
            if (ea.bits_out) ea.bits_out[g * 64 + lane] = nx;
        } else if (wv == 1 && pending >= 0) {
            const float b = rnd_linear_bonus(G.feat, T.lin, lane, pending, nullptr, nullptr);
            if (lane == 0) {
                bonus[pending] = b;
                const float r = ea.reward_scale * b;              // reward = 0 + scale * bonus (mcl.py:403-405)
                if (ea.reward) ea.reward[pending] = r;
                G.racc[0] += (double)r; G.racc[1] = fma((double)r, (double)r, G.racc[1]);
            }
        }
        group_barrier(T, grp, bar_target, lane);

        // ---- C: fp32 store of the new state
        {
            const uint32_t *rows32 = reinterpret_cast<const uint32_t *>(&G.rows[1]);
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<char *>(u_out) + g * 16384, 0, 16384, 0x00020000);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t w = rows32[2 * (16 * wv + 4 * i + q) + (k >> 3)];
                const uint32_t by = (((w >> (4 * (k & 7))) & 0xFu) * 0x00204081u) & 0x01010101u;
                u32x4 o4;
                o4.x = __float_as_uint((float)((by >> 0) & 0xFFu));
                o4.y = __float_as_uint((float)((by >> 8) & 0xFFu));
                o4.z = __float_as_uint((float)((by >> 16) & 0xFFu));
                o4.w = __float_as_uint((float)((by >> 24) & 0xFFu));
                __builtin_amdgcn_raw_buffer_store_b128(o4, ro, voff + 1024u * i, 0, NT_ST ? 2 : 0);
            }
        }
        // ---- D: conv1 (pair table) + relu + 4x4 max-pool
        rnd_pair_conv1(T, G, ltid);
        group_barrier(T, grp, bar_target, lane);
        // ---- E1: conv2 + pool -> feat (read by wave 1 behind the next grid's first barrier)
        rnd_pair_conv2(T, G, ltid);
        pending = g;
    };

    for (int64_t g = first; g < n;) {                 // unrolled by two: the register sets swap roles
        step(g, r0, r1);
        g += stride;
        if (g >= n) break;
        step(g, r1, r0);
        g += stride;
    }
    if (blockIdx.x == 0 && tid == 0) bonus[n] = ws[kWsDeadBonus];
    if (pending >= 0) {                               // the last grid's linear layers
        group_barrier(T, grp, bar_target, lane);
        if (wv == 1) {
            const float b = rnd_linear_bonus(G.feat, T.lin, lane, pending, nullptr, nullptr);
            if (lane == 0) {
                bonus[pending] = b;
                const float r = ea.reward_scale * b;
                if (ea.reward) ea.reward[pending] = r;
                G.racc[0] += (double)r; G.racc[1] = fma((double)r, (double)r, G.racc[1]);
            }
        }
    }
    if (HAS_ACT && ea.publish && lane == 0) {
        if (any_acc) status[CARLE_ST_ANY_NONZERO(slot)] = 1;
        if (notone_acc) status[CARLE_ST_NOT_ALL_ONES(slot)] = 1;
    }
    if (wv == 1 && lane == 0 && ea.merge_moments && ea.partials)
        reinterpret_cast<double2 *>(ea.partials)[blockIdx.x * 4 + grp] = make_double2(G.racc[0], G.racc[1]);
    if (blockIdx.x == 0 && grp == 0 && wv == 2) {
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
                w1.y = 1u | 2u | (slot ? 4u : 0u);
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
                w5.z = gridDim.x * 4u;
                w5.w = (uint32_t)n;
                s4[5] = w5;
            }
        }
    }
}

// validated by the caller; the one-launch step only.  256 workgroups (one per CU) x 4 groups walk all grids.
int ca_step_rnd64_pair(const float *u_in, const float *action, float *u_out, int64_t n, int aw, uint32_t rule,
                       int32_t *status, int slot, float *ws, float *bonus, hipStream_t s, const RndEnvArgs &ea,
                       bool nt_st, bool cached_ld) {
    // (groups <= n and <= the partial pairs carle_rnd_step_partials_doubles(n) provides: n / 4 workgroups, rounded down)
    int64_t blocks = n / 4 < 256 ? n / 4 : 256;
    if (blocks < 1) blocks = 1;
    const dim3 grid((unsigned)blocks), block(1024);
    const int awk = action ? aw : 64;
#define CARLE_PAIR_LAUNCH(ACT, NTS, NTL) \
    hipLaunchKernelGGL((ca_step_rnd64_pair_kernel<ACT, NTS, NTL>), grid, block, 0, s, u_in, action, u_out, n, awk, rule, status, \
                       slot, ws, bonus, ea)
    if (action && cached_ld) CARLE_PAIR_LAUNCH(true, false, false);
    else if (action && nt_st) CARLE_PAIR_LAUNCH(true, true, true);
    else if (action) CARLE_PAIR_LAUNCH(true, false, true);
    else if (nt_st) CARLE_PAIR_LAUNCH(false, true, true);
    else CARLE_PAIR_LAUNCH(false, false, true);
#undef CARLE_PAIR_LAUNCH
    return check_launch("carle_env_step_f32_rnd(pair tables)");
}

}  // namespace carle

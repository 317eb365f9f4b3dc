// ca_rnd_fused.hip -- CARLE.step AND the RND2D bonus of the fresh observation in ONE kernel
// (carle/env.py:188-242 + carle/mcl.py:343-352, 398-407) for 64x64 fp32 grids.
//
// The CA generation is HBM-bound (48 KiB of traffic per grid), the RND2D forward is LDS / VALU
// bound (one table lookup per cell and net); as two kernels on two streams they fight for wave
// slots and LDS and run no faster than back to back (profiles/r01_cfg4_rnd_kernel_stats.csv).
// Here ONE WORKGROUP (4 waves) owns a grid from load to bonus, the new state never leaves the
// CU, and 4 workgroups per CU are in different phases, so the memory phase of one overlaps the
// LDS phase of the others:
//   A  every lane loads 4 + 4 quads (fp32 state / action rows 16w+4i+q, columns 4k..4k+3; the
//      quads of the workgroup's NEXT grid are requested before phase D: register prefetch),
//      toggles, and the 16 lanes of a row OR their nibbles into the row's 64-bit word in LDS;
//   B  wave 0 runs the generation with lane = row (64-bit rotates for west/east, lane +-1 for
//      north/south) and leaves the new rows in LDS;
//   C  every lane expands its nibbles of the new rows into fp32 quads and stores them (nt);
//   D  thread = one 4x4 pooling window: six row segments out of the LDS rows, 16 table lookups
//      for the six conv1 channels (rnd_device.hpp), max -> the zero-bordered planes;
//   E  all threads: conv2 + pooling (rnd_block_conv2); the two 64->16 linear layers, tanh and the
//      bonus (rnd_linear_bonus) run on wave 1 during the NEXT grid's phase B, beside wave 0.
// Three barriers per grid.  Every LDS-reading phase issues all its reads before the first use:
// with 16 waves on the CU a dependent LDS round trip costs 250-500 cycles (profiles/r02_rnd_stamps.txt).
// The bonus of the ALL-DEAD grid is copied to bonus[n] (it is evaluated together with the tables,
// nets_fused.hip: rnd2d_prepare_tables): the epilogue (ca_finish_kernel) uses it for every
// instance when the step turns out to be a whole-batch master reset (env.py:208-216).
#include <stdlib.h>
#include "ca_cells.hpp"
#include "rnd_device.hpp"

// Diagnostic build only (tools/kbench/ca_rnd_variants.hip defines CARLE_RND_STAMPS): wave 0 of every
// workgroup records s_memtime at the phase boundaries of each grid into a debug buffer that nothing
// else reads.  The shipped library compiles these to nothing.
#ifdef CARLE_RND_STAMPS
__device__ unsigned long long *g_rnd_stamps = nullptr;      // [block][iteration < 8][16]
#define RND_STAMP(i) do { if (tid == 0 && g_rnd_stamps && it_ < 8) \
    g_rnd_stamps[((size_t)blockIdx.x * 8 + it_) * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#define RND_STAMP_NEXT() (++it_)
#define RND_STAMP_DECL() int it_ = 0
#else
#define RND_STAMP(i) do {} while (0)
#define RND_STAMP_NEXT() do {} while (0)
#define RND_STAMP_DECL() do {} while (0)
#endif

namespace carle {

// RndBlockShared (rnd_device.hpp) + the toggled input rows + the pooled conv2 features of the grid
// whose linear layers are still pending: 37.6 KB, four workgroups per CU (128 VGPRs).
struct CaRndShared {
    RndBlockShared net;
    uint32_t rows_in[64][2];     // toggled state, one 64-bit word per row (lo, hi)
    float feat[2][64];           // conv2 + pool output of both nets (read one phase later, see E)
    uint32_t notone[4];          // ENV: per wavefront, some action element of the current grid differs from one
    double racc[2];              // ENV: sum / sum of squares of this workgroup's rewards (wave 1, lane 0)
};

// 1 if the fp32 bit pattern is a non-zero NUMBER (NaN counts, -0.0 does not; the truthiness of
// torch.logical_xor, env.py:182) in two VALU ops: the add drops the sign, v_min_u32 clamps to 1.
// (Written as asm: hipcc rewrites umin(x, 1) into compare + select, three ops and a VCC hazard.)
__device__ __forceinline__ uint32_t nz1(uint32_t bits) {
    uint32_t r;
    const uint32_t t = bits + bits;
    asm("v_min_u32 %0, 1, %1" : "=v"(r) : "v"(t));
    return r;
}
__device__ __forceinline__ uint32_t nibble_fast(const uint4 &v) {
    return nz1(v.x) | (nz1(v.y) << 1) | (nz1(v.z) << 2) | (nz1(v.w) << 3);
}

struct Quads { uint4 v[4], a[4]; };     // one grid's loads of a lane: 4 state + 4 action quads

// ENV instantiation (CARLE_STEP_DEFERRED): the whole RND2D.step -- CARLE.step + bonus + reward -- in this ONE launch, no
// epilogue kernel behind it.  As ca_step_64_kernel<..., ENV> (ca_step_cells.hip), plus:
//   master reset    per grid: the four wavefronts combine "some action element of this grid differs from one" through
//                   LDS behind the barrier they share anyway; an all-ones grid asks batch_action_all_ones().  A reset
//                   grid's new rows are zero, so phases C-E store zeros and evaluate the bonus of the all-dead grid by
//                   themselves (mcl.py:403-405 on the observation of env.py:216).
//   reward          reward[g] = reward_scale * bonus[g], written where the bonus is formed
//   reward moments  the rewards exist only when the launch ends: every workgroup leaves (sum, sum of squares) of its
//                   rewards in `partials`; the next launch on the status block (or carle_env_settle) merges them
//                   (settle_moments_wave, ca_cells.hpp)
// (struct RndEnvArgs: common.hpp)

#ifndef CARLE_RND_WG_PER_CU
#define CARLE_RND_WG_PER_CU 4
#endif

// NT_ST: streaming stores (state arrays beyond the Infinity Cache); NT_LD: streaming hint on the STATE loads (the action
// loads always carry it) -- up to 128 MiB per state array the re-read state comes out of the caches faster without it
// (N=4096: 36.7 -> 36.0 us, N=8192: 65.2 -> 63.7; N=16384: 122 -> 134 us, so larger batches keep the hint)
// PACKED_IN (round 5; with ENV and BITS): `u_in` points at the bit-packed copy the previous launch left of the state (64 row
// words per grid) instead of the fp32 cells -- a lane then loads the four row dwords its phase-A stores need (4 B each, the
// same address for the 8 lanes that build a dword) where it loaded four 16-byte quads, and phase A converts the action only.
template <bool HAS_ACT, bool NT_ST = true, bool NT_LD = true, bool ENV = false, bool BITS = false, bool PACKED_IN = false>
__global__ __launch_bounds__(256, CARLE_RND_WG_PER_CU) void ca_step_rnd64_kernel(
    const float *__restrict__ u_in, const float *__restrict__ act, float *__restrict__ u_out, int64_t n,
    int aw, uint32_t rule, int32_t *__restrict__ status, int slot, const float *__restrict__ ws,
    float *__restrict__ bonus, RndEnvArgs ea = RndEnvArgs()) {
    __shared__ CaRndShared s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // ENV with pre-stored action flags (publish == 0): the reset is known before the launch
    const bool reset_all = ENV && HAS_ACT && !ea.publish && status[CARLE_ST_NOT_ALL_ONES(slot)] == 0;
    bool any_acc = false, notone_acc = false;         // ENV: the wavefront's action predicates over all its grids (uniform)
    if (ENV && tid == 64) { s.racc[0] = 0.0; s.racc[1] = 0.0; }
    const int q = lane >> 4, k = lane & 15;
    // action window: offset o = (64-aw)/2 is a multiple of 4 (aw % 8 == 0): a lane's four columns
    // are all inside or all outside; bit i of `rows_in_win` = row 16w+4i+q is inside
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

    // Loads go through buffer descriptors built per grid in SGPRs (base = the grid / its action,
    // 16 KiB / aw*aw*4 bytes of range): a lane contributes one loop-invariant 32-bit byte offset per
    // load, so no 64-bit pointer per row lives in vector registers, and rows outside the action
    // window use an out-of-range offset -- the range check returns zeros, no branch.
    const uint32_t voff = (uint32_t)(256 * wv + lane) * 16u;
    // action rows: a lane's byte offset of load i is abase + arow[i] (one VGPR + four SGPRs; the sum is
    // formed in 32-bit VALU arithmetic, so a negative row of abase wraps back correctly).  Columns
    // outside the window (per lane) and rows outside it (per wave and i: o is a multiple of 4) add
    // 0x40000000 each, which lands beyond the descriptor's range whatever the other term is.
    const uint32_t abase = (HAS_ACT && ax >= 0 && ax < aw) ? (uint32_t)((16 * wv + q - o) * aw + ax) * 4u : 0x40000000u;
    uint32_t arow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ay0 = __builtin_amdgcn_readfirstlane(16 * wv) + 4 * i - o;     // row of q = 0; rows ay0..ay0+3 go together
        arow[i] = (ay0 >= 0 && ay0 < aw) ? (uint32_t)(4 * i * aw) * 4u : 0x40000000u;
    }
    const uint32_t act_bytes = (uint32_t)(aw * aw) * 4u;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    // `valid` false (no such grid): the descriptors get a zero range, every lane reads zeros and no
    // memory is touched -- the loads stay unconditional, so a register set is never live across a
    // phase it is not needed in (a conditional load would keep its old contents alive)
    auto issue = [&](int64_t g, Quads &r, bool valid) __attribute__((always_inline)) {
        if (!valid) g = 0;
        if (PACKED_IN) {
            // row 16w + 4i + q, dword k / 8 of the packed state: r.v[0] holds the four (i = x, y, z, w); r.v[1..3] stay unused
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char *>(reinterpret_cast<const char *>(u_in) + g * 512), 0, valid ? 512 : 0, 0x00020000);
            const uint32_t soff = (uint32_t)(16 * wv + q) * 8u + (uint32_t)(k >> 3) * 4u;
            r.v[0].x = __builtin_amdgcn_raw_buffer_load_b32(rs, soff, 0, 0);
            r.v[0].y = __builtin_amdgcn_raw_buffer_load_b32(rs, soff + 32u, 0, 0);
            r.v[0].z = __builtin_amdgcn_raw_buffer_load_b32(rs, soff + 64u, 0, 0);
            r.v[0].w = __builtin_amdgcn_raw_buffer_load_b32(rs, soff + 96u, 0, 0);
        } else {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char *>(reinterpret_cast<const char *>(u_in) + g * 16384), 0, valid ? 16384 : 0, 0x00020000);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 1024u * i, 0, NT_LD ? 2 : 0);   // aux 2 = nt
                r.v[i] = make_uint4(t.x, t.y, t.z, t.w);
            }
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

    const int64_t first = blockIdx.x, stride = gridDim.x;
    Quads r0, r1;
    issue(first, r0, first < n);                      // the first grid travels while the tables are laid down
    rnd_block_init(s.net, ws, tid);
    __syncthreads();
    ActionSeen seen;
    RND_STAMP_DECL();
    int64_t pending = -1;                             // grid whose features wait in s.feat for the linear layers

    // One grid: `cur` holds its loads, `nxt` receives the next grid's (issued as soon as cur is consumed).
    auto step = [&](int64_t g, Quads &cur, Quads &nxt) __attribute__((always_inline)) {
        RND_STAMP(0);
        // the NEXT grid's loads go out first, into the register set the previous grid vacated a whole
        // iteration ago: this workgroup then always has a grid in flight (issued after the conversion
        // below, 13 % of every iteration had none, and the chip's load stream thinned out with it)
        issue(g + stride, nxt, g + stride < n);
        // ---- A: nibbles of this lane's four rows, toggled; row words by OR over the 16 lanes
        uint32_t x[4];
        ActionSeen seen_g;                            // ENV: the predicates of THIS grid (folded into two scalars below,
#pragma unroll                                        //      no per-lane accumulator lives across the grid loop)
        for (int i = 0; i < 4; ++i) {
            uint32_t nib = PACKED_IN ? 0u : nibble_fast(cur.v[i]);
            if (HAS_ACT) {
                if ((rows_in_win >> i) & 1u) { if (ENV) seen_g.add(cur.a[i]); else seen.add(cur.a[i]); }
                nib ^= nibble_fast(cur.a[i]);         // toggle, env.py:182
            }
            x[i] = nib << (4 * (k & 7));
        }
        bool w_notone = true;                         // ENV: some action element of this wavefront's 16 rows differs from one
        if (ENV && HAS_ACT && ea.publish) {
            w_notone = __ballot(seen_g.not_all_ones<F32Cells>()) != 0ull;
            any_acc |= __ballot(seen_g.any_nonzero<F32Cells>()) != 0ull;
            notone_acc |= w_notone;
            if (lane == 0 && wv != 0) s.notone[wv] = w_notone ? 1u : 0u;
        }
        RND_STAMP(1);
        // the four rows' DPP chains are independent: interleaved, they fill each other's wait states
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] |= __builtin_amdgcn_update_dpp(0u, x[i], 0x111, 0xF, 0xF, true);   // row_shr:1
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] |= __builtin_amdgcn_update_dpp(0u, x[i], 0x112, 0xF, 0xF, true);   // row_shr:2
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] |= __builtin_amdgcn_update_dpp(0u, x[i], 0x114, 0xF, 0xF, true);   // row_shr:4
        if ((k & 7) == 7) {
            if (PACKED_IN) {                          // the toggles' row dwords on the state's
                s.rows_in[16 * wv + 0 + q][k >> 3] = x[0] ^ cur.v[0].x;
                s.rows_in[16 * wv + 4 + q][k >> 3] = x[1] ^ cur.v[0].y;
                s.rows_in[16 * wv + 8 + q][k >> 3] = x[2] ^ cur.v[0].z;
                s.rows_in[16 * wv + 12 + q][k >> 3] = x[3] ^ cur.v[0].w;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) s.rows_in[16 * wv + 4 * i + q][k >> 3] = x[i];
            }
        }
        RND_STAMP(2);
        __syncthreads();
        RND_STAMP(3);

        // ---- B: one generation, lane = row (wave 0; rows above / below by DPP wave rotates = y seam)
        //      E2 (wave 1, at the same time): linear layers + bonus of the PREVIOUS grid of this workgroup
        if (wv == 0) {
            const uint64_t c = ((uint64_t)s.rows_in[lane][1] << 32) | s.rows_in[lane][0];
            const uint64_t wst = rotl64(c, 1), est = rotr64(c, 1);          // torus seam in x
            uint64_t h0, h1, p0, p1;
            row_sums(wst, c, est, h0, h1, p0, p1);
            const uint64_t a0 = dpp64_ror1(h0), a1 = dpp64_ror1(h1);        // row above (lane - 1, wraps)
            const uint64_t b0 = dpp64_rol1(h0), b1 = dpp64_rol1(h1);        // row below
            // (the 18 rule bits expand to 18 SGPR-pair select masks: formed here, per grid, from an opaque copy of the
            // rule -- hoisted out of the grid loop they occupy 36 SGPRs through every phase: the one-launch instantiation
            // then needs 106 SGPRs and moves ~40 scalars per grid through VGPR lanes)
            uint32_t rule_g = rule;
            asm volatile("" : "+s"(rule_g));
            uint64_t nx = apply_rule(neighbour_count(a0, a1, b0, b1, p0, p1), c, rule_g);
            if (ENV && HAS_ACT) {
                // master reset (env.py:208-216): every action element of the WHOLE batch is one
                bool reset = reset_all;
                // wavefront 0 knows its own 16 action rows: not all ones there (always, under an agent) -> no reset, and
                // no LDS round trip on the generation's critical path; else the other wavefronts' verdicts, then the batch
                if (ea.publish && __builtin_expect(!w_notone, 0)) {
                    const bool own_notone = (s.notone[1] | s.notone[2] | s.notone[3]) != 0u;
                    if (!own_notone) reset = batch_action_all_ones<F32Cells>(act, n, aw, status, slot);
                }
                if (reset) nx = 0ull;
            }
            s.net.rows[1 + lane] = nx;
        } else if (wv == 1 && pending >= 0) {
            const float b = rnd_linear_bonus(s.feat, s.net.lin, lane, pending, nullptr, nullptr);
            if (lane == 0) bonus[pending] = b;
            if (ENV && lane == 0) {
                const float r = ea.reward_scale * b;              // reward = 0 + scale * bonus (mcl.py:403-405)
                if (ea.reward) ea.reward[pending] = r;
                s.racc[0] += (double)r; s.racc[1] = fma((double)r, (double)r, s.racc[1]);
            }
        }
        RND_STAMP(4);
        __syncthreads();
        RND_STAMP(5);

        // ---- D: conv1 (table lookup) + relu + 4x4 max-pool; thread = pooling window.  BEFORE the stores of C since the step
        // reads its state packed (round 5): with the memory time gone the launch is bound by LDS + VALU work per CU (three
        // workgroups per CU run as fast as four), and the LDS round trips of the lookups overlap better with the stores
        // issued behind them than the other way round: 30.8 -> 30.1 us, same box.  (The random net's table through the
        // vector L1 here: 31.7 / 33.7 us for 8 / 16 of the lookups -- `profiles/r05_rnd_d_first_ab.txt`.)
        RND_STAMP(6);
        rnd_block_conv1(s.net, tid, ws);
        RND_STAMP(7);
        // ---- C: fp32 store of the new state (the lane's own four quads): the lane's dword of the row
        // word, its nibble, bits -> bytes (one multiply), bytes -> 0.0f / 1.0f (hipcc selects v_cvt_f32_ubyteN)
        {
            const uint32_t *rows32 = reinterpret_cast<const uint32_t *>(&s.net.rows[1]);
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
                __builtin_amdgcn_raw_buffer_store_b128(o4, ro, voff + 1024u * i, 0, NT_ST ? 2 : 0);      // aux 2 = nt
            }
            // BITS: the observation once more, 1 bit per cell (512 B per grid), for a reward net stacked on this RND2D: the
            // row words as phase B left them, by a wavefront that is not on the generation's critical path, through a
            // scalar descriptor (no address VGPRs: this kernel sits at its 128-VGPR budget).  Same box, round 5: 37.0 us
            // (the step without a copy: 36.5; the 2 MiB more are 0.4 of that); stored by wavefront 0 in phase B: 38.5; here
            // with a 64-bit address per lane: 39.2.  A template parameter, not a test of the pointer: the run-time test
            // alone cost the instantiation WITHOUT a copy 1.9 us (36.40-36.49 against 38.28-38.37 us)
            if (BITS && wv == 3) {
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
                    reinterpret_cast<char *>(ea.bits_out) + g * 512, 0, 512, 0x00020000);
                u32x2 w2;
                w2.x = rows32[2 * lane];
                w2.y = rows32[2 * lane + 1];
                __builtin_amdgcn_raw_buffer_store_b64(w2, rb, 8u * lane, 0, 0);
            }
        }

        __syncthreads();
        RND_STAMP(8);

        // ---- E1: conv2 + pool on all threads -> s.feat.  No barrier behind it: the features are read
        // by wave 1 during the NEXT grid's phase B, i.e. behind that grid's first barrier; the planes are
        // rewritten by D two barriers from here, rows_in by A (last read in B, two barriers back).
        rnd_block_conv2(s.net, s.feat, tid);
        pending = g;
        RND_STAMP(9);
        RND_STAMP_NEXT();
    };

    for (int64_t g = first; g < n;) {                 // unrolled by two: the register sets swap roles
        step(g, r0, r1);
        g += stride;
        if (g >= n) break;
        step(g, r1, r0);
        g += stride;
    }
    // the bonus of the all-dead grid (used by the epilogue for every instance after a master reset)
    // was evaluated with the tables: rnd2d_prepare_tables
    if (blockIdx.x == 0 && tid == 0) bonus[n] = ws[kWsDeadBonus];
    if (pending >= 0) {                               // the last grid's linear layers
        __syncthreads();
        if (wv == 1) {
            const float b = rnd_linear_bonus(s.feat, s.net.lin, lane, pending, nullptr, nullptr);
            if (lane == 0) bonus[pending] = b;
            if (ENV && lane == 0) {
                const float r = ea.reward_scale * b;
                if (ea.reward) ea.reward[pending] = r;
                s.racc[0] += (double)r; s.racc[1] = fma((double)r, (double)r, s.racc[1]);
            }
        }
    }
    if (HAS_ACT && !ENV) publish_flags<F32Cells>(seen, status, slot);
    if (ENV) {
        if (HAS_ACT && ea.publish && lane == 0) {            // every writer stores the same value: benign
            if (any_acc) status[CARLE_ST_ANY_NONZERO(slot)] = 1;
            if (notone_acc) status[CARLE_ST_NOT_ALL_ONES(slot)] = 1;
        }
        if (wv == 1 && lane == 0 && ea.merge_moments && ea.partials)
            reinterpret_cast<double2 *>(ea.partials)[blockIdx.x] = make_double2(s.racc[0], s.racc[1]);
        if (blockIdx.x == 0 && wv == 2) {
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

void rnd2d_prepare_tables(const float *pp, const float *rp, float *ws, hipStream_t s);

// workgroups of the fused launch (= reward-moment partial pairs of the one-launch step)
int64_t rnd64_blocks(int64_t n) {
    int64_t blocks = n <= 1024 ? n : (n + 1) / 2;
    if (blocks < 1024 && n > 1024) blocks = 1024;
    const int64_t v = knobs().rnd_blocks;              // experiment knob (tools/kbench_rnd.py), read once
    if (v > 0) blocks = v < n ? v : n;
    return blocks < 1 ? 1 : blocks;
}

// validated by the caller (carle_env_step_f32_rnd); bonus has n + 1 elements.  env != nullptr: the one-launch step.
// packed_in (with env and env->bits_out): u_in points at the bit-packed copy of the state (n x 64 row words), not at fp32 cells
int ca_step_rnd64(const float *u_in, const float *action, float *u_out, int64_t n, int aw, uint32_t rule,
                  int32_t *status, int slot, const float *pp, const float *rp, float *ws, float *bonus,
                  bool tables_valid, hipStream_t s, const RndEnvArgs *env, bool packed_in) {
    if (!tables_valid) rnd2d_prepare_tables(pp, rp, ws, s);
    // Two grids per workgroup (the second one's loads travel during the first one's network phases); batches up to
    // 1024 grids one workgroup each.  Four per CU are resident, the others start as those retire -- out of phase, so
    // the memory system is not idle while every workgroup computes its last grid.  Kernel alone (tools/kbench_rnd.py,
    // profiles/r02_rnd_blocks.txt): N=4096 1024 / 2048 / 4096 workgroups 37.3-38.7 / 36.9 / 40.0 us, N=16384 1024 /
    // 8192 / 16384: 138 / 129.5 / 139 us, N=65536 1024 / 8192 / 32768 / 65536: 585 / 552 / 510 / 535 us (0.69 -> 0.79
    // of 8 TB/s).  (Before the table set-up stopped waiting for each of its loads, fewer, longer workgroups won.)
    int64_t blocks = rnd64_blocks(n);
    // the one-launch step (ENV instantiation: 106 SGPRs, a longer tail) is 1 % faster with the 1024 resident workgroups
    // walking all grids up to N = 8192 (same process, tools/kbench_rnd_sched.py: N=4096 36.92 -> 36.56 us, N=6144 51.27 ->
    // 50.74, N=8192 65.6 either way); the plain kernel keeps two grids per workgroup (N=6144: 49.9 against 50.7)
    if (env && n > 1024 && n <= 8192 && knobs().rnd_blocks <= 0) blocks = 1024;
    // store policy by the size of one state array (ca_step_cells.hip: stream_stores): write-back up to 256 MiB
    const int64_t lim = knobs().stream_store_min_bytes >= 0 ? knobs().stream_store_min_bytes - 1 : 256ll << 20;
    const bool nt_st = n * 16384 > lim;
    const int64_t ld_lim = knobs().rnd_cached_load_max_bytes >= 0 ? knobs().rnd_cached_load_max_bytes : 128ll << 20;
    const bool cached_ld = !nt_st && n * 16384 <= ld_lim;
    const dim3 grid((unsigned)blocks), block(256);
    const RndEnvArgs ea = env ? *env : RndEnvArgs();
    const int awk = action ? aw : 64;
    const bool bits = env && ea.bits_out != nullptr;
    if (packed_in) {
        if (!bits) return fail_arg(CARLE_E_NULL, "fused CA + RND2D step: a packed input state needs the one-launch step with bits_out");
        if (action && nt_st)
            hipLaunchKernelGGL((ca_step_rnd64_kernel<true, true, true, true, true, true>), grid, block, 0, s, u_in, action, u_out, n,
                               awk, rule, status, slot, ws, bonus, ea);
        else if (action)
            hipLaunchKernelGGL((ca_step_rnd64_kernel<true, false, true, true, true, true>), grid, block, 0, s, u_in, action, u_out, n,
                               awk, rule, status, slot, ws, bonus, ea);
        else if (nt_st)
            hipLaunchKernelGGL((ca_step_rnd64_kernel<false, true, true, true, true, true>), grid, block, 0, s, u_in, action, u_out, n,
                               awk, rule, status, slot, ws, bonus, ea);
        else
            hipLaunchKernelGGL((ca_step_rnd64_kernel<false, false, true, true, true, true>), grid, block, 0, s, u_in, action, u_out, n,
                               awk, rule, status, slot, ws, bonus, ea);
        return check_launch("carle_env_step_f32_rnd_packed(fused CA + RND2D)");
    }
#define CARLE_RND_LAUNCH(ACT, NTS, NTL, ENVF) \
    do { \
        if (ENVF && bits) hipLaunchKernelGGL((ca_step_rnd64_kernel<ACT, NTS, NTL, ENVF, ENVF>), grid, block, 0, s, u_in, action, \
                                             u_out, n, awk, rule, status, slot, ws, bonus, ea); \
        else hipLaunchKernelGGL((ca_step_rnd64_kernel<ACT, NTS, NTL, ENVF>), grid, block, 0, s, u_in, action, u_out, n, awk, \
                                rule, status, slot, ws, bonus, ea); \
    } while (0)
#define CARLE_RND_PICK(ENVF) \
    do { \
        if (action && cached_ld) CARLE_RND_LAUNCH(true, false, false, ENVF); \
        else if (action && nt_st) CARLE_RND_LAUNCH(true, true, true, ENVF); \
        else if (action) CARLE_RND_LAUNCH(true, false, true, ENVF); \
        else if (nt_st) CARLE_RND_LAUNCH(false, true, true, ENVF); \
        else CARLE_RND_LAUNCH(false, false, true, ENVF); \
    } while (0)
    if (env) CARLE_RND_PICK(true);
    else CARLE_RND_PICK(false);
#undef CARLE_RND_PICK
#undef CARLE_RND_LAUNCH
    return check_launch("carle_env_step_f32_rnd(fused CA + RND2D)");
}

}  // namespace carle

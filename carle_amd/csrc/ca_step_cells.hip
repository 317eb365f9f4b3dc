// ca_step_cells.hip -- one CA generation on cell-per-element (N,1,H,W) grids: fp32 (the
// reference's own format) and uint8 (CARLE_FMT_U8: the same layout at 1 B per cell).
//
// Replaces CARLE.apply_action + the update branch of CARLE.step (carle/env.py:150-182,219-229):
// ~19 ATen launches and ~15 full-grid passes become ONE pass: read state (+ action), write state.
//
// Fast path, 64x64 grids: one wavefront per grid.  A wave reads the grid with 16 fully
// coalesced 1-KiB dwordx4 loads; lane L = (q = L/16, k = L%16) receives, from load i, the four
// cells (row 4i+q, cols 4k..4k+3) and keeps them as nibble i of ONE 64-bit register: a
// 16-row x 4-column "patch" per lane, 4096 cells per wave, no LDS, no transposition.
//   - west/east neighbours: 1-bit shifts inside each nibble + the edge column of lane k-1 / k+1
//     (rotation inside the 16-lane row group = the torus seam in x)
//   - north/south neighbours: the same nibble of lane L-16 / L+16; the lanes that wrap
//     (q==0 going up, q==3 going down) rotate the word by one nibble = the torus seam in y
//   - 3x3 count by bit-sliced adders, rule as a mux tree with the 18 rule bits in SGPRs
//     (ca_bits.hpp); no MFMA: the stencil is not a contraction.
// Algorithmic HBM traffic: 4 B read + 4 B written per cell (+ 4 B per action cell).
//
// uint8 grids use the SAME lane <-> cell mapping with dword (4 cells) instead of dwordx4 accesses:
// every kernel here is a template over the cell traits of the state (F32Cells / U8Cells) and of
// the action (fp32 as the reference's agents produce it, or uint8); 1 B + 1 B per cell update.
//
// Generic path (any even square size): one thread per cell, neighbours through L1/L2.
#include <stdlib.h>
#include "ca_cells.hpp"
#include "rnd_device.hpp"      // row_nibbles_to_dword
#include "philox.hpp"

namespace carle {

// ---------------------------------------------------------------------------------------------
// 64x64 fast path: one wavefront per grid
// ---------------------------------------------------------------------------------------------
// parameters of the fused device-side random agent (RAND_ACT instantiation): toggles are drawn
// in-kernel with the counter layout of philox.hpp instead of being read from an action tensor
struct RandAct {
    float p = 0.0f;
    uint64_t seed = 0, offset = 0;
    float *action_out = nullptr;   // optional: also materialise the (N,1,AW,AW) action
};

#ifndef CARLE_ACT_EARLY
#define CARLE_ACT_EARLY 0
#endif
static constexpr int kActEarly = CARLE_ACT_EARLY;     // action loads requested before the state is packed

// Store policy of the step kernels, chosen per launch by the size of one state array: up to the 256 MiB of the Infinity
// Cache the output of step t is (partly) still cached when step t+1 reads it back -- default write-back stores;
// beyond, streaming stores.  Loads always carry the streaming hint (every byte is read once per step).  Same-box A/B,
// 64x64 fp32 kernel alone, with action (tools/kbench_64.py, profiles/r02_store_policy.txt), streaming -> write-back:
// N=4096 (64 MiB) 32.5 -> 31.7 us, N=8192 67.3 -> 61.0, N=16384 (256 MiB) 133.6 -> 119.8 (0.75 -> 0.84 of 8 TB/s),
// N=65536 (1 GiB) 537.6 -> 557.5 (worse: stays streaming); uint8 N=65536 (256 MiB) 129.6 -> 118.6.
// uint8 state keeps streaming stores (write-back measured within +-4 % either way: wide 256x256 N=1024 26.4 -> 27.6 us,
// 64x64 N=8192 17.9 -> 18.5, N=65536 129.6 -> 118.6); fp32 wide kernels: 256x256 N=256 28.0 -> 25.7 us, 512x512 N=256
// (256 MiB) 93.4 -> 79.1 us; fused CA + RND2D N=16384 129.5 -> 122.2 us.
static bool stream_stores(int64_t state_bytes, size_t cell_bytes) {
    const int64_t k = knobs().stream_store_min_bytes;       // experiment knob, read once (common.hpp)
    const int64_t lim = k >= 0 ? k - 1 : 256ll << 20;
    return cell_bytes != 4 || state_bytes > lim;
}

// ENV instantiation (CARLE_STEP_DEFERRED, carle_hip.h): the WHOLE CARLE.step in this one launch -- no epilogue kernel
// behind it (a dependent dispatch costs ~2 us against a 32 us kernel at N=4096).  What carle_ca_finish does is split up:
//   master reset    decided per wavefront: its own action not all ones -> no reset (always, for an agent's toggles);
//                   else batch_action_all_ones() (ca_cells.hpp).  With pre-stored flags (publish == 0) every wave reads them.
//   reward := 0     one store per wavefront
//   reward moments  (n, 0, 0) merged by wavefront 0 (the reward of a bare step is all zero whatever the action was)
//   counters        depend on the flags of THIS launch: left pending (CARLE_ST_PENDING), applied by the next launch on the
//                   block -- here by wavefront 0, which settles the PREVIOUS step -- or by carle_env_settle
struct EnvArgs {
    float *reward = nullptr;       // N floats, zeroed (nullable)
    int publish = 1;               // 0: the caller stored the action flags of `slot` beforehand
    int merge_moments = 0;
    const uint64_t *bits_in = nullptr;   // PACKED_IN (64x64): the bit-packed copy a previous launch left of u_in -- read INSTEAD of it
    uint64_t *bits_out = nullptr;  // 64 / 256 / 512 columns: also a bit-packed copy of the new state (H rows of W/64 words per
                                   // grid, carle_pack_f32_b1's layout) -- what the reward-net kernels read 30x faster than
                                   // the fp32 cells (nullable)
};

// The 64x64 step kernel holds a grid as 64 lanes x 16 nibbles: lane (q = lane / 16, k = lane % 16) has, in nibble i, columns
// 4k..4k+3 of row 4i + q.  Transposing the 16 x 16 nibbles of every 16-lane DPP row gives lane (q, i) the whole row 4i + q as
// one 64-bit word (nibble k = columns 4k..4k+3): carle_pack_f32_b1's layout.  Four butterfly stages, 16 DPP moves + 14
// permutes / selects: stage s exchanges, between lanes k and k ^ s, the nibbles whose index differs from the lane in bit s.
__device__ __forceinline__ uint64_t transpose_nibbles16(uint64_t x, int lane) {
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    {   // s = 8 (a dword): the lane with bit 3 clear keeps its low dword and takes the partner's low dword as its high one
        const bool b = (lane & 8) != 0;
        const uint32_t send = b ? lo : hi;
        const uint32_t got = __builtin_amdgcn_update_dpp(0u, send, 0x128, 0xF, 0xF, false);      // row_ror:8 = lane ^ 8
        lo = b ? got : lo;
        hi = b ? hi : got;
    }
    {   // s = 4 (16 bits inside every dword): partner by row_shl:4 for banks 0 / 2, row_shr:4 for banks 1 / 3
        const bool b = (lane & 4) != 0;
        const uint32_t sel = b ? 0x03020706u : 0x05040100u;     // v_perm_b32(partner, own): selector 4-7 = partner's bytes, 0-3 = own
        uint32_t plo = __builtin_amdgcn_update_dpp(0u, lo, 0x104, 0xF, 0x5, false);              // row_shl:4 -> banks 0, 2
        plo = __builtin_amdgcn_update_dpp(plo, lo, 0x114, 0xF, 0xA, false);                        // row_shr:4 -> banks 1, 3
        uint32_t phi = __builtin_amdgcn_update_dpp(0u, hi, 0x104, 0xF, 0x5, false);
        phi = __builtin_amdgcn_update_dpp(phi, hi, 0x114, 0xF, 0xA, false);
        lo = __builtin_amdgcn_perm(plo, lo, sel);
        hi = __builtin_amdgcn_perm(phi, hi, sel);
    }
    {   // s = 2 (bytes inside every 16 bits)
        const bool b = (lane & 2) != 0;
        const uint32_t sel = b ? 0x03070105u : 0x06020400u;
        const uint32_t plo = __builtin_amdgcn_update_dpp(0u, lo, 0x4E, 0xF, 0xF, false);         // quad_perm [2,3,0,1]
        const uint32_t phi = __builtin_amdgcn_update_dpp(0u, hi, 0x4E, 0xF, 0xF, false);
        lo = __builtin_amdgcn_perm(plo, lo, sel);
        hi = __builtin_amdgcn_perm(phi, hi, sel);
    }
    {   // s = 1 (nibbles inside every byte): rotate the partner's dword by a nibble towards the slot it fills
        const bool b = (lane & 1) != 0;
        const uint32_t keep = b ? 0xF0F0F0F0u : 0x0F0F0F0Fu, rot = b ? 4u : 28u;
        const uint32_t plo = __builtin_amdgcn_update_dpp(0u, lo, 0xB1, 0xF, 0xF, false);         // quad_perm [1,0,3,2]
        const uint32_t phi = __builtin_amdgcn_update_dpp(0u, hi, 0xB1, 0xF, 0xF, false);
        lo = (lo & keep) | (__builtin_amdgcn_alignbit(plo, plo, rot) & ~keep);
        hi = (hi & keep) | (__builtin_amdgcn_alignbit(phi, phi, rot) & ~keep);
    }
    return ((uint64_t)hi << 32) | lo;
}

// BITS (ENV, fp32 cells; round 5): the launch also leaves the new state bit-packed (64 row words per grid) for a reward net
// that sits on this environment -- the 64x64 net kernels then read 512 B instead of 16 KiB per grid
// PACKED_IN (with BITS): the state comes in as the 64 row words the previous launch left (512 B per grid instead of 16 KiB of
// fp32 cells that carry the same 4 096 bits): the inverse of the transpose above.  The fp32 state is still written every
// step -- it is the observation the caller gets -- but never read back by the next one.
template <class C, class A, bool HAS_ACT, bool NT_LD = false, bool NT_ST = false, int BLOCK = 256,
          bool RAND_ACT = false, bool ENV = false, bool BITS = false, bool PACKED_IN = false>
__global__ __launch_bounds__(BLOCK) void ca_step_64_kernel(
    const typename C::T *__restrict__ u_in, const typename A::T *__restrict__ act,
    typename C::T *__restrict__ u_out, int64_t n, int aw, uint32_t rule, int32_t *__restrict__ status,
    int slot, RandAct ra = RandAct(), EnvArgs ea = EnvArgs()) {
    const int lane = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    if (g >= n) return;                              // wave-uniform
    static_assert(!(ENV && RAND_ACT), "the in-kernel random agent keeps the two-launch step");
    static_assert(!PACKED_IN || (ENV && BITS), "the packed input is the previous launch's packed output");
    const int q = lane >> 4, k = lane & 15;
    const typename C::T *gin = u_in + g * 4096 + 4 * lane;   // 1024 quads per grid, 64 per load

    typename C::Quad v[16];
    uint64_t c = 0;
    uint64_t w_in = 0;                               // PACKED_IN: row 4k + q of the state (transposed into `c` once the
    if (PACKED_IN) {                                 // action loads are out: ONE memory round trip, not two)
        w_in = __builtin_nontemporal_load(ea.bits_in + g * 64 + 4 * k + q);
        if (!(HAS_ACT && !RAND_ACT)) c = transpose_nibbles16(w_in, lane);
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = C::template load<NT_LD>(gin + i * 256);
        if (!(HAS_ACT && !RAND_ACT)) {
#pragma unroll
            for (int i = 0; i < 16; ++i) c |= (uint64_t)C::nibble(v[i]) << (4 * i);
        }
    }

    bool own_notone = true;                          // ENV: some action element of this grid differs from one
    if (HAS_ACT) {
        // window offset o = (64-aw)/2 is a multiple of 4 (host guarantees aw % 8 == 0), so a
        // lane's four columns are all inside or all outside the window
        const int o = (64 - aw) >> 1;
        const int ax = 4 * k - o;
        const bool col_in = ax >= 0 && ax < aw;
        const typename A::T *abase0 = act + g * (int64_t)aw * aw;      // the grid's action: always a valid address
        const typename A::T *abase = abase0 + ax;
        ActionSeen seen;
        uint64_t tog = 0;
        if (RAND_ACT) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ay = 4 * i + q - o;
                if (col_in && ay >= 0 && ay < aw) {
                    const uint64_t quad = (uint64_t)((g * aw + ay) * (int64_t)aw + ax) >> 2;
                    const uint32_t nib = bernoulli_nibble(quad, ra.seed, ra.offset, ra.p);
                    uint4 a;
                    a.x = (nib & 1u) ? kOneBits : 0u;
                    a.y = (nib & 2u) ? kOneBits : 0u;
                    a.z = (nib & 4u) ? kOneBits : 0u;
                    a.w = (nib & 8u) ? kOneBits : 0u;
                    seen.add(a);
                    tog |= (uint64_t)nib << (4 * i);
                    if (ra.action_out)
                        *reinterpret_cast<uint4 *>(ra.action_out + (g * aw + ay) * (int64_t)aw + ax) = a;
                }
            }
        } else {
            // The 16 action loads are UNCONDITIONAL (lanes / rows outside the window read the grid's first quad and
            // drop it): a load inside a per-row `if` is followed by its own s_waitcnt vmcnt(0) -- sixteen
            // dependent memory round trips per wave instead of one (round 2, seen in the disassembly).
            typename A::Quad a[16];
            if (PACKED_IN) {                          // no state quads: the sixteen action loads are the launch's only wide loads
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int ay = 4 * i + q - o;
                    const bool in = col_in && ay >= 0 && ay < aw;
                    a[i] = A::template load<NT_LD>(in ? abase + (int64_t)ay * aw : abase0);
                }
                c = transpose_nibbles16(w_in, lane);
            } else {
#pragma unroll
                for (int i = 0; i < kActEarly; ++i) {
                    const int ay = 4 * i + q - o;
                    const bool in = col_in && ay >= 0 && ay < aw;
                    a[i] = A::template load<NT_LD>(in ? abase + (int64_t)ay * aw : abase0);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) c |= (uint64_t)C::nibble(v[i]) << (4 * i);
                // the remaining loads reuse the state's registers: their addresses are made to depend on the packed
                // state (an opaque zero), or the compiler requests all 32 quads at once (137 VGPRs, 3 waves per SIMD)
                uint32_t z = 0;
                if (kActEarly < 16)
                    asm volatile("v_mov_b32 %0, 0" : "=v"(z) : "v"((uint32_t)c), "v"((uint32_t)(c >> 32)));
#pragma unroll
                for (int i = kActEarly; i < 16; ++i) {
                    const int ay = 4 * i + q - o;
                    const bool in = col_in && ay >= 0 && ay < aw;
                    a[i] = A::template load<NT_LD>((in ? abase + (int64_t)ay * aw : abase0) + z);
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ay = 4 * i + q - o;
                if (col_in && ay >= 0 && ay < aw) {
                    see(seen, a[i]);
                    tog |= (uint64_t)A::nibble(a[i]) << (4 * i);
                }
            }
        }
        c ^= tog;                                    // toggle, env.py:182
        if (!ENV) {
            publish_flags<A>(seen, status, slot);
        } else if (ea.publish) {
            own_notone = publish_flags<A>(seen, status, slot);
        }
    }

    uint64_t nxt = patch_generation64(c, lane, rule);

    if (ENV && HAS_ACT) {
        // master reset (env.py:208-216): every action element of the WHOLE batch is one
        bool reset;
        if (!ea.publish) {
            reset = __builtin_nontemporal_load(status + CARLE_ST_NOT_ALL_ONES(slot)) == 0;   // stored before the launch
        } else {
            reset = false;
            if (__builtin_expect(!own_notone, 0)) reset = batch_action_all_ones<A>(act, n, aw, status, slot);
        }
        if (reset) nxt = 0ull;
    }

    typename C::T *gout = u_out + g * 4096 + 4 * lane;
#pragma unroll
    for (int i = 0; i < 16; ++i) C::template store<NT_ST>(gout + i * 256, (uint32_t)(nxt >> (4 * i)) & 0xFu);
    if (ENV && BITS) ea.bits_out[g * 64 + 4 * k + q] = transpose_nibbles16(nxt, lane);      // lane (q, i = k): row 4i + q

    if (ENV) {
        if (ea.reward && lane == 0) ea.reward[g] = 0.0f;               // env.py:238
        if (g == 0) env_step_bookkeeping(status, slot, n, HAS_ACT, ea.merge_moments && ea.reward != nullptr);
    }
}

// ca_step_64_kernel<..., ENV, BITS, PACKED_IN> with an action, TWO grids per wavefront (g and g + ceil(n / 2)).  With one grid
// per wavefront and every wavefront of a 4 096-grid batch resident at once, the launch is ONE generation of waves in lockstep:
// everybody loads, then everybody computes, then everybody stores -- the memory system sees reads and writes one after the
// other.  Here the second grid's action is requested as soon as the first one's is packed into its toggle word, so its 16 KiB
// travel while the first grid's generation is computed and its 16 KiB are stored.  Pays exactly where the halved wavefront
// count still fills the chip evenly, two per SIMD (same box, N = 3328 / 3584 / 3840 / 4096: 20.1 / 19.9 / 22.7 / 22.0 us against
// 20.6 / 21.9 / 22.9 / 23.8); below (N = 3072: 19.5 against 17.3) and above (4352: 26.7 / 25.9, 8192: 44.5 / 43.0) the longer
// waves cost more in imbalance than the overlap gives.  A persistent loop of this shape (2 048 wavefronts, any N) gained 4 % at
// N = 4096 and lost 10 % at N = 65536: larger batches desynchronise by themselves.
template <class A, bool NT_ST>
__global__ __launch_bounds__(64) void ca_step_64_packed2_kernel(
    const typename A::T *__restrict__ act, float *__restrict__ u_out, int64_t n, int aw, uint32_t rule,
    int32_t *__restrict__ status, int slot, EnvArgs ea) {
    const int lane = threadIdx.x;
    const int64_t half = (n + 1) >> 1;
    const int64_t g0 = blockIdx.x, g1 = g0 + half;
    const bool two = g1 < n;                         // (wave-uniform)
    const int q = lane >> 4, k = lane & 15;
    const int o = (64 - aw) >> 1;
    const int ax = 4 * k - o;
    const bool col_in = ax >= 0 && ax < aw;

    auto issue = [&](int64_t g, typename A::Quad (&a)[16], uint64_t &w) __attribute__((always_inline)) {
        w = __builtin_nontemporal_load(ea.bits_in + g * 64 + 4 * k + q);
        const typename A::T *abase0 = act + g * (int64_t)aw * aw, *abase = abase0 + ax;
#pragma unroll
        for (int i = 0; i < 16; ++i) {               // unconditional loads (see ca_step_64_kernel)
            const int ay = 4 * i + q - o;
            const bool in = col_in && ay >= 0 && ay < aw;
            a[i] = A::template load<true>(in ? abase + (int64_t)ay * aw : abase0);
        }
    };
    auto toggled = [&](typename A::Quad (&a)[16], uint64_t w, bool &own_notone) __attribute__((always_inline)) -> uint64_t {
        uint64_t c = transpose_nibbles16(w, lane), tog = 0;
        ActionSeen seen;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ay = 4 * i + q - o;
            if (col_in && ay >= 0 && ay < aw) {
                see(seen, a[i]);
                tog |= (uint64_t)A::nibble(a[i]) << (4 * i);
            }
        }
        own_notone = true;
        if (ea.publish) own_notone = publish_flags<A>(seen, status, slot);
        return c ^ tog;                              // toggle, env.py:182
    };
    auto finish = [&](int64_t g, uint64_t c, bool own_notone) __attribute__((always_inline)) {
        uint64_t nxt = patch_generation64(c, lane, rule);
        bool reset;                                  // master reset (env.py:208-216)
        if (!ea.publish) {
            reset = __builtin_nontemporal_load(status + CARLE_ST_NOT_ALL_ONES(slot)) == 0;
        } else {
            reset = false;
            if (__builtin_expect(!own_notone, 0)) reset = batch_action_all_ones<A>(act, n, aw, status, slot);
        }
        if (reset) nxt = 0ull;
        float *gout = u_out + g * 4096 + 4 * lane;
#pragma unroll
        for (int i = 0; i < 16; ++i) F32Cells::template store<NT_ST>(gout + i * 256, (uint32_t)(nxt >> (4 * i)) & 0xFu);
        ea.bits_out[g * 64 + 4 * k + q] = transpose_nibbles16(nxt, lane);
        if (ea.reward && lane == 0) ea.reward[g] = 0.0f;               // env.py:238
    };

    typename A::Quad a[16];
    uint64_t w;
    bool nn0, nn1;
    issue(g0, a, w);
    const uint64_t c0 = toggled(a, w, nn0);
    if (two) issue(g1, a, w);                        // in flight during the first grid's generation and stores
    finish(g0, c0, nn0);
    if (two) {
        const uint64_t c1 = toggled(a, w, nn1);
        finish(g1, c1, nn1);
    }
    if (g0 == 0) env_step_bookkeeping(status, slot, n, 1, ea.merge_moments && ea.reward != nullptr);
}

// ---------------------------------------------------------------------------------------------
// small batches of 64x64 fp32 grids (N <= 16): the WHOLE CARLE.step in ONE launch
// ---------------------------------------------------------------------------------------------
// One workgroup, one wavefront per grid.  Every grid of the batch is inside the workgroup, so the
// whole-batch master reset (env.py:208-216) needs no second kernel: the waves meet at a barrier,
// combine their action predicates through LDS, and the epilogue of carle_ca_finish (reset, counters,
// zero reward, optional reward moments; abi.hip) runs in place.  Halves the launches of the
// configurations whose step is bound by the host (BASELINE configs[0]: N = 1).
template <bool HAS_ACT, bool RAND_ACT>
__global__ __launch_bounds__(1024) void ca_env_step_small64_kernel(
    const float *__restrict__ u_in, const float *__restrict__ act, float *__restrict__ u_out,
    float *__restrict__ reward, int64_t n, int aw, uint32_t rule, int32_t *__restrict__ st, int slot,
    int publish, int merge_moments, RandAct ra) {
    __shared__ uint32_t s_any[16], s_notone[16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t g = wv;                                  // blockDim.x = 64 * n
    const int q = lane >> 4, k = lane & 15;
    const float *gin = u_in + g * 4096 + 4 * lane;
    uint4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = ld16<false>(reinterpret_cast<const uint4 *>(gin + i * 256));
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) c |= (uint64_t)nibble_of(v[i]) << (4 * i);
    ActionSeen seen;
    if (HAS_ACT) {
        const int o = (64 - aw) >> 1;
        const int ax = 4 * k - o;
        const bool col_in = ax >= 0 && ax < aw;
        const float *abase0 = act + g * (int64_t)aw * aw;   // the grid's action: always a valid address
        const float *abase = abase0 + ax;
        uint64_t tog = 0;
        if (RAND_ACT) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ay = 4 * i + q - o;
                if (col_in && ay >= 0 && ay < aw) {
                    uint4 a;
                    const uint64_t quad = (uint64_t)((g * aw + ay) * (int64_t)aw + ax) >> 2;
                    const uint32_t nib = bernoulli_nibble(quad, ra.seed, ra.offset, ra.p);
                    a.x = (nib & 1u) ? kOneBits : 0u;
                    a.y = (nib & 2u) ? kOneBits : 0u;
                    a.z = (nib & 4u) ? kOneBits : 0u;
                    a.w = (nib & 8u) ? kOneBits : 0u;
                    if (ra.action_out)
                        *reinterpret_cast<uint4 *>(ra.action_out + (g * aw + ay) * (int64_t)aw + ax) = a;
                    seen.add(a);
                    tog |= (uint64_t)nibble_of(a) << (4 * i);
                }
            }
        } else {
            // unconditional loads behind the packed state (see ca_step_64_kernel): two memory round trips per wave
            // instead of seventeen, within the 128 registers a 1024-thread workgroup allows
            uint32_t z;
            asm volatile("v_mov_b32 %0, 0" : "=v"(z) : "v"((uint32_t)c), "v"((uint32_t)(c >> 32)));
            uint4 a[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ay = 4 * i + q - o;
                const bool in = col_in && ay >= 0 && ay < aw;
                a[i] = ld16<false>(reinterpret_cast<const uint4 *>((in ? abase + (int64_t)ay * aw : abase0) + z));
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ay = 4 * i + q - o;
                if (col_in && ay >= 0 && ay < aw) {
                    seen.add(a[i]);
                    tog |= (uint64_t)nibble_of(a[i]) << (4 * i);
                }
            }
        }
        c ^= tog;                                          // toggle, env.py:182
    }
    const uint64_t nxt = patch_generation64(c, lane, rule);

    // ---- whole-batch predicates: through LDS; flags a caller stored beforehand (publish == 0:
    // oversize / exact predicates) are read from the status block instead
    const bool w_any = HAS_ACT && __ballot(seen.any_nonzero<F32Cells>()) != 0ull;       // wave-wide: every lane votes
    const bool w_notone = HAS_ACT && __ballot(seen.not_all_ones<F32Cells>()) != 0ull;
    if (lane == 0) {
        s_any[wv] = w_any;
        s_notone[wv] = w_notone;
    }
    __syncthreads();
    const uint4 *st4 = reinterpret_cast<const uint4 *>(st);
    const uint4 w0 = st4[0], w1 = st4[1], w2 = st4[2], w3 = st4[3];
    bool any = false, notone = false;
    if (publish) {
        for (int i = 0; i < (int)n; ++i) { any |= s_any[i] != 0u; notone |= s_notone[i] != 0u; }
    } else {
        any = (slot ? w0.y : w0.x) != 0u;
        notone = (slot ? w0.w : w0.z) != 0u;
    }
    const bool reset = HAS_ACT && !notone;                 // env.py:208: every element of the batch is 1.0
    any = HAS_ACT && any;

    float *gout = u_out + g * 4096 + 4 * lane;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        F32Cells::store<false>(gout + i * 256, reset ? 0u : (uint32_t)(nxt >> (4 * i)) & 0xFu);
    if (reward && threadIdx.x < n) reward[threadIdx.x] = 0.0f;          // env.py:238
    if (threadIdx.x == 0) {
        if (merge_moments && reward) {                     // moments of an all-zero reward vector
            double *mom_out = moments_out_slot(st, reinterpret_cast<double *>(((uint64_t)w3.y << 32) | w3.x));
            double *mom_acc = reinterpret_cast<double *>(((uint64_t)w3.w << 32) | w3.z);
            if (mom_out && mom_acc) {
                const double nb = (double)n, na = mom_acc[0], ma = mom_acc[1], qa = mom_acc[2], tot = na + nb;
                mom_out[0] = nb; mom_out[1] = 0.0; mom_out[2] = 0.0;
                if (tot > 0.0) {
                    mom_acc[0] = tot;
                    mom_acc[1] = ma - ma * nb / tot;
                    mom_acc[2] = qa + ma * ma * na * nb / tot;
                }
            }
        }
        // the predicates travelled through LDS: leave them in the step's flag slot as the step kernel of the general
        // path does (consumers after the step -- PufferDetector's "no action this step" -- read them there)
        uint4 w0f = w0;
        if (publish && HAS_ACT) {
            if (slot) { w0f.y = any ? 1u : 0u; w0f.w = notone ? 1u : 0u; } else { w0f.x = any ? 1u : 0u; w0f.z = notone ? 1u : 0u; }
        }
        finish_status(st, w0f, w1, w2, slot, reset, any);
    }
}

// k consecutive steps of a small batch in ONE launch (carle_env_run_f32, N <= 16): the grids stay in registers between
// the steps, every step is complete inside the workgroup -- toggle, generation, whole-batch predicates through LDS (two
// alternating flag arrays: one barrier per step), master reset, counters, reward moments -- exactly as k launches of
// ca_env_step_small64_kernel would leave the state, the status block and the running moments (same arithmetic, same
// order), without a dispatch per step (3.2 us each; BASELINE configs[0] is N = 1, 256 steps).  u_out may be u_in.
template <bool HAS_ACT, bool RAND_ACT>
__global__ __launch_bounds__(1024) void ca_env_run_small64_kernel(
    const float *__restrict__ u_in, const float *__restrict__ act, float *__restrict__ u_out,
    float *__restrict__ reward, int64_t n, int aw, uint32_t rule, int32_t *__restrict__ st, int slot0,
    int merge_moments, RandAct ra, int steps) {
    __shared__ uint32_t s_any[2][16], s_notone[2][16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t g = wv;                                  // blockDim.x = 64 * n
    const int q = lane >> 4, k = lane & 15;
    uint64_t c = 0;
    {
        const float *gin = u_in + g * 4096 + 4 * lane;
        uint4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = ld16<false>(reinterpret_cast<const uint4 *>(gin + i * 256));
#pragma unroll
        for (int i = 0; i < 16; ++i) c |= (uint64_t)nibble_of(v[i]) << (4 * i);
    }
    const uint4 *st4 = reinterpret_cast<const uint4 *>(st);
    uint4 w0s = st4[0], w1 = st4[1], w2 = st4[2];
    const uint4 w3 = st4[3];
    settle_pending(w0s, w1, w2);                           // a deferred one-launch step before this call
    int64_t step_number = (int64_t)(((uint64_t)w1.w << 32) | w1.z);
    int64_t since = (int64_t)(((uint64_t)w2.y << 32) | w2.x);
    int64_t resets = (int64_t)(((uint64_t)w2.w << 32) | w2.z);
    double *mom_out = reinterpret_cast<double *>(((uint64_t)w3.y << 32) | w3.x);
    double *mom_acc = reinterpret_cast<double *>(((uint64_t)w3.w << 32) | w3.z);
    const bool moments = merge_moments && reward && mom_out && mom_acc;
    double na = 0.0, ma = 0.0, qa = 0.0;
    if (moments && threadIdx.x == 0) { na = mom_acc[0]; ma = mom_acc[1]; qa = mom_acc[2]; }
    const int o = (64 - aw) >> 1, ax = 4 * k - o;
    const bool col_in = HAS_ACT && ax >= 0 && ax < aw;
    bool reset = false, any = false, notone = false;
    for (int t = 0; t < steps; ++t) {
        ActionSeen seen;
        if (HAS_ACT) {
            uint64_t tog = 0;
            if (RAND_ACT) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int ay = 4 * i + q - o;
                    if (col_in && ay >= 0 && ay < aw) {
                        const uint64_t quad = (uint64_t)((g * aw + ay) * (int64_t)aw + ax) >> 2;
                        const uint32_t nib = bernoulli_nibble(quad, ra.seed, ra.offset + (uint64_t)t, ra.p);
                        uint4 a;
                        a.x = (nib & 1u) ? kOneBits : 0u;
                        a.y = (nib & 2u) ? kOneBits : 0u;
                        a.z = (nib & 4u) ? kOneBits : 0u;
                        a.w = (nib & 8u) ? kOneBits : 0u;
                        seen.add(a);
                        tog |= (uint64_t)nib << (4 * i);
                    }
                }
            } else {
                const float *abase0 = act + ((int64_t)t * n + g) * (int64_t)aw * aw;   // step t's action of this grid
                const float *abase = abase0 + ax;
                uint4 a[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {                                         // unconditional loads
                    const int ay = 4 * i + q - o;
                    const bool in = col_in && ay >= 0 && ay < aw;
                    a[i] = ld16<false>(reinterpret_cast<const uint4 *>(in ? abase + (int64_t)ay * aw : abase0));
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int ay = 4 * i + q - o;
                    if (col_in && ay >= 0 && ay < aw) {
                        seen.add(a[i]);
                        tog |= (uint64_t)nibble_of(a[i]) << (4 * i);
                    }
                }
            }
            c ^= tog;                                      // toggle, env.py:182
        }
        const uint64_t nxt = patch_generation64(c, lane, rule);
        const int fb = t & 1;
        const bool w_any = HAS_ACT && __ballot(seen.any_nonzero<F32Cells>()) != 0ull;
        const bool w_notone = HAS_ACT && __ballot(seen.not_all_ones<F32Cells>()) != 0ull;
        if (lane == 0) { s_any[fb][wv] = w_any; s_notone[fb][wv] = w_notone; }
        __syncthreads();
        any = false; notone = false;
        for (int i = 0; i < (int)n; ++i) { any |= s_any[fb][i] != 0u; notone |= s_notone[fb][i] != 0u; }
        reset = HAS_ACT && !notone;                        // env.py:208: every element of the batch is 1.0
        any = HAS_ACT && any;
        c = reset ? 0ull : nxt;
        if (threadIdx.x == 0) {                            // the arithmetic of finish_status / the moments merge, per step
            if (!any) since += 1;
            if (reset) { step_number = 0; since = 0; resets += 1; } else { step_number += 1; }
            if (moments) {
                const double nb = (double)n, tot = na + nb;
                if (tot > 0.0) {
                    const double ma_new = ma - ma * nb / tot, qa_new = qa + ma * ma * na * nb / tot;
                    na = tot; ma = ma_new; qa = qa_new;
                }
            }
        }
    }
    float *gout = u_out + g * 4096 + 4 * lane;
#pragma unroll
    for (int i = 0; i < 16; ++i) F32Cells::store<false>(gout + i * 256, (uint32_t)(c >> (4 * i)) & 0xFu);
    if (reward && threadIdx.x < n) reward[threadIdx.x] = 0.0f;          // env.py:238
    if (threadIdx.x == 0 && steps > 0) {
        if (moments) {
            double *row = moments_out_slot_steps(st, mom_out, (double)n, steps);       // one row per step under a trace
            row[0] = (double)n; row[1] = 0.0; row[2] = 0.0;
            mom_acc[0] = na; mom_acc[1] = ma; mom_acc[2] = qa;
        }
        // the status block as the last step's kernel + epilogue leave it: its flags in its slot, the other slot recycled
        // (the predicates travel through LDS, the flag words are only recycled: step t clears the slot of steps t - 1 and
        // t + 1, so after two or more steps both slots are clear)
        const int slot = (slot0 + steps - 1) & 1;
        uint4 o0 = st4[0], o1 = w1, o2;
        if (slot) { o0.x = 0u; o0.z = 0u; } else { o0.y = 0u; o0.w = 0u; }
        if (steps >= 2) { o0.x = 0u; o0.y = 0u; o0.z = 0u; o0.w = 0u; }
        if (HAS_ACT) {                                     // the last step's predicates, in its slot
            if (slot) { o0.y = any ? 1u : 0u; o0.w = notone ? 1u : 0u; } else { o0.x = any ? 1u : 0u; o0.z = notone ? 1u : 0u; }
        }
        o1.x = reset ? 1u : 0u;
        o1.z = (uint32_t)step_number; o1.w = (uint32_t)((uint64_t)step_number >> 32);
        o2.x = (uint32_t)since; o2.y = (uint32_t)((uint64_t)since >> 32);
        o2.z = (uint32_t)resets; o2.w = (uint32_t)((uint64_t)resets >> 32);
        uint4 *so = reinterpret_cast<uint4 *>(st);
        so[0] = o0; so[1] = o1; so[2] = o2;
        so[4] = make_uint4(0u, 0u, 0u, 0u);
    }
}

// k action-free steps of ANY number of 64x64 fp32 grids in one launch (carle_env_run_f32 without actions): one wave per
// grid, the grid stays in the wave's 64-bit patch registers for all k generations (no predicates, no master reset
// without an action: env.py:191,208), one load and one store per grid.  Block 0 advances the counters by k and merges
// the k all-zero reward batches into the running moments exactly as k epilogues would.  VALU-bound: ~620 instructions
// per grid and generation.  u_out may be u_in.
__global__ __launch_bounds__(64) void ca_run_noaction64_kernel(const float *__restrict__ u_in, float *__restrict__ u_out,
                                                              float *__restrict__ reward, int64_t n, uint32_t rule,
                                                              int32_t *__restrict__ st, int slot0, int merge_moments,
                                                              int steps, int nt_st) {
    const int lane = threadIdx.x;
    const int64_t g = blockIdx.x;
    const float *gin = u_in + g * 4096 + 4 * lane;
    uint4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = ld16<true>(reinterpret_cast<const uint4 *>(gin + i * 256));
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) c |= (uint64_t)nibble_of(v[i]) << (4 * i);
    for (int t = 0; t < steps; ++t) c = patch_generation64(c, lane, rule);
    float *gout = u_out + g * 4096 + 4 * lane;
    if (nt_st) {
#pragma unroll
        for (int i = 0; i < 16; ++i) F32Cells::store<true>(gout + i * 256, (uint32_t)(c >> (4 * i)) & 0xFu);
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) F32Cells::store<false>(gout + i * 256, (uint32_t)(c >> (4 * i)) & 0xFu);
    }
    if (reward && lane == 0) reward[g] = 0.0f;             // env.py:238
    if (g == 0 && lane == 0 && steps > 0) {
        const uint4 *st4 = reinterpret_cast<const uint4 *>(st);
        uint4 w0 = st4[0], w1 = st4[1], w2 = st4[2];
        const uint4 w3 = st4[3];
        settle_pending(w0, w1, w2);                         // a deferred one-launch step before this call
        if (merge_moments && reward) {
            double *mom_out = reinterpret_cast<double *>(((uint64_t)w3.y << 32) | w3.x);
            double *mom_acc = reinterpret_cast<double *>(((uint64_t)w3.w << 32) | w3.z);
            if (mom_out && mom_acc) {
                double na = mom_acc[0], ma = mom_acc[1], qa = mom_acc[2];
                const double nb = (double)n;
                for (int t = 0; t < steps; ++t) {          // k merges of (n, 0, 0), in the epilogue's arithmetic
                    const double tot = na + nb;
                    if (tot > 0.0) {
                        const double ma_new = ma - ma * nb / tot, qa_new = qa + ma * ma * na * nb / tot;
                        na = tot; ma = ma_new; qa = qa_new;
                    }
                }
                double *row = moments_out_slot_steps(st, mom_out, nb, steps);
                row[0] = nb; row[1] = 0.0; row[2] = 0.0;
                mom_acc[0] = na; mom_acc[1] = ma; mom_acc[2] = qa;
            }
        }
        int64_t step_number = (int64_t)(((uint64_t)w1.w << 32) | w1.z) + steps;
        int64_t since = (int64_t)(((uint64_t)w2.y << 32) | w2.x) + steps;
        const int slot = (slot0 + steps - 1) & 1;
        uint4 o0 = w0, o1 = w1, o2 = w2;
        if (slot) { o0.x = 0u; o0.z = 0u; } else { o0.y = 0u; o0.w = 0u; }     // flag words recycled as by k epilogues
        if (steps >= 2) { o0.x = 0u; o0.y = 0u; o0.z = 0u; o0.w = 0u; }
        o1.x = 0u;                                          // CARLE_ST_LAST_RESET
        o1.z = (uint32_t)step_number; o1.w = (uint32_t)((uint64_t)step_number >> 32);
        o2.x = (uint32_t)since; o2.y = (uint32_t)((uint64_t)since >> 32);
        uint4 *so = reinterpret_cast<uint4 *>(st);
        so[0] = o0; so[1] = o1; so[2] = o2;
        so[4] = make_uint4(0u, 0u, 0u, 0u);
    }
}

// ---------------------------------------------------------------------------------------------
// wide fast path: W = 256 * NS (NS = 1, 2), one wavefront per strip of R = 16 * RW rows
// ---------------------------------------------------------------------------------------------
// One load instruction covers 256 consecutive floats = (a 256-column block of) one row, so lane k
// owns the column quad 4k..4k+3 of every row of the strip (for NS = 2 also quad 256+4k..).  Row
// 16m+i of the strip is nibble i of word m: north/south neighbours are 4-bit shifts INSIDE the
// lane (carried across words), west/east neighbours are the edge columns of lanes k-1 / k+1
// (rotation over the whole wave, and over the column blocks, = the x torus seam).  The rows just
// above and below the strip (y torus seam included) are loaded as two extra "halo" rows.
// Workgroup ids are remapped so that vertically adjacent strips run on the same XCD (blocks b and
// b+8 share an L2): the halo rows are then L2 hits instead of second HBM reads.
static constexpr int kWideRW = 1;     // words (x16 rows) per strip; tuned in tools/kbench

template <int NS, int RW>
struct WideGeom {
    static constexpr int kW = 256 * NS;
    static constexpr int kR = 16 * RW;
};

__device__ __forceinline__ void horiz_neighbours(uint64_t c, uint64_t cw, uint64_t ce,
                                                 uint64_t &wst, uint64_t &est) {
    wst = ((c << 1) & 0xEEEEEEEEEEEEEEEEull) | ((cw >> 3) & 0x1111111111111111ull);
    est = ((c >> 1) & 0x7777777777777777ull) | ((ce << 3) & 0x8888888888888888ull);
}

// CARLE_WIDE_WAVES (WV) consecutive strips of a grid form one workgroup: a strip's first and last row reach the
// neighbouring waves through LDS (one nibble per lane, after the toggle), so only the workgroup's outer two halo
// rows are read from memory -- 2 extra row loads per 16 * WV rows instead of per 16 (12.5 % -> 3 % more load
// instructions, and the halo rows no longer miss the L2: 4 % less HBM traffic).  Same-box A/B, fp32, kernel alone
// (tools/kbench_wide.py, profiles/r02_kbench_wide.txt): WV = 1 / 2 / 4 / 8 -> 256x256 N=4096 372-394 / 373 /
// 345-359 / 340-352 us, 512x512 N=1024 393 / 363 / 345-348 / 353 us; with the action window 4 is best.
#ifndef CARLE_WIDE_WAVES
#define CARLE_WIDE_WAVES 4
#endif
static constexpr int kWideWaves = CARLE_WIDE_WAVES;

// ENV instantiation (CARLE_STEP_DEFERRED): the whole CARLE.step in this one launch, as ca_step_64_kernel<..., ENV>.  A
// workgroup owns four strips of ONE grid and most strips never see an action element, so the master reset is decided from
// a sample every workgroup can afford: wavefront 0 loads the FIRST ROW of its grid's action window with its other first
// loads (256 bytes, a cache hit for all but the first workgroup of the grid); an element there that is not one rules the
// reset out (always, under an agent), otherwise batch_action_all_ones() verifies the batch.  The verdict reaches the
// other wavefronts through the LDS exchange (and barrier) the edge rows use anyway.
// PK (round 5, fp32 cells): 1 = the launch also leaves the new state bit-packed in ea.bits_out (H rows of W/64 words per grid),
// by the nibble transpose of ca_step_64_kernel -- the 16 lanes of a DPP row hold 16 rows x 16 column quads of one word column;
// 2 = and it reads its INPUT state from ea.bits_in, the copy the previous launch left (one 8-byte load per lane and 16-row
// block instead of sixteen quads; the two outer halo rows: one dword each).  With or without ENV.
template <class C, class A, bool HAS_ACT, int NS, int RW, bool RAND_ACT = false, bool NT_ST = true, bool ENV = false, int PK = 0>
__global__ __launch_bounds__(64 * kWideWaves) void ca_step_wide_kernel(
    const typename C::T *__restrict__ u_in, const typename A::T *__restrict__ act,
    typename C::T *__restrict__ u_out, int64_t n, int ah, int aw, uint32_t rule,
    int32_t *__restrict__ status, int slot, RandAct ra = RandAct(), EnvArgs ea = EnvArgs()) {
    static_assert(!(ENV && RAND_ACT), "the in-kernel random agent keeps the two-launch step");
    static_assert(!ENV || kWideWaves > 1, "the verdict travels with the edge rows");
    __shared__ uint32_t s_reset;
    const bool reset_all = ENV && HAS_ACT && !ea.publish && status[CARLE_ST_NOT_ALL_ONES(slot)] == 0;
    constexpr int W = WideGeom<NS, RW>::kW, H = W, R = WideGeom<NS, RW>::kR;
    constexpr int WV = kWideWaves;
    constexpr int kStrips = H / R, kGroups = kStrips / WV;
    static_assert(kStrips % WV == 0, "strips per grid must be a multiple of the waves per workgroup");
    const int k = threadIdx.x & 63;                  // lane = column quad
    const int wv = threadIdx.x >> 6;
    const bool top_global = WV == 1 || wv == 0, bot_global = WV == 1 || wv == WV - 1;   // halo row comes from memory
    __shared__ uint32_t edge[WV][2][NS][64];         // [wave][first / last row][segment][lane]: nibbles
    // XCD-aware bijective remap of the list position (cdna_hip_programming.md T1): positions b, b + 8,
    // ... share an XCD and map to consecutive strips (halo rows are L2 hits).  The loop runs once per
    // workgroup in the shipped launch (one workgroup per group of WV strips); a capped launch walks the list with
    // stride gridDim.x (then a multiple of 8).
    const int64_t nwg = n * kGroups;
    const int64_t qd = nwg / 8, rm = nwg % 8;
  for (int64_t b = blockIdx.x; b < nwg; b += gridDim.x) {
    const int64_t xcd = b % 8;
    const int64_t wid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + b / 8;
    const int64_t g = wid / kGroups;
    const int r0 = ((int)(wid % kGroups) * WV + wv) * R;
    const typename C::T *gin = u_in + g * (int64_t)H * W;
    const int rt = (r0 + H - 1) % H, rb = (r0 + R) % H;      // halo rows (torus in y)

    // Action window, 64 columns wide (the reference's default, env.py:25-26), read from memory: its quads are
    // requested BEFORE the state rows, compactly -- one load covers four window rows (16 lanes each) instead
    // of one row on 16 of the 64 lanes -- and meet their column lanes by three lane permutes further down.
    // (Loaded in the state's own mapping after the state had been packed, the window cost the strips that touch
    // it a second, dependent memory round trip: 417 against 392 us per step at 256x256, N=4096.)
    const int oy = HAS_ACT ? (H - ah) >> 1 : 0, ox = HAS_ACT ? (W - aw) >> 1 : 0;   // ox % 4 == 0 (host guarantees)
    // does this strip (with its halo rows) touch the window rows at all?  (wave-uniform)
    const bool touches = HAS_ACT && ((r0 + R > oy && r0 < oy + ah) || (top_global && rt >= oy && rt < oy + ah) ||
                                     (bot_global && rb >= oy && rb < oy + ah));
    const bool compact = touches && !RAND_ACT && aw == 64;
    const typename A::T *ga = (HAS_ACT && !RAND_ACT) ? act + g * (int64_t)ah * aw : nullptr;
    typename A::Quad aq[RW][4], aht[NS], ahb[NS];
    typename A::Quad achk = typename A::Quad{};      // ENV: a quad of the first window row (lanes beyond it repeat quad 0)
    if (ENV && HAS_ACT && ea.publish && wv == 0)
        achk = A::template load<false>(act + g * (int64_t)ah * aw + 4 * (k < (aw >> 2) ? k : 0));
    uint32_t arow_ok = 0;                            // bit 4m+j: this lane's row of load (m, j) is a window row
#pragma unroll
    for (int m = 0; m < RW; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) aq[m][j] = typename A::Quad{};
#pragma unroll
    for (int s = 0; s < NS; ++s) { aht[s] = typename A::Quad{}; ahb[s] = typename A::Quad{}; }
    if (HAS_ACT && !RAND_ACT && compact) {
#pragma unroll
        for (int m = 0; m < RW; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ay = r0 + 16 * m + 4 * j + (k >> 4) - oy;
                if (ay >= 0 && ay < ah) {
                    aq[m][j] = A::template load<true>(ga + (int64_t)ay * 64 + 4 * (k & 15));
                    arow_ok |= 1u << (4 * m + j);
                }
            }
        const int ayt = rt - oy, ayb = rb - oy;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int ax = 256 * s + 4 * k - ox;
            if (top_global && ax >= 0 && ax < 64 && ayt >= 0 && ayt < ah) aht[s] = A::template load<false>(ga + (int64_t)ayt * 64 + ax);
            if (bot_global && ax >= 0 && ax < 64 && ayb >= 0 && ayb < ah) ahb[s] = A::template load<false>(ga + (int64_t)ayb * 64 + ax);
        }
    }

    // The workgroup's outer halo rows (waves 0 and WV-1 only) are requested before the strip's own rows.  The
    // rows the other waves do not load start as an OPAQUE zero: with a literal zero the compiler folds the nibble
    // of the unloaded path, sinks the conversion into the `if` and waits for the load there -- a second, dependent
    // memory round trip for the edge waves (seen in the disassembly, round 2).
    uint32_t oz;
    asm volatile("v_mov_b32 %0, 0" : "=v"(oz));
    uint64_t c[NS][RW];
    uint32_t halo[NS];                               // bits 0-3: row above, bits 4-7: row below
    if (PK == 2) {
        // the state as row words: lane (j = k / 16, i = k % 16) loads word 4s + j of row r0 + 16m + i; the transpose leaves lane k
        // with the nibbles of column quad k (of segment s) for the sixteen rows.  Halo rows: the dword that holds this lane's quad.
        const uint64_t *bin = ea.bits_in + g * (int64_t)H * (W / 64);
        const uint32_t *bin32 = reinterpret_cast<const uint32_t *>(bin);
        uint32_t ht[NS], hb[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            ht[s] = oz; hb[s] = oz;
            if (top_global) ht[s] = bin32[(int64_t)rt * (W / 32) + 8 * s + (k >> 3)];
            if (bot_global) hb[s] = bin32[(int64_t)rb * (W / 32) + 8 * s + (k >> 3)];
        }
        uint64_t wd[NS][RW];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int m = 0; m < RW; ++m)
                wd[s][m] = __builtin_nontemporal_load(bin + (int64_t)(r0 + 16 * m + (k & 15)) * (W / 64) + 4 * s + (k >> 4));
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int m = 0; m < RW; ++m) c[s][m] = transpose_nibbles16(wd[s][m], k);
#pragma unroll
        for (int s = 0; s < NS; ++s)
            halo[s] = ((ht[s] >> (4 * (k & 7))) & 0xFu) | (((hb[s] >> (4 * (k & 7))) & 0xFu) << 4);
    } else {
    typename C::Quad vt[NS], vb[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        vt[s] = C::splat(oz);
        vb[s] = C::splat(oz);
        // halo rows are re-read by the neighbouring workgroup: default cache policy keeps them in L2
        if (top_global) vt[s] = C::template load<false>(gin + (int64_t)rt * W + 256 * s + 4 * k);
        if (bot_global) vb[s] = C::template load<false>(gin + (int64_t)rb * W + 256 * s + 4 * k);
    }

#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int m = 0; m < RW; ++m) {
            typename C::Quad v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i)
                v[i] = C::template load<true>(gin + (int64_t)(r0 + 16 * m + i) * W + 256 * s + 4 * k);
            uint64_t acc = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc |= (uint64_t)C::nibble(v[i]) << (4 * i);
            c[s][m] = acc;
        }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) halo[s] = C::nibble(vt[s]) | (C::nibble(vb[s]) << 4);
    }

    if (HAS_ACT) {
        if (compact) {
            ActionSeen seen;
#pragma unroll
            for (int m = 0; m < RW; ++m) {
                uint64_t part = 0;                   // this lane's rows 4j + k/16 of window column quad k%16
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if ((arow_ok >> (4 * m + j)) & 1u) see(seen, aq[m][j]);
                    part |= (uint64_t)A::nibble(aq[m][j]) << (4 * (4 * j + (k >> 4)));
                }
                part |= shfl64(part, k ^ 16);
                part |= shfl64(part, k ^ 32);        // all 16 rows of column quad k%16, on four lanes
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int kq = 64 * s + k - (ox >> 2);           // this lane's column quad inside the window
                    const uint64_t t = shfl64(part, kq & 15);
                    c[s][m] ^= (kq >= 0 && kq < 16) ? t : 0ull;
                }
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) halo[s] ^= A::nibble(aht[s]) | (A::nibble(ahb[s]) << 4);
            if (!ENV || ea.publish) publish_flags<A>(seen, status, slot);      // (ENV, publish == 0: the caller stored them)
        } else if (touches) {
            // in-kernel random agent: the nibble of action quad (ay, ax) is a pure function of its
            // index, so halo rows draw the same toggles as the strip that owns the row
            auto drawn = [&](int ay, int ax) -> uint32_t {
                const uint64_t quad = (uint64_t)((g * ah + ay) * (int64_t)aw + ax) >> 2;
                return bernoulli_nibble(quad, ra.seed, ra.offset, ra.p);
            };
            ActionSeen seen;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int ax = 256 * s + 4 * k - ox;
                const bool col_in = ax >= 0 && ax < aw;
#pragma unroll
                for (int m = 0; m < RW; ++m) {
                    uint64_t tog = 0;
                    if (RAND_ACT) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int ay = r0 + 16 * m + i - oy;
                            if (col_in && ay >= 0 && ay < ah) {
                                const uint32_t nib = drawn(ay, ax);
                                uint4 a;
                                a.x = (nib & 1u) ? kOneBits : 0u;
                                a.y = (nib & 2u) ? kOneBits : 0u;
                                a.z = (nib & 4u) ? kOneBits : 0u;
                                a.w = (nib & 8u) ? kOneBits : 0u;
                                seen.add(a);
                                tog |= (uint64_t)nib << (4 * i);
                                if (ra.action_out)
                                    *reinterpret_cast<uint4 *>(ra.action_out + (g * ah + ay) * (int64_t)aw + ax) = a;
                            }
                        }
                    } else {
                        // windows of other widths: a window row per load, unconditional (lanes / rows outside the
                        // window read the grid's first quad and drop it) -- no wait per load
                        typename A::Quad a[16];
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int ay = r0 + 16 * m + i - oy;
                            a[i] = A::template load<true>((col_in && ay >= 0 && ay < ah) ? ga + (int64_t)ay * aw + ax : ga);
                        }
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int ay = r0 + 16 * m + i - oy;
                            if (col_in && ay >= 0 && ay < ah) {
                                see(seen, a[i]);
                                tog |= (uint64_t)A::nibble(a[i]) << (4 * i);
                            }
                        }
                    }
                    c[s][m] ^= tog;
                }
                const int ayt = rt - oy, ayb = rb - oy;
                const bool tin = top_global && col_in && ayt >= 0 && ayt < ah, bin = bot_global && col_in && ayb >= 0 && ayb < ah;
                if (RAND_ACT) {
                    if (tin) halo[s] ^= drawn(ayt, ax);
                    if (bin) halo[s] ^= drawn(ayb, ax) << 4;
                } else {
                    const typename A::Quad at = A::template load<false>(tin ? ga + (int64_t)ayt * aw + ax : ga);
                    const typename A::Quad ab = A::template load<false>(bin ? ga + (int64_t)ayb * aw + ax : ga);
                    halo[s] ^= (tin ? A::nibble(at) : 0u) | ((bin ? A::nibble(ab) : 0u) << 4);
                }
            }
            if (!ENV || ea.publish) publish_flags<A>(seen, status, slot);      // (ENV, publish == 0: the caller stored them)
        } else if (status != nullptr && k == 0) {
            // a strip outside the window sees no action element; the flags describe the whole
            // action, which the in-window strips publish.  Nothing to do.
        }
    }

    if (ENV && HAS_ACT && wv == 0) {                 // master reset (env.py:208-216): see above
        bool reset = reset_all;
        if (ea.publish) {
            ActionSeen sample;
            see(sample, achk);
            const bool notone = __ballot(sample.not_all_ones<A>()) != 0ull;
            if (__builtin_expect(!notone, 0)) reset = batch_action_all_ones<A>(act, n, aw, status, slot);
        }
        if (k == 0) s_reset = reset ? 1u : 0u;
    }
    if (WV > 1) {                                    // the inner halo rows: the neighbouring waves' edge rows, toggled
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            edge[wv][0][s][k] = (uint32_t)c[s][0] & 0xFu;
            edge[wv][1][s][k] = (uint32_t)(c[s][RW - 1] >> 60);
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (!top_global) halo[s] |= edge[wv - 1][1][s][k];
            if (!bot_global) halo[s] |= edge[wv + 1][0][s][k] << 4;
        }
    }
    const bool wg_reset = ENV && HAS_ACT && s_reset != 0u;      // read once, with the edge rows

    // horizontal sums of every row (strip words and the two halo rows)
    uint64_t h0[NS][RW], h1[NS][RW], p0[NS][RW], p1[NS][RW];
    uint32_t hh0[NS], hh1[NS];
    const int lw = (k + 63) & 63, le = (k + 1) & 63;
#pragma unroll
    for (int m = 0; m < RW; ++m) {
        uint64_t prv[NS], nxt[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) { prv[s] = shfl64(c[s][m], lw); nxt[s] = shfl64(c[s][m], le); }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const uint64_t cw = (NS == 1 || k > 0) ? prv[s] : prv[(s + NS - 1) % NS];
            const uint64_t ce = (NS == 1 || k < 63) ? nxt[s] : nxt[(s + 1) % NS];
            uint64_t wst, est;
            horiz_neighbours(c[s][m], cw, ce, wst, est);
            row_sums(wst, c[s][m], est, h0[s][m], h1[s][m], p0[s][m], p1[s][m]);
        }
    }
    {
        uint32_t prv[NS], nxt[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) { prv[s] = __shfl(halo[s], lw, 64); nxt[s] = __shfl(halo[s], le, 64); }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const uint32_t cw = (NS == 1 || k > 0) ? prv[s] : prv[(s + NS - 1) % NS];
            const uint32_t ce = (NS == 1 || k < 63) ? nxt[s] : nxt[(s + 1) % NS];
            uint64_t wst, est, a, bq, pa, pb;
            horiz_neighbours((uint64_t)halo[s], (uint64_t)cw, (uint64_t)ce, wst, est);
            row_sums(wst, (uint64_t)halo[s], est, a, bq, pa, pb);
            hh0[s] = (uint32_t)a & 0xFFu;
            hh1[s] = (uint32_t)bq & 0xFFu;
        }
    }

    typename C::T *gout = u_out + g * (int64_t)H * W;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int m = 0; m < RW; ++m) {
            // triple sums of the row above / below every row of word m
            const uint64_t up0 = (h0[s][m] << 4) | (m == 0 ? (uint64_t)(hh0[s] & 0xFu) : (h0[s][m > 0 ? m - 1 : 0] >> 60));
            const uint64_t up1 = (h1[s][m] << 4) | (m == 0 ? (uint64_t)(hh1[s] & 0xFu) : (h1[s][m > 0 ? m - 1 : 0] >> 60));
            const uint64_t dn0 = (h0[s][m] >> 4) |
                ((m == RW - 1 ? (uint64_t)(hh0[s] >> 4) : (h0[s][m < RW - 1 ? m + 1 : 0] & 0xFull)) << 60);
            const uint64_t dn1 = (h1[s][m] >> 4) |
                ((m == RW - 1 ? (uint64_t)(hh1[s] >> 4) : (h1[s][m < RW - 1 ? m + 1 : 0] & 0xFull)) << 60);
            uint64_t nx = apply_rule(neighbour_count(up0, up1, dn0, dn1, p0[s][m], p1[s][m]),
                                     c[s][m], rule);
            if (ENV && HAS_ACT && wg_reset) nx = 0ull;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                C::template store<NT_ST>(gout + (int64_t)(r0 + 16 * m + i) * W + 256 * s + 4 * k,
                                        (uint32_t)(nx >> (4 * i)) & 0xFu);
            if (PK != 0) {                           // lane (j, i): word 4s + j of row r0 + 16m + i
                ea.bits_out[g * (int64_t)H * (W / 64) + (int64_t)(r0 + 16 * m + (k & 15)) * (W / 64) + 4 * s + (k >> 4)] =
                    transpose_nibbles16(nx, k);
            } else if (ENV && ea.bits_out) {         // kernel argument: the same for every wavefront
                // the 16 lanes of a DPP row hold the nibbles of 64 columns of one grid row: OR them into the row's two
                // dwords (lanes 7 and 15 of the row end up with them) -- 5 operations per row
                uint32_t *b32 = reinterpret_cast<uint32_t *>(ea.bits_out) + g * (int64_t)H * (W / 32);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const uint32_t dw = row_nibbles_to_dword((uint32_t)(nx >> (4 * i)) & 0xFu, k & 15);
                    if ((k & 7) == 7) b32[(int64_t)(r0 + 16 * m + i) * (W / 32) + 8 * s + (k >> 3)] = dw;
                }
            }
        }
    }
    if (ENV && ea.reward && wid % kGroups == 0 && threadIdx.x == 0) ea.reward[g] = 0.0f;      // env.py:238
    if (WV > 1 && b + gridDim.x < nwg) __syncthreads();      // capped launch only: `edge` / s_reset are rewritten
  }
    if (ENV && blockIdx.x == 0 && wv == 0)
        env_step_bookkeeping(status, slot, n, HAS_ACT, ea.merge_moments && ea.reward != nullptr);
}

// Launch size of the wavefront-per-strip kernels: one workgroup per group of kWideWaves strips.  (The kernels can walk the
// strip list persistently, and the bit-packed kernel gains 9 % that way, but here -- 62-90 VGPRs,
// 5-8 waves per SIMD, strips of unequal cost under the action window -- the one-shot launch measured
// 6-20 % faster: profiles/r02_kbench_wide.txt.  CARLE_WIDE_BLOCKS caps the launch for experiments.)
static unsigned persistent_blocks(int64_t work_items) {
    int64_t cap = work_items;
    if (const int64_t v = knobs().wide_blocks; v >= 8) cap = v / 8 * 8;
    return (unsigned)(work_items < cap ? work_items : cap);
}

template <class C, class A, int NS, int RW>
static int launch_wide(const typename C::T *u_in, const typename A::T *action, typename C::T *u_out,
                       int64_t n, int ah, int aw, uint32_t rule, int32_t *status, int slot,
                       hipStream_t s, const EnvArgs *env = nullptr, const EnvArgs *packed = nullptr) {
    const int64_t wgs = n * (WideGeom<NS, RW>::kW / WideGeom<NS, RW>::kR / kWideWaves);
    if (wgs > 0x7FFFFFFFll) return fail_arg(CARLE_E_DIMS, "carle_ca_step: batch too large");
    constexpr int64_t kGridBytes = (int64_t)WideGeom<NS, RW>::kW * WideGeom<NS, RW>::kW * (int64_t)sizeof(typename C::T);
    const bool nt_st = stream_stores(n * kGridBytes, sizeof(typename C::T));
    const dim3 grid(persistent_blocks(wgs)), block(64 * kWideWaves);
    if (packed) {
        // fp32 cells with the packed copy (PK): packed->bits_out always, packed->bits_in selects PK = 2; `env`: one launch
        if (sizeof(typename C::T) != 4) return fail_arg(CARLE_E_DIMS, "carle_env_step: the packed copy goes with fp32 cells");
        const EnvArgs ea = *packed;
        const RandAct ra;
        const int aah = action ? ah : 0, aaw = action ? aw : 0;
#define CARLE_WIDE_PK(ACT, NTS, ENVF, PKV) \
        hipLaunchKernelGGL((ca_step_wide_kernel<C, A, ACT, NS, RW, false, NTS, ENVF, PKV>), grid, block, 0, s, u_in, action, u_out, n, \
                           aah, aaw, rule, status, slot, ra, ea)
#define CARLE_WIDE_PK2(ENVF, PKV) \
        do { \
            if (action && nt_st) CARLE_WIDE_PK(true, true, ENVF, PKV); \
            else if (action) CARLE_WIDE_PK(true, false, ENVF, PKV); \
            else if (nt_st) CARLE_WIDE_PK(false, true, ENVF, PKV); \
            else CARLE_WIDE_PK(false, false, ENVF, PKV); \
        } while (0)
        if (env && ea.bits_in) CARLE_WIDE_PK2(true, 2);
        else if (env) CARLE_WIDE_PK2(true, 1);
        else if (ea.bits_in) CARLE_WIDE_PK2(false, 2);
        else CARLE_WIDE_PK2(false, 1);
#undef CARLE_WIDE_PK2
#undef CARLE_WIDE_PK
        return check_launch("carle_env_step(wide, packed copy)");
    }
    if (env) {                                       // the one-launch step (CARLE_STEP_DEFERRED)
        const EnvArgs ea = *env;
        const RandAct ra;
        if (action && nt_st)
            hipLaunchKernelGGL((ca_step_wide_kernel<C, A, true, NS, RW, false, true, true>), grid, block, 0, s, u_in, action, u_out, n,
                               ah, aw, rule, status, slot, ra, ea);
        else if (action)
            hipLaunchKernelGGL((ca_step_wide_kernel<C, A, true, NS, RW, false, false, true>), grid, block, 0, s, u_in, action, u_out, n,
                               ah, aw, rule, status, slot, ra, ea);
        else if (nt_st)
            hipLaunchKernelGGL((ca_step_wide_kernel<C, A, false, NS, RW, false, true, true>), grid, block, 0, s, u_in, action, u_out, n,
                               0, 0, rule, status, slot, ra, ea);
        else
            hipLaunchKernelGGL((ca_step_wide_kernel<C, A, false, NS, RW, false, false, true>), grid, block, 0, s, u_in, action, u_out, n,
                               0, 0, rule, status, slot, ra, ea);
        return check_launch("carle_env_step(wide, one launch)");
    }
    if (action && nt_st)
        hipLaunchKernelGGL((ca_step_wide_kernel<C, A, true, NS, RW, false, true>), grid, block, 0, s, u_in, action, u_out, n,
                           ah, aw, rule, status, slot);
    else if (action)
        hipLaunchKernelGGL((ca_step_wide_kernel<C, A, true, NS, RW, false, false>), grid, block, 0, s, u_in, action, u_out, n,
                           ah, aw, rule, status, slot);
    else if (nt_st)
        hipLaunchKernelGGL((ca_step_wide_kernel<C, A, false, NS, RW, false, true>), grid, block, 0, s, u_in, action, u_out, n,
                           0, 0, rule, status, slot);
    else
        hipLaunchKernelGGL((ca_step_wide_kernel<C, A, false, NS, RW, false, false>), grid, block, 0, s, u_in, action, u_out, n,
                           0, 0, rule, status, slot);
    return check_launch("carle_ca_step(wide)");
}

// ---------------------------------------------------------------------------------------------
// generic path: any even square grid, one thread per cell
// ---------------------------------------------------------------------------------------------
template <class C, class A, bool HAS_ACT>
__global__ __launch_bounds__(256) void ca_step_generic_kernel(
    const typename C::T *__restrict__ u_in, const typename A::T *__restrict__ act,
    typename C::T *__restrict__ u_out, int64_t n, int h, int w, int ah, int aw, uint32_t rule,
    int32_t *__restrict__ status, int slot) {
    const int64_t cells = n * (int64_t)h * w;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int oy = (h - ah) >> 1, ox = (w - aw) >> 1;
    ActionSeen seen;
    if (idx < cells) {
        const int x = (int)(idx % w);
        const int y = (int)((idx / w) % h);
        const int64_t g = idx / ((int64_t)h * w);
        const typename C::T *gu = u_in + g * (int64_t)h * w;
        const typename A::T *ga = HAS_ACT ? act + g * (int64_t)ah * aw : nullptr;
        auto alive = [&](int yy, int xx) -> uint32_t {
            uint32_t a = C::alive(gu[yy * w + xx]);
            if (HAS_ACT) {                           // unconditional load (element 0 outside the window): no wait per tap
                const int ay = yy - oy, ax = xx - ox;
                const bool in = ay >= 0 && ay < ah && ax >= 0 && ax < aw;
                const typename A::T t = ga[in ? ay * aw + ax : 0];
                a ^= in ? A::alive(t) : 0u;
            }
            return a;
        };
        const int ym = y == 0 ? h - 1 : y - 1, yp = y == h - 1 ? 0 : y + 1;
        const int xm = x == 0 ? w - 1 : x - 1, xp = x == w - 1 ? 0 : x + 1;
        const uint32_t cnt = alive(ym, xm) + alive(ym, x) + alive(ym, xp) + alive(y, xm) +
                             alive(y, xp) + alive(yp, xm) + alive(yp, x) + alive(yp, xp);
        const uint32_t c = alive(y, x);
        const uint32_t nxt = (rule >> (c ? cnt : 9 + cnt)) & 1u;
        u_out[idx] = C::make(nxt);
        const int ay = y - oy, ax = x - ox;           // this cell owns its own action element
        if (HAS_ACT && ay >= 0 && ay < ah && ax >= 0 && ax < aw) seen.add(A::raw(ga[ay * aw + ax]));
    }
    if (HAS_ACT) publish_flags<A>(seen, status, slot);
}

// ---------------------------------------------------------------------------------------------
// toggle only (CARLE.apply_action called on its own, carle/env.py:150-182): elementwise
// ---------------------------------------------------------------------------------------------
template <class C, class A>
__global__ __launch_bounds__(256) void ca_toggle_kernel(
    const typename C::T *__restrict__ u_in, const typename A::T *__restrict__ act,
    typename C::T *__restrict__ u_out, int64_t n, int h, int w, int ah, int aw) {
    const int64_t cells = n * (int64_t)h * w;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= cells) return;
    const int x = (int)(idx % w);
    const int y = (int)((idx / w) % h);
    const int64_t g = idx / ((int64_t)h * w);
    uint32_t a = C::alive(u_in[idx]);
    const int ay = y - ((h - ah) >> 1), ax = x - ((w - aw) >> 1);
    if (ay >= 0 && ay < ah && ax >= 0 && ax < aw) a ^= A::alive(act[(g * ah + ay) * (int64_t)aw + ax]);
    u_out[idx] = C::make(a);
}

}  // namespace carle

using namespace carle;

namespace carle {
// validation + dispatch shared by the fp32 and uint8 entry points (`who` prefixes error texts)
template <class C, class A>
static int ca_step_cells(const typename C::T *u_in, const typename A::T *action, typename C::T *u_out,
                         int64_t n, int h, int w, int ah, int aw, uint32_t rule, int32_t *status,
                         int slot, void *stream, const char *who) {
    auto fail = [&](int code, const char *what) {
        char buf[200];
        snprintf(buf, sizeof buf, "%s: %s", who, what);
        return fail_arg(code, buf);
    };
    if (n == 0) return 0;                      // empty batch: nothing to launch, pointers may be NULL
    if (!u_in || !u_out) return fail(CARLE_E_NULL, "u_in/u_out is NULL");
    if (int e = check_dims(n, h, w, action != nullptr, ah, aw)) return e;
    if (rule >> 18) return fail(CARLE_E_RULE, "rule mask exceeds 18 bits");
    if (slot != 0 && slot != 1) return fail(CARLE_E_DIMS, "slot must be 0 or 1");
    if (!aligned16(u_in) || !aligned16(u_out) || (action && !aligned16(action)))
        return fail(CARLE_E_ALIGN, "pointers must be 16-byte aligned");
    const size_t bytes = (size_t)n * h * w * sizeof(typename C::T);
    if (overlap(u_in, bytes, u_out, bytes)) return fail(CARLE_E_ALIAS, "u_in and u_out overlap");
    hipStream_t s = static_cast<hipStream_t>(stream);

    if (h == 64 && w == 64 && (!action || (aw % 8) == 0)) {
        // one wavefront per workgroup, non-temporal loads and stores: measured best of the
        // {plain, nt} x {64,128,256,512}-thread variants (tools/kbench, profiles/r01_kbench_f32_64.txt)
        if (n > 0x7FFFFFFFll) return fail(CARLE_E_DIMS, "batch too large");
        const unsigned blocks = (unsigned)n;
        RandAct ra;
        const bool nt_st = stream_stores((int64_t)bytes, sizeof(typename C::T));
        if (action && nt_st)
            hipLaunchKernelGGL((ca_step_64_kernel<C, A, true, true, true, 64>), dim3(blocks), dim3(64), 0,
                               s, u_in, action, u_out, n, aw, rule, status, slot, ra);
        else if (action)
            hipLaunchKernelGGL((ca_step_64_kernel<C, A, true, true, false, 64>), dim3(blocks), dim3(64), 0,
                               s, u_in, action, u_out, n, aw, rule, status, slot, ra);
        else if (nt_st)
            hipLaunchKernelGGL((ca_step_64_kernel<C, A, false, true, true, 64>), dim3(blocks), dim3(64), 0,
                               s, u_in, action, u_out, n, 0, rule, status, slot, ra);
        else
            hipLaunchKernelGGL((ca_step_64_kernel<C, A, false, true, false, 64>), dim3(blocks), dim3(64), 0,
                               s, u_in, action, u_out, n, 0, rule, status, slot, ra);
        return check_launch(who);
    }
    // 256x256 / 512x512: wavefront-per-strip path; needs a 16-byte aligned action window
    if ((w == 256 || w == 512) && (!action || ((aw % 4) == 0 && ((w - aw) % 8) == 0))) {
        if (w == 256)
            return launch_wide<C, A, 1, kWideRW>(u_in, action, u_out, n, ah, aw, rule, status, slot, s);
        return launch_wide<C, A, 2, kWideRW>(u_in, action, u_out, n, ah, aw, rule, status, slot, s);
    }
    const int64_t cells = n * (int64_t)h * w;
    const int64_t blocks = (cells + 255) / 256;
    if (blocks > 0x7FFFFFFFll) return fail(CARLE_E_DIMS, "batch too large");
    if (action)
        hipLaunchKernelGGL((ca_step_generic_kernel<C, A, true>), dim3((unsigned)blocks), dim3(256), 0, s, u_in,
                           action, u_out, n, h, w, ah, aw, rule, status, slot);
    else
        hipLaunchKernelGGL((ca_step_generic_kernel<C, A, false>), dim3((unsigned)blocks), dim3(256), 0, s, u_in,
                           action, u_out, n, h, w, 0, 0, rule, status, slot);
    return check_launch(who);
}

// CARLE_STEP_DEFERRED: the whole step of a 64x64 batch in ONE launch (ca_step_64_kernel<..., ENV>); -1000 when the
// geometry does not qualify (the caller then takes step kernel + epilogue kernel).
template <class C, class A>
static int env_step_one_launch(const typename C::T *u_in, const typename A::T *action, typename C::T *u_out,
                               float *reward, int64_t n, int h, int w, int ah, int aw, uint32_t rule, int32_t *status,
                               int slot, int step_flags, void *stream, const char *who, uint64_t *bits_out = nullptr,
                               int *bits_written = nullptr, const uint64_t *bits_in = nullptr) {
    if (bits_written) *bits_written = 0;
    const bool fast64 = h == 64 && w == 64 && (!action || (aw % 8) == 0);
    const bool wide = h == w && (w == 256 || w == 512) && (!action || ((aw % 4) == 0 && ((w - aw) % 8) == 0));
    if ((!fast64 && !wide) || n < 1 || n > 0x7FFFFFFFll || !status) return -1000;
    if (knobs().two_launch_step > 0) return -1000;                     // A/B switch (CARLE_TWO_LAUNCH_STEP)
    // 256 / 512 grids: the one-launch form pays where a step is bound by its dispatches (the reference's default 16
    // instances: 16.6 -> 13.1 us per step); at N = 4096 x 256x256 the sample load and the verdict exchange cost the
    // HBM-bound kernel 1.3 % (350.5 against 348.5 us for step kernel + epilogue) -- above 32 MiB per state array the
    // step keeps its two launches
    if (wide && (int64_t)n * h * w * (int64_t)sizeof(typename C::T) > (32ll << 20)) return -1000;
    auto fail = [&](int code, const char *what) {
        char buf[200];
        snprintf(buf, sizeof buf, "%s: %s", who, what);
        return fail_arg(code, buf);
    };
    if (bits_in && !((fast64 || wide) && bits_out && sizeof(typename C::T) == 4))
        return fail(CARLE_E_DIMS, "a packed input state: 64 / 256 / 512-column fp32 grids, with bits_out");
    if ((!u_in && !bits_in) || !u_out) return fail(CARLE_E_NULL, "u_in/u_out is NULL");
    if (int e = check_dims(n, h, w, action != nullptr, ah, aw)) return e;
    if (rule >> 18) return fail(CARLE_E_RULE, "rule mask exceeds 18 bits");
    if (slot != 0 && slot != 1) return fail(CARLE_E_DIMS, "slot must be 0 or 1");
    if ((u_in && !aligned16(u_in)) || !aligned16(u_out) || (action && !aligned16(action)) || !aligned16(status))
        return fail(CARLE_E_ALIGN, "pointers must be 16-byte aligned");
    const size_t bytes = (size_t)n * h * w * sizeof(typename C::T);
    if (u_in && !bits_in && overlap(u_in, bytes, u_out, bytes)) return fail(CARLE_E_ALIAS, "u_in and u_out overlap");
    if (bits_in && ((reinterpret_cast<uintptr_t>(bits_in) & 7u) != 0 ||
                    overlap(bits_in, (size_t)n * h * (w / 8), bits_out, (size_t)n * h * (w / 8))))
        return fail(CARLE_E_ALIAS, "bits_in must be 8-byte aligned and must not overlap bits_out");
    EnvArgs ea;
    ea.reward = reward;
    ea.publish = (step_flags & CARLE_STEP_PUBLISH_FLAGS) ? 1 : 0;
    ea.merge_moments = (step_flags & CARLE_STEP_MERGE_MOMENTS) ? 1 : 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (wide) {
        if (bits_out) {
            if ((reinterpret_cast<uintptr_t>(bits_out) & 7u) != 0) return fail(CARLE_E_ALIGN, "bits_out must be 8-byte aligned");
            ea.bits_out = bits_out;
            ea.bits_in = bits_in;
            if (bits_written) *bits_written = 1;
        }
        const EnvArgs *pk = bits_in ? &ea : nullptr;      // (fp32 in + copy out keeps the round-3 store of the ENV instantiation)
        if (w == 256) return launch_wide<C, A, 1, kWideRW>(u_in, action, u_out, n, ah, aw, rule, status, slot, s, &ea, pk);
        return launch_wide<C, A, 2, kWideRW>(u_in, action, u_out, n, ah, aw, rule, status, slot, s, &ea, pk);
    }
    const dim3 grid((unsigned)n), block(64);
    const bool nt_st = stream_stores((int64_t)bytes, sizeof(typename C::T));
    const RandAct ra;
    if (bits_out && sizeof(typename C::T) == 4) {      // fp32 cells under a reward net: + the bit-packed copy (BITS)
        if ((reinterpret_cast<uintptr_t>(bits_out) & 7u) != 0) return fail(CARLE_E_ALIGN, "bits_out must be 8-byte aligned");
        ea.bits_out = bits_out;
        if (bits_written) *bits_written = 1;
        if (bits_in) {                                 // the state comes in packed: u_in is not read
            ea.bits_in = bits_in;
            // with an action, 3072 < n <= 4096: two grids per wavefront (CARLE_PACKED_GRIDS = 1 / 2 forces one / two)
            const int64_t kg = knobs().packed_grids;
            if (action && n >= 2 && (kg == 2 || (kg != 1 && n > 3072 && n <= 4096))) {
                if constexpr (sizeof(typename C::T) == 4 && sizeof(typename A::T) == 4) {
                    const dim3 grid2((unsigned)((n + 1) / 2));
                    if (nt_st)
                        hipLaunchKernelGGL((ca_step_64_packed2_kernel<A, true>), grid2, block, 0, s, action, u_out, n, aw, rule,
                                           status, slot, ea);
                    else
                        hipLaunchKernelGGL((ca_step_64_packed2_kernel<A, false>), grid2, block, 0, s, action, u_out, n, aw, rule,
                                           status, slot, ea);
                    return check_launch(who);
                }
            }
            if (action && nt_st)
                hipLaunchKernelGGL((ca_step_64_kernel<C, A, true, true, true, 64, false, true, true, true>), grid, block, 0, s,
                                   u_in, action, u_out, n, aw, rule, status, slot, ra, ea);
            else if (action)
                hipLaunchKernelGGL((ca_step_64_kernel<C, A, true, true, false, 64, false, true, true, true>), grid, block, 0, s,
                                   u_in, action, u_out, n, aw, rule, status, slot, ra, ea);
            else if (nt_st)
                hipLaunchKernelGGL((ca_step_64_kernel<C, A, false, true, true, 64, false, true, true, true>), grid, block, 0, s,
                                   u_in, action, u_out, n, 0, rule, status, slot, ra, ea);
            else
                hipLaunchKernelGGL((ca_step_64_kernel<C, A, false, true, false, 64, false, true, true, true>), grid, block, 0, s,
                                   u_in, action, u_out, n, 0, rule, status, slot, ra, ea);
            return check_launch(who);
        }
        if (action && nt_st)
            hipLaunchKernelGGL((ca_step_64_kernel<C, A, true, true, true, 64, false, true, true>), grid, block, 0, s, u_in,
                               action, u_out, n, aw, rule, status, slot, ra, ea);
        else if (action)
            hipLaunchKernelGGL((ca_step_64_kernel<C, A, true, true, false, 64, false, true, true>), grid, block, 0, s, u_in,
                               action, u_out, n, aw, rule, status, slot, ra, ea);
        else if (nt_st)
            hipLaunchKernelGGL((ca_step_64_kernel<C, A, false, true, true, 64, false, true, true>), grid, block, 0, s, u_in,
                               action, u_out, n, 0, rule, status, slot, ra, ea);
        else
            hipLaunchKernelGGL((ca_step_64_kernel<C, A, false, true, false, 64, false, true, true>), grid, block, 0, s, u_in,
                               action, u_out, n, 0, rule, status, slot, ra, ea);
        return check_launch(who);
    }
    if (action && nt_st)
        hipLaunchKernelGGL((ca_step_64_kernel<C, A, true, true, true, 64, false, true>), grid, block, 0, s, u_in, action,
                           u_out, n, aw, rule, status, slot, ra, ea);
    else if (action)
        hipLaunchKernelGGL((ca_step_64_kernel<C, A, true, true, false, 64, false, true>), grid, block, 0, s, u_in, action,
                           u_out, n, aw, rule, status, slot, ra, ea);
    else if (nt_st)
        hipLaunchKernelGGL((ca_step_64_kernel<C, A, false, true, true, 64, false, true>), grid, block, 0, s, u_in, action,
                           u_out, n, 0, rule, status, slot, ra, ea);
    else
        hipLaunchKernelGGL((ca_step_64_kernel<C, A, false, true, false, 64, false, true>), grid, block, 0, s, u_in, action,
                           u_out, n, 0, rule, status, slot, ra, ea);
    return check_launch(who);
}

template <class C, class A>
static int ca_toggle_cells(const typename C::T *u_in, const typename A::T *action, typename C::T *u_out,
                           int64_t n, int h, int w, int ah, int aw, void *stream, const char *who) {
    if (!u_in || !u_out || !action) return fail_arg(CARLE_E_NULL, "carle_ca_toggle: NULL pointer");
    if (int e = check_dims(n, h, w, true, ah, aw)) return e;
    if (n == 0) return 0;
    const int64_t cells = n * (int64_t)h * w;
    const int64_t blocks = (cells + 255) / 256;
    if (blocks > 0x7FFFFFFFll) return fail_arg(CARLE_E_DIMS, "carle_ca_toggle: batch too large");
    hipLaunchKernelGGL((ca_toggle_kernel<C, A>), dim3((unsigned)blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), u_in, action, u_out, n, h, w, ah, aw);
    return check_launch(who);
}
}  // namespace carle

namespace carle {
// Whole CARLE.step of a small 64x64 fp32 batch in one launch (see ca_env_step_small64_kernel);
// returns -1000 when the geometry does not qualify (the caller then takes the two-launch path).
// rand != nullptr: in-kernel Bernoulli toggles instead of an action tensor.
int env_step_small64(const float *u_in, const float *action, float *u_out, float *reward, int64_t n, int h,
                     int w, int ah, int aw, uint32_t rule, int32_t *status, int slot, int step_flags,
                     const RandAct *rand, void *stream) {
    const bool has_act = action != nullptr || rand != nullptr;
    if (h != 64 || w != 64 || n < 1 || n > 16 || !status || (has_act && (aw % 8) != 0)) return -1000;
    // in-kernel toggles are ~100 instructions per quad: more than four grids of them on ONE compute unit take longer
    // than the two launches of the general path (N=16: 19.6 against ~10 us per step)
    if (rand && n > 4) return -1000;
    if (!u_in || !u_out) return fail_arg(CARLE_E_NULL, "carle_env_step_f32: u_in/u_out is NULL");
    if (int e = check_dims(n, h, w, has_act, ah, aw)) return e;
    if (rule >> 18) return fail_arg(CARLE_E_RULE, "carle_env_step_f32: rule mask exceeds 18 bits");
    if (slot != 0 && slot != 1) return fail_arg(CARLE_E_DIMS, "carle_env_step_f32: slot must be 0 or 1");
    if (!aligned16(u_in) || !aligned16(u_out) || (action && !aligned16(action)) || !aligned16(status))
        return fail_arg(CARLE_E_ALIGN, "carle_env_step_f32: pointers must be 16-byte aligned");
    const size_t bytes = (size_t)n * h * w * sizeof(float);
    if (overlap(u_in, bytes, u_out, bytes)) return fail_arg(CARLE_E_ALIAS, "carle_env_step_f32: u_in and u_out overlap");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid(1), block((unsigned)(64 * n));
    const int publish = rand ? 1 : (step_flags & CARLE_STEP_PUBLISH_FLAGS) ? 1 : 0;
    const int merge = (step_flags & CARLE_STEP_MERGE_MOMENTS) ? 1 : 0;
    if (rand)
        hipLaunchKernelGGL((ca_env_step_small64_kernel<true, true>), grid, block, 0, s, u_in, action, u_out, reward, n,
                           aw, rule, status, slot, publish, merge, *rand);
    else if (action)
        hipLaunchKernelGGL((ca_env_step_small64_kernel<true, false>), grid, block, 0, s, u_in, action, u_out, reward,
                           n, aw, rule, status, slot, publish, merge, RandAct());
    else
        hipLaunchKernelGGL((ca_env_step_small64_kernel<false, false>), grid, block, 0, s, u_in, action, u_out, reward,
                           n, 0, rule, status, slot, publish, merge, RandAct());
    return check_launch("carle_env_step_f32(small batch)");
}

// k steps of a small batch in one launch (ca_env_run_small64_kernel); -1000 when the geometry does not qualify.
// `actions`: k stacked (N,1,AW,AW) tensors, or nullptr with rand (in-kernel toggles, Philox offset + t) or without.
int env_run_small64(const float *u_in, const float *actions, float *u_out, float *reward, int64_t n, int h, int w, int ah,
                    int aw, uint32_t rule, float p, uint64_t seed, uint64_t offset, int k, int32_t *status, int slot0,
                    int step_flags, void *stream) {
    RandAct ra;
    ra.p = p; ra.seed = seed; ra.offset = offset;
    const RandAct *rand = (!actions && p >= 0.0f) ? &ra : nullptr;
    const bool has_act = actions != nullptr || rand != nullptr;
    if (!has_act) { ah = 0; aw = 0; }
    if (h != 64 || w != 64 || n < 1 || n > 16 || k < 1 || !status || (has_act && (aw % 8) != 0)) return -1000;
    if (rand && n > 4) return -1000;                       // as env_step_small64: the draws of many grids on one CU
    if (!u_in || !u_out) return fail_arg(CARLE_E_NULL, "carle_env_run_f32: buffer is NULL");
    if (int e = check_dims(n, h, w, has_act, ah, aw)) return e;
    if (rule >> 18) return fail_arg(CARLE_E_RULE, "carle_env_run_f32: rule mask exceeds 18 bits");
    if (!aligned16(u_in) || !aligned16(u_out) || (actions && !aligned16(actions)) || !aligned16(status))
        return fail_arg(CARLE_E_ALIGN, "carle_env_run_f32: pointers must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid(1), block((unsigned)(64 * n));
    const int merge = (step_flags & CARLE_STEP_MERGE_MOMENTS) ? 1 : 0;
    if (rand)
        hipLaunchKernelGGL((ca_env_run_small64_kernel<true, true>), grid, block, 0, s, u_in, actions, u_out, reward, n, aw,
                           rule, status, slot0, merge, *rand, k);
    else if (actions)
        hipLaunchKernelGGL((ca_env_run_small64_kernel<true, false>), grid, block, 0, s, u_in, actions, u_out, reward, n,
                           aw, rule, status, slot0, merge, RandAct(), k);
    else
        hipLaunchKernelGGL((ca_env_run_small64_kernel<false, false>), grid, block, 0, s, u_in, actions, u_out, reward, n,
                           0, rule, status, slot0, merge, RandAct(), k);
    return check_launch("carle_env_run_f32(small batch)");
}

// k action-free steps of a 64x64 fp32 batch in one launch (ca_run_noaction64_kernel); -1000 when it does not apply.
int env_run_noaction64(const float *u_in, float *u_out, float *reward, int64_t n, int h, int w, uint32_t rule, int k,
                       int32_t *status, int slot0, int step_flags, void *stream) {
    if (h != 64 || w != 64 || n < 1 || k < 1 || !status || n > 0x7FFFFFFFll) return -1000;
    if (!u_in || !u_out) return fail_arg(CARLE_E_NULL, "carle_env_run_f32: buffer is NULL");
    if (rule >> 18) return fail_arg(CARLE_E_RULE, "carle_env_run_f32: rule mask exceeds 18 bits");
    if (!aligned16(u_in) || !aligned16(u_out) || !aligned16(status))
        return fail_arg(CARLE_E_ALIGN, "carle_env_run_f32: pointers must be 16-byte aligned");
    const int merge = (step_flags & CARLE_STEP_MERGE_MOMENTS) ? 1 : 0;
    hipLaunchKernelGGL(ca_run_noaction64_kernel, dim3((unsigned)n), dim3(64), 0, static_cast<hipStream_t>(stream), u_in, u_out,
                       reward, n, rule, status, slot0, merge, k, stream_stores(n * 16384, 4) ? 1 : 0);
    return check_launch("carle_env_run_f32(no action)");
}

int env_step_small64_rand(const float *u_in, float *u_out, float *reward, int64_t n, int h, int w, int ah, int aw,
                          uint32_t rule, float p, uint64_t seed, uint64_t offset, float *action_out,
                          int32_t *status, int slot, int step_flags, void *stream) {
    RandAct ra;
    ra.p = p; ra.seed = seed; ra.offset = offset; ra.action_out = action_out;
    if (action_out && !aligned16(action_out)) return -1000;
    return env_step_small64(u_in, nullptr, u_out, reward, n, h, w, ah, aw, rule, status, slot, step_flags, &ra, stream);
}
}  // namespace carle

namespace carle {
// The wide step kernel (no epilogue) of N x 256x256 / 512x512 fp32 grids with the packed copy: bits_out always, the input state
// from bits_in when given (u_in is then not read).  -1000: geometry without the wide kernel.
int ca_step_wide_packed_f32(const float *u_in, const uint64_t *bits_in, const float *action, float *u_out, uint64_t *bits_out,
                            int64_t n, int h, int w, int ah, int aw, uint32_t rule, int32_t *status, int slot, void *stream) {
    const bool wide = h == w && (w == 256 || w == 512) && (!action || ((aw % 4) == 0 && ((w - aw) % 8) == 0));
    if (!wide || n < 1 || !bits_out) return -1000;
    if ((!u_in && !bits_in) || !u_out) return fail_arg(CARLE_E_NULL, "carle_env_step_f32 (packed copy): u_in/u_out is NULL");
    if (int e = check_dims(n, h, w, action != nullptr, ah, aw)) return e;
    if (rule >> 18) return fail_arg(CARLE_E_RULE, "carle_env_step_f32 (packed copy): rule mask exceeds 18 bits");
    if (slot != 0 && slot != 1) return fail_arg(CARLE_E_DIMS, "carle_env_step_f32 (packed copy): slot must be 0 or 1");
    if ((u_in && !aligned16(u_in)) || !aligned16(u_out) || (action && !aligned16(action)) || !aligned16(bits_out) ||
        (bits_in && !aligned16(bits_in)))
        return fail_arg(CARLE_E_ALIGN, "carle_env_step_f32 (packed copy): pointers must be 16-byte aligned");
    const size_t bytes = (size_t)n * h * w * sizeof(float), pbytes = (size_t)n * h * (w / 8);
    if (u_in && !bits_in && overlap(u_in, bytes, u_out, bytes))
        return fail_arg(CARLE_E_ALIAS, "carle_env_step_f32 (packed copy): u_in and u_out overlap");
    if (bits_in && overlap(bits_in, pbytes, bits_out, pbytes))
        return fail_arg(CARLE_E_ALIAS, "carle_env_step_f32 (packed copy): bits_in and bits_out overlap");
    EnvArgs ea;
    ea.bits_in = bits_in;
    ea.bits_out = bits_out;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (w == 256) return launch_wide<F32Cells, F32Cells, 1, kWideRW>(u_in, action, u_out, n, ah, aw, rule, status, slot, s, nullptr, &ea);
    return launch_wide<F32Cells, F32Cells, 2, kWideRW>(u_in, action, u_out, n, ah, aw, rule, status, slot, s, nullptr, &ea);
}

int env_step_one_launch_f32(const float *u_in, const float *action, float *u_out, float *reward, int64_t n, int h,
                            int w, int ah, int aw, uint32_t rule, int32_t *status, int slot, int step_flags,
                            void *stream, uint64_t *bits_out, int *bits_written, const uint64_t *bits_in) {
    return env_step_one_launch<F32Cells, F32Cells>(u_in, action, u_out, reward, n, h, w, ah, aw, rule, status, slot,
                                                   step_flags, stream, "carle_env_step_f32", bits_out, bits_written, bits_in);
}
int env_step_one_launch_u8(const uint8_t *u_in, const void *action, int action_fmt, uint8_t *u_out, float *reward,
                           int64_t n, int h, int w, int ah, int aw, uint32_t rule, int32_t *status, int slot,
                           int step_flags, void *stream) {
    if (action_fmt == CARLE_FMT_U8)
        return env_step_one_launch<U8Cells, U8Cells>(u_in, static_cast<const uint8_t *>(action), u_out, reward, n, h, w,
                                                     ah, aw, rule, status, slot, step_flags, stream, "carle_env_step_u8");
    if (action_fmt == CARLE_FMT_F32)
        return env_step_one_launch<U8Cells, F32Cells>(u_in, static_cast<const float *>(action), u_out, reward, n, h, w,
                                                      ah, aw, rule, status, slot, step_flags, stream, "carle_env_step_u8");
    return -1000;
}
}  // namespace carle

extern "C" int carle_ca_step_f32(const float *u_in, const float *action, float *u_out, int64_t n,
                                 int h, int w, int ah, int aw, uint32_t rule, int32_t *status,
                                 int slot, void *stream) {
    return ca_step_cells<F32Cells, F32Cells>(u_in, action, u_out, n, h, w, ah, aw, rule, status, slot,
                                             stream, "carle_ca_step_f32");
}

// uint8 state; the action is fp32 (what the reference's agents emit) or uint8
extern "C" int carle_ca_step_u8(const uint8_t *u_in, const void *action, int action_fmt, uint8_t *u_out,
                                int64_t n, int h, int w, int ah, int aw, uint32_t rule, int32_t *status,
                                int slot, void *stream) {
    if (action_fmt == CARLE_FMT_U8)
        return ca_step_cells<U8Cells, U8Cells>(u_in, static_cast<const uint8_t *>(action), u_out, n, h, w,
                                               ah, aw, rule, status, slot, stream,
                                               "carle_ca_step_u8");
    if (action_fmt == CARLE_FMT_F32)
        return ca_step_cells<U8Cells, F32Cells>(u_in, static_cast<const float *>(action), u_out, n, h, w,
                                                ah, aw, rule, status, slot, stream,
                                                "carle_ca_step_u8");
    return fail_arg(CARLE_E_FORMAT, "carle_ca_step_u8: action_fmt must be CARLE_FMT_F32 or CARLE_FMT_U8");
}

extern "C" int carle_ca_toggle_f32(const float *u_in, const float *action, float *u_out,
                                   int64_t n, int h, int w, int ah, int aw, void *stream) {
    return ca_toggle_cells<F32Cells, F32Cells>(u_in, action, u_out, n, h, w, ah, aw, stream,
                                               "carle_ca_toggle_f32");
}

extern "C" int carle_ca_toggle_u8(const uint8_t *u_in, const void *action, int action_fmt,
                                  uint8_t *u_out, int64_t n, int h, int w, int ah, int aw, void *stream) {
    if (action_fmt == CARLE_FMT_U8)
        return ca_toggle_cells<U8Cells, U8Cells>(u_in, static_cast<const uint8_t *>(action), u_out, n, h,
                                                 w, ah, aw, stream, "carle_ca_toggle_u8");
    if (action_fmt == CARLE_FMT_F32)
        return ca_toggle_cells<U8Cells, F32Cells>(u_in, static_cast<const float *>(action), u_out, n, h,
                                                  w, ah, aw, stream, "carle_ca_toggle_u8");
    return fail_arg(CARLE_E_FORMAT, "carle_ca_toggle_u8: action_fmt must be CARLE_FMT_F32 or CARLE_FMT_U8");
}

// CARLE.step with the device-side RandomAgent fused in (carle/agents.py:35-42 + env.py:188-242):
// Bernoulli(p) toggles over the centred AW x AW window, drawn in the step kernel; fp32 or uint8 state.
namespace carle {
template <class C>
static int ca_step_cells_rand(const typename C::T *u_in, typename C::T *u_out, int64_t n, int h, int w, int ah,
                              int aw, uint32_t rule, float p, uint64_t seed, uint64_t offset, float *action_out,
                              int32_t *status, int slot, void *stream, const char *who) {
    auto fail = [&](int code, const char *what) {
        char buf[320];
        snprintf(buf, sizeof buf, "%s: %s", who, what);
        return fail_arg(code, buf);
    };
    if (n == 0) return 0;
    if (!u_in || !u_out) return fail(CARLE_E_NULL, "u_in/u_out is NULL");
    if (int e = check_dims(n, h, w, true, ah, aw)) return e;
    const bool fast64 = h == 64 && (aw % 8) == 0;
    const bool wide = (h == 256 || h == 512) && (aw % 4) == 0 && ((w - aw) % 8) == 0;
    if (!fast64 && !wide)
        return fail(CARLE_E_DIMS, "fused random toggles need 64x64 grids with AW % 8 == 0, or 256 / 512 grids with "
                                  "AW % 4 == 0 and (W-AW) % 8 == 0 (generate the action with carle_bernoulli_f32 "
                                  "otherwise)");
    if (rule >> 18) return fail(CARLE_E_RULE, "rule mask exceeds 18 bits");
    if (slot != 0 && slot != 1) return fail(CARLE_E_DIMS, "slot must be 0 or 1");
    if (!aligned16(u_in) || !aligned16(u_out) || (action_out && !aligned16(action_out)))
        return fail(CARLE_E_ALIGN, "pointers must be 16-byte aligned");
    const size_t bytes = (size_t)n * h * w * sizeof(typename C::T);
    if (overlap(u_in, bytes, u_out, bytes)) return fail(CARLE_E_ALIAS, "u_in and u_out overlap");
    if (n > 0x7FFFFFFFll) return fail(CARLE_E_DIMS, "batch too large");
    RandAct ra;
    ra.p = p; ra.seed = seed; ra.offset = offset; ra.action_out = action_out;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (wide) {
        const int64_t wgs = n * (h / (16 * kWideRW * kWideWaves));
        if (wgs > 0x7FFFFFFFll) return fail(CARLE_E_DIMS, "batch too large");
        if (h == 256)
            hipLaunchKernelGGL((ca_step_wide_kernel<C, F32Cells, true, 1, kWideRW, true>), dim3(persistent_blocks(wgs)), dim3(64 * kWideWaves), 0, s,
                               u_in, (const float *)nullptr, u_out, n, ah, aw, rule, status, slot, ra);
        else
            hipLaunchKernelGGL((ca_step_wide_kernel<C, F32Cells, true, 2, kWideRW, true>), dim3(persistent_blocks(wgs)), dim3(64 * kWideWaves), 0, s,
                               u_in, (const float *)nullptr, u_out, n, ah, aw, rule, status, slot, ra);
        return check_launch(who);
    }
    if (stream_stores(n * (int64_t)h * w * (int64_t)sizeof(typename C::T), sizeof(typename C::T)))
        hipLaunchKernelGGL((ca_step_64_kernel<C, F32Cells, true, true, true, 64, true>), dim3((unsigned)n), dim3(64), 0, s,
                           u_in, (const float *)nullptr, u_out, n, aw, rule, status, slot, ra);
    else
        hipLaunchKernelGGL((ca_step_64_kernel<C, F32Cells, true, true, false, 64, true>), dim3((unsigned)n), dim3(64), 0, s,
                           u_in, (const float *)nullptr, u_out, n, aw, rule, status, slot, ra);
    return check_launch(who);
}
}  // namespace carle

extern "C" int carle_ca_step_f32_rand(const float *u_in, float *u_out, int64_t n, int h, int w,
                                      int ah, int aw, uint32_t rule, float p, uint64_t seed,
                                      uint64_t offset, float *action_out, int32_t *status, int slot,
                                      void *stream) {
    return ca_step_cells_rand<F32Cells>(u_in, u_out, n, h, w, ah, aw, rule, p, seed, offset, action_out, status, slot,
                                        stream, "carle_ca_step_f32_rand");
}

extern "C" int carle_ca_step_u8_rand(const uint8_t *u_in, uint8_t *u_out, int64_t n, int h, int w, int ah, int aw,
                                     uint32_t rule, float p, uint64_t seed, uint64_t offset, float *action_out,
                                     int32_t *status, int slot, void *stream) {
    return ca_step_cells_rand<U8Cells>(u_in, u_out, n, h, w, ah, aw, rule, p, seed, offset, action_out, status, slot,
                                       stream, "carle_ca_step_u8_rand");
}

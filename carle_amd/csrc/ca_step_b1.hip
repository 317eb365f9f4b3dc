// ca_step_b1.hip -- one CA generation on bit-packed grids (CARLE_FMT_B1) + pack / unpack.
//
// Format: row-major, W/64 little-endian uint64 words per row, bit j of word i = column 64*i+j.
// 0.25 B of HBM traffic per cell update (1/32 of the fp32 drop-in format).
//
// Kernel: one wavefront per strip of 64 rows, lane = row.  A lane holds its whole row (WW = W/64
// words) in registers: west/east neighbours are 1-bit funnel shifts across the row's words
// (wrap = x torus seam), north/south neighbours come from lanes r-1 / r+1 through DPP wave
// rotates (wave_ror:1 / wave_rol:1: a VALU move, no LDS); for 64x64 grids the rotate's own
// wrap-around IS the y torus seam, taller grids load the two rows bordering the strip into lanes
// 0 and 63.  Count = bit-sliced adders, rule = equality masks selected by the SGPR rule bits
// (ca_bits.hpp).  No MFMA, no LDS.
#include "ca_cells.hpp"
#include "common.hpp"
#include "philox.hpp"
#include <stdlib.h>

// Diagnostic build only (tools/kbench/ca_b1_variants.hip defines CARLE_B1_STAMPS): every wave records
// s_memtime per work item at its start, after its loads have arrived, after the compute and at its end.
#ifdef CARLE_B1_STAMPS
__device__ unsigned long long *g_b1_stamps = nullptr;       // [work item][4]
#define B1_STAMP(i) do { if (threadIdx.x == 0 && g_b1_stamps) \
    g_b1_stamps[(size_t)wid * 4 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define B1_STAMP(i) do {} while (0)
#endif

namespace carle {

// Multi-strip grids use plain (cached) loads and stores: a lane reads its row as W/128 loads of 16 B and the
// lines have to stay cached between them -- with the streaming (nt) hint the 512x512 step takes 110 us
// instead of 48 us (measured, round 2).
template <int WW>
__device__ __forceinline__ void load_row(const uint64_t *__restrict__ p, uint64_t (&r)[WW]) {
    if constexpr (WW == 1) {
        r[0] = __builtin_nontemporal_load(p);
    } else {
#pragma unroll
        for (int j = 0; j < WW; j += 2) {
            const uint4 v = *reinterpret_cast<const uint4 *>(p + j);
            r[j] = ((uint64_t)v.y << 32) | v.x;
            r[j + 1] = ((uint64_t)v.w << 32) | v.z;
        }
    }
}

template <int WW>
__device__ __forceinline__ void store_row(uint64_t *__restrict__ p, const uint64_t (&r)[WW]) {
    if constexpr (WW == 1) {
        __builtin_nontemporal_store(r[0], p);
    } else {
#pragma unroll
        for (int j = 0; j < WW; j += 2) {
            uint4 v;
            v.x = (uint32_t)r[j]; v.y = (uint32_t)(r[j] >> 32);
            v.z = (uint32_t)r[j + 1]; v.w = (uint32_t)(r[j + 1] >> 32);
            *reinterpret_cast<uint4 *>(p + j) = v;
        }
    }
}

// XOR the packed action row (AWW words at column offset ox = 32*(WW-AWW), a multiple of 32 bits)
// into a grid row.  `ok` = this lane's row is inside the window.  Every 32-bit half of every grid
// word reads "its" action dword with an UNCONDITIONAL load from a clamped (always valid) address
// and masks the result: 2*WW independent loads that issue back to back (one latency, not 2*WW
// dependent ones as with per-word branches).
template <int WW>
__device__ __forceinline__ void toggle_row(uint64_t (&c)[WW], const uint64_t *__restrict__ arow,
                                           int aww, bool ok, uint64_t &seen_or, uint64_t &seen_and) {
    const uint32_t *a32 = reinterpret_cast<const uint32_t *>(arow);
    const int shift_dwords = WW - aww;                      // ox / 32
    const int ndw = 2 * aww;                                // dwords in the action row
    uint32_t v[2 * WW];
#pragma unroll
    for (int d = 0; d < 2 * WW; ++d) {
        const int i = d - shift_dwords;                     // action dword under grid dword d
        const bool in = ok && i >= 0 && i < ndw;
        const uint32_t x = a32[in ? i : 0];
        v[d] = in ? x : 0u;
    }
    uint32_t o = 0u, n = 0xFFFFFFFFu;
#pragma unroll
    for (int d = 0; d < 2 * WW; ++d) {
        const int i = d - shift_dwords;
        if (ok && i >= 0 && i < ndw) { o |= v[d]; n &= v[d]; }
    }
#pragma unroll
    for (int j = 0; j < WW; ++j) c[j] ^= ((uint64_t)v[2 * j + 1] << 32) | v[2 * j];
    seen_or |= o;
    if (ok) seen_and &= (n == 0xFFFFFFFFu) ? ~0ull : 0ull;
}

// The same for the usual 64-column window (AW == 64: one action word per row, at bit offset
// 32*(WW-1) = the high half of grid word (WW-1)/2 and the low half of the next one): two dwords
// instead of 2*WW, light enough to live inside the full-occupancy instantiation.
template <int WW>
__device__ __forceinline__ void toggle_row64(uint64_t (&c)[WW], const uint64_t *__restrict__ arow, bool ok,
                                             uint32_t &seen_or, uint32_t &seen_and) {
    const uint32_t *a32 = reinterpret_cast<const uint32_t *>(arow);
    const uint32_t lo = a32[0], hi = a32[1];                 // arow is clamped to a valid row by the caller
    const uint32_t l = ok ? lo : 0u, h = ok ? hi : 0u;
    if constexpr (WW == 1) {
        c[0] ^= ((uint64_t)h << 32) | l;
    } else {
        c[(WW - 1) / 2] ^= (uint64_t)l << 32;
        c[(WW - 1) / 2 + 1] ^= (uint64_t)h;
    }
    seen_or |= l | h;
    if (ok) seen_and &= lo & hi;
}

// The same with the two action dwords already in registers (the in-kernel random agent).
template <int WW>
__device__ __forceinline__ void toggle_words64(uint64_t (&c)[WW], uint32_t lo, uint32_t hi, bool ok,
                                               uint32_t &seen_or, uint32_t &seen_and) {
    const uint32_t l = ok ? lo : 0u, h = ok ? hi : 0u;
    if constexpr (WW == 1) {
        c[0] ^= ((uint64_t)h << 32) | l;
    } else {
        c[(WW - 1) / 2] ^= (uint64_t)l << 32;
        c[(WW - 1) / 2 + 1] ^= (uint64_t)h;
    }
    seen_or |= l | h;
    if (ok) seen_and &= lo & hi;
}

// RULE_MODE: the two rules of the BASELINE configs get a 2-instruction closed form; everything
// else goes through the rule-independent 17-instruction multiplexer tree (no per-rule branches,
// no extra live registers -- the branchy sparse form measured slower here: 142 VGPRs).
enum { kRuleGeneric = 0, kRuleLife = 1, kRuleHighLife = 2 };
static constexpr uint32_t kLifeMask = (1u << 12) | 0xCu;                  // B3/S23
static constexpr uint32_t kHighLifeMask = (1u << 12) | (1u << 15) | 0xCu;  // B36/S23

template <int RULE_MODE>
__device__ __forceinline__ uint32_t next_state(const Count4<uint32_t> &n, uint32_t c, uint32_t rule) {
    if constexpr (RULE_MODE == kRuleLife) {
        // count==3, or alive and count==2:  s1 & ~s2 & (s0 | c)   (count 8 has s1 == 0)
        return bop3<(kA & ~kB & kC) & 0xFF>(n.s1, n.s2, n.s0 | c);
    } else if constexpr (RULE_MODE == kRuleHighLife) {
        // Life, plus dead and count==6:  s1 & ((~s2 & (s0 | c)) | (s2 & ~s0 & ~c))
        const uint32_t x = bop3<((~kB & (kA | kC)) | (kB & ~kA & ~kC)) & 0xFF>(n.s0, n.s2, c);
        return n.s1 & x;
    } else {
        return apply_rule(n, c, rule);
    }
}

// ---- coalesced strip traffic (WW >= 2, single-step kernel) --------------------------------------------------------
// A 64-row strip is contiguous in memory (64 * 8 * WW bytes).  The wave moves it with fully coalesced 16-byte
// accesses -- lane L of access j touches bytes 1024 j + 16 L, one contiguous KiB per instruction --
// and changes between that layout and lane = row through its own LDS image of the strip, which is simply the
// memory image: write linear / read row-wise on the way in, the reverse on the way out (row-wise accesses of 64-byte
// rows are 2-way bank conflicts: a few dozen cycles per strip).  One wave per workgroup: no barrier, the LDS
// pipeline keeps a wave's accesses in order.  The two rows bordering the strip are one more access (WW / 2 lanes each).
// Before: every lane read its row as WW / 2 accesses of 16 bytes at a stride of 8 * WW bytes (a quarter of every
// 64-byte sector per instruction, cached loads and stores only -- with the streaming hint 110 us instead of 48),
// and the dirty lines of the whole output sat in the L2 until the end-of-kernel write-back.
// Three strip modes, chosen per launch (tools/b1_strip_modes.sh, profiles/r02_kbench_b1_lds.txt; kernel alone, us without /
// with the 64x64 window; rows = row-per-lane, lds = StripLds with default cache policy, nt = StripLds streaming):
//   512x512  N=1024 (32 MiB)  rows 17.4 / 16.7   lds 12.5 / 13.8   nt 11.2 / 16.4
//            N=2048           rows 27.6 / 27.3   lds 21.5 / 23.2   nt 23.5 / 26.4
//            N=4096 (128 MiB) rows 48.0 / 51.8   lds 41.5 / 47.5   nt 47.3 / 51.7     (0.70 / 0.65 -> 0.81 / 0.71 of 8 TB/s)
//            N=6144 (192 MiB) rows 91.3 / 97.2   lds 74.6 / 78.7   nt 71.3 / 77.2
//            N=8192 (256 MiB) rows 120 / 129     lds 101 / 107     nt 96 / 104
//   256x256  N=4096 (32 MiB)  rows 11.5 / 12.3   lds 12.0 / 12.1   nt 13.6 / 14.5
//            N=16384          rows 40.9 / 58.6   lds 41.7 / 53.4   nt 49.0 / 52.6
//            N=32768          rows 108 / 117     lds 101 / 107     nt 98.4 / 105.8
// Up to the size of the Infinity Cache the output of step t is (partly) still cached when step t+1 reads it: no
// streaming hint below 192 MiB per array.  256-column grids UNDER THE 64x64 WINDOW (half of their strips meet it) take the
// hints from 112 MiB (round 4, one wave per strip: N=12288 = 96 MiB 30.9 us without against 34.7 with, N=16384 = 128 MiB
// 50.9 against 45.0, N=20480 62.0 against 54.8; without a window N=16384 stays better without: 38.6 against 42.6; 512-column
// grids at 128 MiB too: 41.8 against 44.9).
static int b1_strip_mode(int64_t n, int ww, bool window = false) {
    int64_t lds_bytes = ww == 8 ? 0 : 64ll << 20, nt_bytes = (ww == 4 && window) ? 112ll << 20 : 192ll << 20;
    if (knobs().b1_lds_min_bytes >= 0) lds_bytes = knobs().b1_lds_min_bytes;      // experiment knobs, read once
    if (knobs().b1_nt_min_bytes >= 0) nt_bytes = knobs().b1_nt_min_bytes;
    const int64_t bytes = n * (int64_t)(64 * ww) * ww * 8;
    if ((ww != 8 && ww != 4) || bytes < lds_bytes) return 0;
    return bytes >= nt_bytes ? 2 : 1;
}
template <bool NT>
__device__ __forceinline__ uint4 ld16s(const uint4 *p) {       // NT: the four dword hints merge into one dwordx4 ... nt
    if (!NT) return *p;
    uint4 v;
    v.x = __builtin_nontemporal_load(&p->x); v.y = __builtin_nontemporal_load(&p->y);
    v.z = __builtin_nontemporal_load(&p->z); v.w = __builtin_nontemporal_load(&p->w);
    return v;
}
template <bool NT>
__device__ __forceinline__ void st16s(uint4 *p, const uint4 &v) {
    if (!NT) { *p = v; return; }
    __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y);
    __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w);
}
template <int WW>
struct StripLds {
    uint4 img[64 * WW / 2];          // the strip
    uint4 halo[2][WW / 2 > 0 ? WW / 2 : 1];   // row above, row below
    uint4 act[32];                   // the grid's packed 64x64 action record (64 row words), see load_strip<.., ACTREC>
    uint2 hsum[2][2 * WW];           // (g0, g1): horizontal triple sums of the two bordering rows per 32-bit chunk (strip_halo_sums)
};

// The horizontal triple sums (west + centre + east, two bit planes) of the strip's two bordering rows, for all chunks at
// once: lane k < NC takes chunk k of the row above, lane NC + k chunk k of the row below (lane = CHUNK here, not row),
// reads the chunk and its two neighbours from the rows' LDS image (after the window toggle), and leaves (g0, g1) in
// l.hsum.  ~10 VALU instructions per strip.  The compute phase then reads ONE 8-byte pair per chunk as the `old` operand
// of its DPP row shifts instead of forming the sums itself in every lane: the two alignbit + two bitop3 per chunk that
// only lanes 0 and 63 had a use for were 64 of the ~600 VALU instructions of a strip -- and the packed kernel is
// VALU-bound at N = 4096 (round 4).
template <int WW>
__device__ __forceinline__ void strip_halo_sums(StripLds<WW> &l, int lane) {
    constexpr int NC = 2 * WW;
    static_assert(2 * NC <= 64, "one lane per chunk of the two rows");
    const int row = (lane >> (NC == 16 ? 4 : NC == 8 ? 3 : NC == 4 ? 2 : 1)) & 1, k = lane & (NC - 1);
    const uint32_t *h32 = reinterpret_cast<const uint32_t *>(l.halo[row]);
    const uint32_t hk = h32[k], hprev = h32[(k + NC - 1) & (NC - 1)], hnext = h32[(k + 1) & (NC - 1)];
    const uint32_t hw = __builtin_amdgcn_alignbit(hk, hprev, 31);     // (h << 1) | (prev >> 31): x torus seam at k = 0
    const uint32_t he = __builtin_amdgcn_alignbit(hnext, hk, 1);
    const uint32_t g0 = bop3<kXor3>(hw, hk, he), g1 = bop3<kMaj>(hw, hk, he);
    if (lane < 2 * NC) l.hsum[row][k] = make_uint2(g0, g1);
}

// REC (kernels with the 64-column window), `with_rec` (wave-uniform): the grid's 512-byte action record (the 64 row words
// of the window) is requested behind the strip's loads -- one more coalesced access, 8 bytes per lane -- and laid down in
// l.act before `on_image` runs: the record travels WITH its window strip.  (Requesting it one strip early -- into two
// registers, or as an LDS-DMA behind the first strip of the unit -- measured no better: 45.9 / 47.4 against 45.3 us at
// N=4096 512x512.)
// `on_image` runs between the two halves: the strip lies in LDS in memory layout, no row is in a register yet -- the
// place where the window is toggled (toggle_strip_image).
struct NoStripHook { __device__ __forceinline__ void operator()() const {} };
// HALREG false: the two bordering rows stay in LDS (l.halo) -- the compute phase reads them chunk by chunk.
template <int WW, bool NT, bool REC = false, bool HALREG = true, class Hook = NoStripHook>
__device__ __forceinline__ void load_strip(StripLds<WW> &l, const uint64_t *__restrict__ gstrip,
                                           const uint64_t *__restrict__ grow_top, const uint64_t *__restrict__ grow_bot,
                                           bool multi, int lane, uint64_t (&c)[WW], uint64_t (&hal)[WW],
                                           const uint64_t *__restrict__ act_rec = nullptr, bool with_rec = false,
                                           Hook on_image = Hook()) {
    constexpr int Q = WW / 2;                                // 16-byte quads per row = accesses per strip
    const uint4 *g4 = reinterpret_cast<const uint4 *>(gstrip);
    uint4 v[Q];
#pragma unroll
    for (int j = 0; j < Q; ++j) v[j] = ld16s<NT>(g4 + j * 64 + lane);
    // halo: lanes 0..Q-1 the row above, Q..2Q-1 the row below (default cache policy: the neighbouring strip owns them);
    // lanes >= 2Q repeat lane Q's access -- load AND store: with the store under `if (lane < 2Q)` the compiler sinks the
    // load into the branch, behind the waits for the strip's loads: a second dependent round trip per strip (rounds 2-3)
    const int hq = lane < Q ? lane : (lane < 2 * Q ? lane - Q : 0);
    const uint4 *h4 = reinterpret_cast<const uint4 *>(lane < Q ? grow_top : grow_bot) + hq;
    const uint4 hv = *h4;
    uint2 av = make_uint2(0u, 0u);
    if (REC && with_rec) av = reinterpret_cast<const uint2 *>(act_rec)[lane];
#pragma unroll
    for (int j = 0; j < Q; ++j) l.img[j * 64 + lane] = v[j];
    l.halo[lane < Q ? 0 : 1][hq] = hv;
    if (REC && with_rec) reinterpret_cast<uint2 *>(l.act)[lane] = av;
    on_image();
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const uint4 r = l.img[lane * Q + q];
        c[2 * q] = ((uint64_t)r.y << 32) | r.x;
        c[2 * q + 1] = ((uint64_t)r.w << 32) | r.z;
    }
    if (HALREG) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const uint4 r = l.halo[lane == 63 ? 1 : 0][q];
            const bool use = multi && (lane == 0 || lane == 63);
            hal[2 * q] = use ? (((uint64_t)r.y << 32) | r.x) : 0ull;
            hal[2 * q + 1] = use ? (((uint64_t)r.w << 32) | r.z) : 0ull;
        }
    }
}

// The 64-column window applied to the strip IMAGE (and to the images of its two bordering rows) in LDS: the window's
// 64 columns are dwords WW - 1 and WW of a row (bit offset 32 (WW - 1)), so a lane XORs the two dwords of its action row
// into its image row -- two ds_xor_b32 without a result, no register beyond the record words.  (In registers the toggle
// held both 64-bit row arrays and its selects live at once: 92 VGPRs, 5 waves per SIMD instead of 8.)
// One wavefront per workgroup: the LDS pipeline keeps its accesses in order, the row reads behind this see the toggle.
template <int WW>
__device__ __forceinline__ void toggle_strip_image(StripLds<WW> &l, int lane, bool in, uint2 wr, bool hin, uint2 wh) {
    uint32_t *img32 = reinterpret_cast<uint32_t *>(l.img);
    if (in) {
        __hip_atomic_fetch_xor(img32 + lane * 2 * WW + WW - 1, wr.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_xor(img32 + lane * 2 * WW + WW, wr.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    if (hin) {
        uint32_t *h32 = reinterpret_cast<uint32_t *>(l.halo[lane == 63 ? 1 : 0]);
        __hip_atomic_fetch_xor(h32 + WW - 1, wh.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_xor(h32 + WW, wh.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
}

template <int WW, bool NT>
__device__ __forceinline__ void store_strip(StripLds<WW> &l, uint64_t *__restrict__ gstrip, int lane,
                                            const uint64_t (&nx)[WW]) {
    constexpr int Q = WW / 2;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        uint4 r;
        r.x = (uint32_t)nx[2 * q]; r.y = (uint32_t)(nx[2 * q] >> 32);
        r.z = (uint32_t)nx[2 * q + 1]; r.w = (uint32_t)(nx[2 * q + 1] >> 32);
        l.img[lane * Q + q] = r;
    }
    uint4 *g4 = reinterpret_cast<uint4 *>(gstrip);
#pragma unroll
    for (int j = 0; j < Q; ++j) st16s<NT>(g4 + j * 64 + lane, l.img[j * 64 + lane]);
}

// the fused device-side random agent (ACT == 1, RAND): the 64 toggles of an action row are drawn in
// the kernel with the counter layout of philox.hpp (element index of the (N,1,64,64) action tensor),
// bit-identical to packing carle_bernoulli_f32's output
struct RandB1 {
    float p = 0.0f;
    uint64_t seed = 0, offset = 0;
};
__device__ __forceinline__ uint64_t drawn_row64(int64_t g, int ay, const RandB1 &ra) {
    const uint64_t q0 = (uint64_t)((g * 64 + ay) * 64) >> 2;        // first quad of action row ay of grid g
    uint64_t wrd = 0ull;
#pragma unroll 4
    for (int c = 0; c < 16; ++c) wrd |= (uint64_t)bernoulli_nibble(q0 + c, ra.seed, ra.offset, ra.p) << (4 * c);
    return wrd;
}

// ENV instantiation (CARLE_STEP_DEFERRED): the whole CARLE.step of the packed environment in this one launch (as
// ca_step_64_kernel<..., ENV>, ca_step_cells.hip).  The master reset is decided per work unit: a unit walks its window
// strip first and learns there whether its share of the grid's action window is all ones (never, under an agent) -- only
// then does it ask batch_action_all_ones_raw(); a reset unit writes zero strips.  Block 0 does the bookkeeping.
struct B1EnvArgs {
    float *reward = nullptr;       // N floats, zeroed (nullable)
    int publish = 1;               // 0: the caller stored the action flags of `slot` beforehand (fp32 actions packed by
                                   //    carle_pack_f32_b1, which publishes the exact fp32 predicates)
    int merge_moments = 0;
    // cached strip mode (LDS == 1): grids with an index >= nt_from travel with the streaming hint (loads and stores), the
    // others with the default policy -- see b1_resident_grids()
    int64_t nt_from = INT64_MAX;
};

// ACT: 0 = no action, 1 = 64-column window (toggle_row64), 2 = any packed window (toggle_row)
template <int WW, int ACT, int RULE_MODE, bool RAND = false, int LDS = 0, bool ENV = false>      // LDS: 0 row-per-lane, 1 StripLds cached, 2 StripLds streaming
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_sgpr(90))) void ca_step_b1_kernel(
    const uint64_t *__restrict__ u_in, const uint64_t *__restrict__ act, uint64_t *__restrict__ u_out,
    int64_t n, int ah, int aw, uint32_t rule, int32_t *__restrict__ status, int slot,
    int strip_first, int strip_count, RandB1 ra = RandB1(), int dbg = 0, B1EnvArgs ea = B1EnvArgs()) {
    static_assert(!ENV || (ACT != 2 && !RAND), "one-launch step: no action or the 64-column window from memory");
    // the launch covers strips [strip_first, strip_first + strip_count) (mod WW) of every grid:
    // wide windows: the host launches the strips that can meet the action window with ACT = 2 and
    // the others with the leaner ACT = 0 instantiation (half the registers)
    constexpr int H = 64 * WW;
    constexpr bool MULTI = WW > 1;                          // more than one 64-row strip per grid
    constexpr bool LDS_T = MULTI && LDS != 0;               // coalesced strip traffic through LDS (see StripLds)
    // (the streaming mode hints loads AND stores; CARLE_B1_PROBE_NT_LD / _ST -- probe builds only, tools/ab_libs.sh -- put
    // the hint on one side of the cached mode: which of the two the Infinity-Cache reuse of step t's output hangs on)
#ifdef CARLE_B1_PROBE_NT_LD
    constexpr bool LD_NT = true;
#else
    constexpr bool LD_NT = LDS == 2;
#endif
#ifdef CARLE_B1_PROBE_NT_ST
    constexpr bool ST_NT = true;
#else
    constexpr bool ST_NT = LDS == 2;
#endif
    // the strip's two bordering rows stay in LDS and the compute phase reads them one 32-bit chunk at a time (a rolling
    // window of three): 16 registers that only lanes 0 and 63 had a use for (window kernels: 76 -> <= 64 VGPRs, eight
    // waves per SIMD instead of six).  Kernels that toggle the bordering rows in registers keep them there.
    constexpr bool HROLL = LDS_T && (ACT == 0 || (ACT == 1 && !RAND));
    __shared__ StripLds<LDS_T ? WW : 2> lds;
    const int lane = threadIdx.x;
    // Persistent waves: the launch has at most 32 waves per CU and every wave walks a list of work units
    // with stride gridDim.x.  One-shot waves spent more time being launched and retired than running
    // (profiles/r02_b1_stamps.txt: 5 us of instructions per strip in 13 us of wave slot).
    // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 shares an XCD,
    // and gridDim.x is a multiple of 8), so XCD x takes the grids g = x (mod 8): the strips of a grid
    // share an L2 (halo rows are L2 hits).  (Placement is not guaranteed: any other dealing only changes
    // the speed.)  Partial launches (wide windows: strip_count < WW) walk single strips.
    // Launches with the 64-column action window (ACT == 1) on multi-strip grids walk HALF-GRID units -- WW / 2
    // consecutive strips per wave -- instead of single strips.  What the window costs is not its toggle but its
    // 4096 small, cold action records: with ONE shared, L2-hot record the 512x512 step takes 48.5 us against 47.5 us
    // without a window, with one record per grid 55-57 us.  The window is centred, so both halves of a grid hold
    // one window strip: every unit costs the same (a wave that walks single strips with a fixed stride draws a
    // binomial number of the dearer window strips), a unit walks its window strip FIRST -- the records are
    // cheapest to fetch at the very start of the launch, before the state streams saturate the memory system --
    // and fetches its window words before anything else.  Same-box A/B (profiles/r02_kbench_b1_units.txt):
    // 52.3 us against 55.8 us for the single-strip walk (0.645 against 0.606 of 8 TB/s).  What did not help or
    // hurt: consecutive instead of interleaved grids per XCD, L2 prefetch of the records by a few waves, the nt
    // hint on the state traffic (2.3x slower), the window strip later in the unit, peeling the first strip out
    // of the loop.  Launches without a window keep the single-strip walk (47.9 us; half-grid units: 48.7 us).
    const bool reset_all = ENV && ACT == 1 && !ea.publish && status[CARLE_ST_NOT_ALL_ONES(slot)] == 0;
    if (ENV && ea.reward)
        for (int64_t i = (int64_t)blockIdx.x * 64 + lane; i < n; i += (int64_t)gridDim.x * 64) ea.reward[i] = 0.0f;   // env.py:238
    const int64_t n8 = ((n + 7) / 8) * 8;
    const bool halves = MULTI && strip_count == WW && ACT == 1 && !(dbg & 8);    // dbg bit 3: the single-strip walk
    const int per_unit = halves ? WW / 2 : 1;
    const int64_t total = halves ? n8 * 2 : n8 * strip_count;
    constexpr int HS = WW / 2 > 0 ? WW / 2 : 1;
  for (int64_t unit = blockIdx.x; unit < total; unit += gridDim.x) {
    const int64_t j = unit >> 3;
    const int64_t g = halves ? 8 * (j >> 1) + (unit & 7) : 8 * (j / strip_count) + (unit & 7);
    if (g >= n) continue;
    const bool unit_nt = ENV && LDS == 1 && g >= ea.nt_from;     // (wave-uniform) this grid is not kept in the Infinity Cache
    // the window words of the unit's window strip (HS >= 2: exactly one strip of a unit meets the 64-row
    // window, the one at the middle of the grid) and of that strip's bordering row
    uint32_t pa_lo = 0u, pa_hi = 0u, ph_lo = 0u, ph_hi = 0u;
    bool unit_reset = reset_all;                             // ENV: this unit's strips are zeroed (master reset)
    // LDS strip modes: the action record travels with the window strip through LDS (load_strip<.., ACTREC>) and the
    // toggle happens there (toggle_strip_image) -- no earlier fetch, no toggle in registers
    constexpr bool act_lds = LDS_T && ACT == 1 && !RAND;
    const bool preload = ACT == 1 && !RAND && halves && HS >= 2 && !(dbg & 1) && !act_lds;
    if (preload) {
        const int r0w = ((j & 1) ? WW - HS : HS - 1) * 64, oyw = (H - ah) >> 1;
        const int rw = r0w + lane;
        const int hrw = lane == 0 ? (r0w + H - 1) % H : (lane == 63 ? (r0w + 64) % H : -1);
        const bool inw = rw >= oyw && rw < oyw + ah, hinw = hrw >= oyw && hrw < oyw + ah;
        const uint32_t *a32 = reinterpret_cast<const uint32_t *>(act + g * (int64_t)ah + (inw ? rw - oyw : 0));
        const uint32_t *h32 = reinterpret_cast<const uint32_t *>(act + g * (int64_t)ah + (hinw ? hrw - oyw : 0));
        pa_lo = a32[0]; pa_hi = a32[1]; ph_lo = h32[0]; ph_hi = h32[1];
        // ... and waited for: the records arrive while the memory system is still idle (measured faster than
        // letting them travel beside the first strip's state loads: 52.3 against 53.8-57.9 us)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(pa_lo), "+v"(pa_hi), "+v"(ph_lo), "+v"(ph_hi));
    }
    // ENV, master reset (env.py:208-216): the window strips of the unit note whether their share of the action window
    // holds anything but ones; a unit that saw only ones (never, under an agent) asks the batch BEHIND its last strip and
    // overwrites its strips with zeros if the answer is yes -- no call while rows are in registers, no fetch ahead of
    // the strips, any strip order
    bool unit_notone = false;
    // single-strip units (dbg bit 3) of the one-launch step: a strip WITHOUT rows under the window cannot see the action; it
    // looks at a sample instead -- the first row word of its grid's record (uniform: a scalar load, needed only behind the
    // strip) -- and knows "no reset" unless that word is all ones, as the wavefront-per-strip fp32 kernel does
    // (ca_step_wide_kernel<..., ENV>); only then does it ask the batch
    uint64_t sample = 0ull;
    if (ENV && ACT == 1 && !halves && ea.publish) sample = act[g * (int64_t)ah];
    // one strip: `first` = the unit's first strip (its window strip when the window words were fetched above)
    auto strip = [&](const int sub, const bool first) __attribute__((always_inline)) {
    const int64_t wid = halves ? unit * per_unit + sub : unit;          // stamp index
    // strip order inside a half: from the middle of the grid (the window strip) outwards; the lower half
    // mirrors the upper one
    const int li = HS - 1 - sub;
    const int r0 = halves ? ((j & 1) ? WW - 1 - li : li) * 64
                          : ((strip_first + (int)(j % strip_count)) % WW) * 64;
    (void)wid;
    B1_STAMP(0);
    const int r = r0 + lane;
    const uint64_t *gin = u_in + g * (int64_t)H * WW;
    const int rt = (r0 + H - 1) % H, rb = (r0 + 64) % H;   // rows bordering the strip (y torus)
    const int hr = lane == 0 ? rt : (lane == 63 ? rb : -1);

    uint64_t c[WW], hal[WW];
    // does the strip (with its bordering rows) meet the action window at all?  (wave-uniform)
    const int oy_ = (H - ah) >> 1;
    const bool touches_ = ACT == 1 && ((r0 < oy_ + ah && r0 + 64 > oy_) ||
                                       (MULTI && ((rt >= oy_ && rt < oy_ + ah) || (rb >= oy_ && rb < oy_ + ah))));
    if (ENV && ACT == 1 && !halves && ea.publish && !(r0 < oy_ + ah && r0 + 64 > oy_))
        unit_notone |= sample != ~0ull;                      // no row of this strip is under the window: the sample decides
    if constexpr (LDS_T) {
        if constexpr (act_lds) {
            // a strip under the window: predicates and the toggle happen on the LDS image, before any row occupies a
            // register.  The record arrives with this strip (StripLds::act holds its 64 row words: the window is square,
            // check_dims, so AW == 64 means AH == 64).
            auto on_image = [&]() __attribute__((always_inline)) {
                if (!touches_) return;
                const bool in = r >= oy_ && r < oy_ + ah, hin = hr >= oy_ && hr < oy_ + ah;
                const uint2 *a2 = reinterpret_cast<const uint2 *>(lds.act);
                const uint2 wr = a2[in ? r - oy_ : 0], wh = a2[hin ? hr - oy_ : 0];
                if (status != nullptr && (!ENV || ea.publish)) {
                    const bool any = __ballot(in && (wr.x | wr.y) != 0u) != 0ull;
                    const bool notone = __ballot(in && (wr.x & wr.y) != 0xFFFFFFFFu) != 0ull;
                    if (lane == 0) {
                        if (any) status[CARLE_ST_ANY_NONZERO(slot)] = 1;
                        if (notone) status[CARLE_ST_NOT_ALL_ONES(slot)] = 1;
                    }
                    unit_notone |= notone;
                }
                toggle_strip_image<WW>(lds, lane, in, wr, hin, wh);
            };
            if (unit_nt)
                load_strip<WW, true, true, !HROLL>(lds, gin + (int64_t)r0 * WW, gin + (int64_t)rt * WW, gin + (int64_t)rb * WW, MULTI, lane,
                                                   c, hal, act + g * (int64_t)ah, touches_, on_image);
            else
                load_strip<WW, LD_NT, true, !HROLL>(lds, gin + (int64_t)r0 * WW, gin + (int64_t)rt * WW, gin + (int64_t)rb * WW, MULTI, lane,
                                                    c, hal, act + g * (int64_t)ah, touches_, on_image);
        } else {
            if (unit_nt)
                load_strip<WW, true, false, !HROLL>(lds, gin + (int64_t)r0 * WW, gin + (int64_t)rt * WW, gin + (int64_t)rb * WW, MULTI, lane, c, hal);
            else
                load_strip<WW, LD_NT, false, !HROLL>(lds, gin + (int64_t)r0 * WW, gin + (int64_t)rt * WW, gin + (int64_t)rb * WW, MULTI, lane, c, hal);
        }
        if constexpr (HROLL) strip_halo_sums<(LDS_T ? WW : 2)>(lds, lane);     // behind the window toggle (LDS keeps a wave's order)
    } else {
        load_row<WW>(gin + (int64_t)r * WW, c);
#pragma unroll
        for (int j = 0; j < WW; ++j) hal[j] = 0ull;
        if (MULTI && hr >= 0) load_row<WW>(gin + (int64_t)hr * WW, hal);
    }

    if (ACT == 1) {
        const int oy = (H - ah) >> 1;
        const bool touches = (r0 < oy + ah && r0 + 64 > oy) ||
                             (MULTI && ((rt >= oy && rt < oy + ah) || (rb >= oy && rb < oy + ah)));
        if (touches && !act_lds) {              // wave-uniform; the LDS modes toggle on the strip image (load_strip's hook)
            const uint64_t *ga = act + g * (int64_t)ah;
            uint32_t s_or = 0u, s_and = 0xFFFFFFFFu, d0 = 0u, d1 = 0xFFFFFFFFu;
            const bool in = r >= oy && r < oy + ah;
            if (RAND) {
                // only the lanes under the window draw (a halo row draws the same bits as the strip
                // that owns the row: the toggles are a pure function of the element index)
                uint64_t wrd = 0ull, hwrd = 0ull;
                if (in) wrd = drawn_row64(g, r - oy, ra);
                toggle_words64<WW>(c, (uint32_t)wrd, (uint32_t)(wrd >> 32), in, s_or, s_and);
                if (MULTI) {
                    const bool hin = hr >= oy && hr < oy + ah;
                    if (hin) hwrd = drawn_row64(g, hr - oy, ra);
                    toggle_words64<WW>(hal, (uint32_t)hwrd, (uint32_t)(hwrd >> 32), hin, d0, d1);
                }
            } else if (preload && first) {
                toggle_words64<WW>(c, pa_lo, pa_hi, in, s_or, s_and);
                const bool hin = hr >= oy && hr < oy + ah;
                toggle_words64<WW>(hal, ph_lo, ph_hi, hin, d0, d1);
            } else {
            toggle_row64<WW>(c, ga + (in ? r - oy : 0), in, s_or, s_and);
            if (MULTI) {
                const bool hin = hr >= oy && hr < oy + ah;
                toggle_row64<WW>(hal, ga + (hin ? hr - oy : 0), hin, d0, d1);
            }
            }
            if (status != nullptr && (!ENV || ea.publish)) {
                const bool any = __ballot(s_or != 0u) != 0ull;
                const bool notone = __ballot(in && s_and != 0xFFFFFFFFu) != 0ull;
                if (lane == 0) {
                    if (any) status[CARLE_ST_ANY_NONZERO(slot)] = 1;
                    if (notone) status[CARLE_ST_NOT_ALL_ONES(slot)] = 1;
                }
                unit_notone |= notone;
            }
        }
    }
    if (ACT == 2) {
        __builtin_amdgcn_sched_barrier(0);      // keep the toggle phase's registers out of the
        const int oy = (H - ah) >> 1, aww = aw >> 6;    // compute phase (occupancy: <= 64 VGPRs)
        // wave-uniform: does the strip or one of its halo rows meet the action window at all?
        const bool touches = (r0 < oy + ah && r0 + 64 > oy) ||
                             (MULTI && ((rt >= oy && rt < oy + ah) || (rb >= oy && rb < oy + ah)));
        if (touches) {
            const uint64_t *ga = act + g * (int64_t)ah * aww;
            uint64_t s_or = 0ull, s_and = ~0ull, d0 = 0ull, d1 = ~0ull;
            const bool in = r >= oy && r < oy + ah;
            toggle_row<WW>(c, ga + (int64_t)(in ? r - oy : 0) * aww, aww, in, s_or, s_and);
            __builtin_amdgcn_sched_barrier(0);
            if (MULTI) {
                const bool hin = hr >= oy && hr < oy + ah;
                toggle_row<WW>(hal, ga + (int64_t)(hin ? hr - oy : 0) * aww, aww, hin, d0, d1);
            }
            if (status != nullptr) {
                const bool any = __ballot(s_or != 0ull) != 0ull;
                const bool notone = __ballot(in && s_and != ~0ull) != 0ull;
                if (lane == 0) {
                    if (any) status[CARLE_ST_ANY_NONZERO(slot)] = 1;
                    if (notone) status[CARLE_ST_NOT_ALL_ONES(slot)] = 1;
                }
            }
        }
    }

    if (ACT == 2) __builtin_amdgcn_sched_barrier(0);
#ifdef CARLE_B1_STAMPS
    { uint64_t keep = c[0]; asm volatile("s_waitcnt vmcnt(0)" : "+v"(keep)); c[0] = keep; }
#endif
    B1_STAMP(1);

    // ---- compute phase on 32-bit chunks (NC = 2*WW per row): every operation below is a single
    // full-rate VALU instruction -- v_alignbit_b32 for the 1-bit funnel shifts across chunk
    // borders (the wrap j-1 / j+1 is the x torus seam), v_bitop3_b32 for the adders and the rule,
    // v_mov_b32_dpp for the rows above / below.
    constexpr int NC = 2 * WW;
    uint32_t cc[NC], hc[NC];
#pragma unroll
    for (int j = 0; j < WW; ++j) {
        cc[2 * j] = (uint32_t)c[j]; cc[2 * j + 1] = (uint32_t)(c[j] >> 32);
        hc[2 * j] = HROLL ? 0u : (uint32_t)hal[j]; hc[2 * j + 1] = HROLL ? 0u : (uint32_t)(hal[j] >> 32);
    }
    // HROLL: the triple sums of the lane's bordering row (lane 63: the row below, every other lane the row above -- only
    // lanes 0 and 63 have a use for theirs) come from l.hsum, one 8-byte pair per chunk, read one chunk ahead of its use
    const uint2 *hsum = &lds.hsum[0][0] + (lane == 63 ? NC : 0);
    uint2 gs = make_uint2(0u, 0u), gs_next = make_uint2(0u, 0u);
    uint32_t hidx = 0;
    if constexpr (HROLL) { gs = hsum[0]; gs_next = hsum[1 % NC]; }
    uint32_t nx32[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const uint32_t prev = cc[(k + NC - 1) % NC], next = cc[(k + 1) % NC];
        const uint32_t wst = __builtin_amdgcn_alignbit(cc[k], prev, 31);     // (c << 1) | (prev >> 31)
        const uint32_t est = __builtin_amdgcn_alignbit(next, cc[k], 1);      // (c >> 1) | (next << 31)
        uint32_t h0, h1, p0, p1;
        row_sums(wst, cc[k], est, h0, h1, p0, p1);
        uint32_t a0, a1, b0, b1;
        if (MULTI) {
            uint32_t g0, g1;
            if constexpr (HROLL) {
                g0 = gs.x; g1 = gs.y;
            } else {
                const uint32_t hprev = hc[(k + NC - 1) % NC], hk = hc[k], hnext = hc[(k + 1) % NC];
                const uint32_t hw = __builtin_amdgcn_alignbit(hk, hprev, 31);
                const uint32_t he = __builtin_amdgcn_alignbit(hnext, hk, 1);
                g0 = bop3<kXor3>(hw, hk, he); g1 = bop3<kMaj>(hw, hk, he);
            }
            // lanes with no source keep `old` = the sums of the halo row they loaded themselves
            a0 = __builtin_amdgcn_update_dpp(g0, h0, 0x138, 0xF, 0xF, false);   // wave_shr:1, row above
            a1 = __builtin_amdgcn_update_dpp(g1, h1, 0x138, 0xF, 0xF, false);
            b0 = __builtin_amdgcn_update_dpp(g0, h0, 0x130, 0xF, 0xF, false);   // wave_shl:1, row below
            b1 = __builtin_amdgcn_update_dpp(g1, h1, 0x130, 0xF, 0xF, false);
        } else {
            a0 = __builtin_amdgcn_update_dpp(0u, h0, 0x13C, 0xF, 0xF, false);   // wave_ror:1 = y seam
            a1 = __builtin_amdgcn_update_dpp(0u, h1, 0x13C, 0xF, 0xF, false);
            b0 = __builtin_amdgcn_update_dpp(0u, h0, 0x134, 0xF, 0xF, false);   // wave_rol:1
            b1 = __builtin_amdgcn_update_dpp(0u, h1, 0x134, 0xF, 0xF, false);
        }
        nx32[k] = next_state<RULE_MODE>(neighbour_count(a0, a1, b0, b1, p0, p1), cc[k], rule);
        // chunk after chunk: left alone, the scheduler of the window kernels computes the row sums of ALL chunks first
        // (its maximum-ILP order once some other region has cost it the occupancy it aimed for): 145 VGPRs instead of 65.
        // An empty asm ties chunk k + 1's centre word (and the index of the next bordering-row read) to chunk k's result.
        if ((HROLL || (ACT == 1 && !RAND)) && k + 1 < NC) {
            if constexpr (HROLL) {
                asm volatile("" : "+v"(cc[k + 1]), "+v"(hidx) : "v"(nx32[k]));
                gs = gs_next;
                gs_next = hsum[hidx + (k + 2) % NC];
            } else {
                asm volatile("" : "+v"(cc[k + 1]) : "v"(nx32[k]));
            }
        }
    }
    uint64_t nx[WW];
#pragma unroll
    for (int j = 0; j < WW; ++j) nx[j] = ((uint64_t)nx32[2 * j + 1] << 32) | nx32[2 * j];
    if (ENV && ACT == 1 && unit_reset) {                     // master reset, env.py:208-216
#pragma unroll
        for (int j = 0; j < WW; ++j) nx[j] = 0ull;
    }
    B1_STAMP(2);
    if constexpr (LDS_T) {
        if (unit_nt) store_strip<WW, true>(lds, u_out + (g * (int64_t)H + r0) * WW, lane, nx);
        else store_strip<WW, ST_NT>(lds, u_out + (g * (int64_t)H + r0) * WW, lane, nx);
    } else store_row<WW>(u_out + (g * (int64_t)H + r) * WW, nx);
    B1_STAMP(3);
    };
    // dbg bit 2: the units start at different strips of their half -- at any moment the wavefronts of the launch read
    // all strip positions, not all the same 4 KB of every 16 KB (the window strip comes when it comes)
    const int rot = (act_lds && (dbg & 4)) ? (int)((g >> 3) & (HS - 1)) : 0;
#pragma nounroll
    for (int s_ = 0; s_ < per_unit; ++s_) {
        const int sub = (s_ + rot) & (per_unit - 1);
        strip(sub, sub == 0);
    }
    if (ENV && ACT == 1 && ea.publish && __builtin_expect(!unit_notone && !unit_reset, 0)) {
        // this unit's share of the window was all ones: ask the batch; a master reset zeroes the unit's strips
        if (batch_action_all_ones_raw(act, n, (int64_t)ah * 8 / 16, 0xFFFFFFFFu, status, slot)) {
            for (int s_ = 0; s_ < per_unit; ++s_) {
                const int li = HS - 1 - s_;
                const int r0 = halves ? ((j & 1) ? WW - 1 - li : li) * 64 : ((strip_first + (int)(j % strip_count)) % WW) * 64;
                uint64_t *dst = u_out + (g * (int64_t)H + r0 + lane) * WW;
#pragma unroll
                for (int q = 0; q < WW; ++q) dst[q] = 0ull;
            }
        }
    }
  }
    if (ENV && blockIdx.x == 0)
        env_step_bookkeeping(status, slot, n, ACT != 0, ea.merge_moments && ea.reward != nullptr);
}

// ---------------------------------------------------------------------------------------------
// k generations without actions in ONE launch (= k calls of CARLE.step with an all-zero action,
// carle/env.py:188-242): the grid stays in registers between generations.
// ---------------------------------------------------------------------------------------------
// One workgroup of WW waves per grid, lane = row as above.  Per generation the only traffic is, for
// grids taller than 64 rows, the horizontal sums of each wave's first and last row through a
// double-buffered LDS array (the rows bordering the next wave's strip; the wrap over the WW waves
// is the y torus seam) and ONE workgroup barrier.  HBM traffic: 0.25 B per cell per k generations.
template <int WW, int RULE_MODE>
__global__ __launch_bounds__(64 * WW) void ca_multistep_b1_kernel(
    const uint64_t *__restrict__ u_in, uint64_t *__restrict__ u_out, int64_t n, uint32_t rule, int k,
    int32_t *__restrict__ status) {
    constexpr int H = 64 * WW, NC = 2 * WW;
    constexpr bool MULTI = WW > 1;
    __shared__ uint32_t s_edge[2][MULTI ? WW : 1][2][NC][2];   // [buffer][wave][first/last row][chunk][h0/h1]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t g = blockIdx.x;
    const int r = 64 * w + lane;

    uint64_t c[WW];
    load_row<WW>(u_in + (g * (int64_t)H + r) * WW, c);
    uint32_t cc[NC];
#pragma unroll
    for (int j = 0; j < WW; ++j) { cc[2 * j] = (uint32_t)c[j]; cc[2 * j + 1] = (uint32_t)(c[j] >> 32); }

    for (int t = 0; t < k; ++t) {
        uint32_t h0[NC], h1[NC], p0[NC], p1[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const uint32_t prev = cc[(j + NC - 1) % NC], next = cc[(j + 1) % NC];
            const uint32_t wst = __builtin_amdgcn_alignbit(cc[j], prev, 31);
            const uint32_t est = __builtin_amdgcn_alignbit(next, cc[j], 1);
            row_sums(wst, cc[j], est, h0[j], h1[j], p0[j], p1[j]);
        }
        uint32_t g0[NC], g1[NC];
        if (MULTI) {
            const int buf = t & 1;
            if (lane == 0 || lane == 63) {
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    s_edge[buf][w][lane == 63][j][0] = h0[j];
                    s_edge[buf][w][lane == 63][j][1] = h1[j];
                }
            }
            __syncthreads();
            // lane 0 needs the LAST row of the wave above, lane 63 the FIRST row of the wave below
            const int src_w = lane == 0 ? (w + WW - 1) % WW : (w + 1) % WW;
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                g0[j] = s_edge[buf][src_w][lane == 0][j][0];
                g1[j] = s_edge[buf][src_w][lane == 0][j][1];
            }
        }
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            uint32_t a0, a1, b0, b1;
            if (MULTI) {
                a0 = __builtin_amdgcn_update_dpp(g0[j], h0[j], 0x138, 0xF, 0xF, false);   // wave_shr:1
                a1 = __builtin_amdgcn_update_dpp(g1[j], h1[j], 0x138, 0xF, 0xF, false);
                b0 = __builtin_amdgcn_update_dpp(g0[j], h0[j], 0x130, 0xF, 0xF, false);   // wave_shl:1
                b1 = __builtin_amdgcn_update_dpp(g1[j], h1[j], 0x130, 0xF, 0xF, false);
            } else {
                a0 = __builtin_amdgcn_update_dpp(0u, h0[j], 0x13C, 0xF, 0xF, false);      // wave_ror:1
                a1 = __builtin_amdgcn_update_dpp(0u, h1[j], 0x13C, 0xF, 0xF, false);
                b0 = __builtin_amdgcn_update_dpp(0u, h0[j], 0x134, 0xF, 0xF, false);      // wave_rol:1
                b1 = __builtin_amdgcn_update_dpp(0u, h1[j], 0x134, 0xF, 0xF, false);
            }
            cc[j] = next_state<RULE_MODE>(neighbour_count(a0, a1, b0, b1, p0[j], p1[j]), cc[j], rule);
        }
    }
#pragma unroll
    for (int j = 0; j < WW; ++j) c[j] = ((uint64_t)cc[2 * j + 1] << 32) | cc[2 * j];
    store_row<WW>(u_out + (g * (int64_t)H + r) * WW, c);

    // counters of k action-free steps (env.py:199-200, 230): step_number += k, steps_since_action += k
    if (status != nullptr && g == 0 && threadIdx.x == 0) {
        settle_status_in_place(status);                   // a deferred one-launch step before this call
        int64_t *cnt = reinterpret_cast<int64_t *>(status);
        cnt[CARLE_ST_STEP_NUMBER / 2] += k;
        cnt[CARLE_ST_STEPS_SINCE_ACTION / 2] += k;
        if (k > 0) status[CARLE_ST_LAST_RESET] = 0;
    }
}

template <int WW>
static int launch_multistep(const uint64_t *u_in, uint64_t *u_out, int64_t n, uint32_t rule, int k,
                            int32_t *status, hipStream_t s) {
    if (n > 0x7FFFFFFFll) return fail_arg(CARLE_E_DIMS, "carle_ca_multistep_b1: batch too large");
    const dim3 grid((unsigned)n), block(64 * WW);
    if (rule == kLifeMask)
        hipLaunchKernelGGL((ca_multistep_b1_kernel<WW, kRuleLife>), grid, block, 0, s, u_in, u_out, n, rule, k, status);
    else if (rule == kHighLifeMask)
        hipLaunchKernelGGL((ca_multistep_b1_kernel<WW, kRuleHighLife>), grid, block, 0, s, u_in, u_out, n, rule, k, status);
    else
        hipLaunchKernelGGL((ca_multistep_b1_kernel<WW, kRuleGeneric>), grid, block, 0, s, u_in, u_out, n, rule, k, status);
    return check_launch("carle_ca_multistep_b1");
}

// ---------------------------------------------------------------------------------------------
// pack fp32 -> b1 (also publishes the action predicates of the fp32 source), unpack b1 -> fp32
// ---------------------------------------------------------------------------------------------
// one wave packs 64 words: iteration i loads 64 consecutive floats (one word), the ballot of
// "non-zero" is the packed word, kept by lane i; then one coalesced 512-B store.
__global__ __launch_bounds__(64) void pack_f32_b1_kernel(const float *__restrict__ src,
                                                         uint64_t *__restrict__ dst, int64_t words,
                                                         int32_t *__restrict__ status, int slot) {
    const int lane = threadIdx.x;
    const int64_t w0 = (int64_t)blockIdx.x * 64;
    uint64_t mine = 0ull;
    uint32_t all_and = 0xFFFFFFFFu, all_or = 0u;
    const int cnt = (int)((words - w0) < 64 ? (words - w0) : 64);
    for (int i = 0; i < cnt; ++i) {
        const uint32_t v = __float_as_uint(src[(w0 + i) * 64 + lane]);
        all_and &= v;
        all_or |= v;
        const uint64_t b = __ballot((v & 0x7FFFFFFFu) != 0u);
        if (lane == i) mine = b;
    }
    if (lane < cnt) dst[w0 + lane] = mine;
    if (status != nullptr) {
        const bool any = __ballot((all_or & 0x7FFFFFFFu) != 0u) != 0ull;
        const bool notone = __ballot(all_and != 0x3F800000u || all_or != 0x3F800000u) != 0ull;
        if (lane == 0) {
            if (any) status[CARLE_ST_ANY_NONZERO(slot)] = 1;
            if (notone) status[CARLE_ST_NOT_ALL_ONES(slot)] = 1;
        }
    }
}

// one thread expands 4 cells (one nibble) into a float4
__global__ __launch_bounds__(256) void unpack_b1_f32_kernel(const uint64_t *__restrict__ src,
                                                            float4 *__restrict__ dst, int64_t quads) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= quads) return;
    const uint64_t wv = src[i >> 4];
    const uint32_t nib = (uint32_t)(wv >> (4 * (i & 15)));
    float4 o;
    o.x = (nib & 1u) ? 1.0f : 0.0f;
    o.y = (nib & 2u) ? 1.0f : 0.0f;
    o.z = (nib & 4u) ? 1.0f : 0.0f;
    o.w = (nib & 8u) ? 1.0f : 0.0f;
    dst[i] = o;
}

// How many grids of a packed state array stay cacheable (cached strip mode of the one-launch step): round 5.  The Infinity
// Cache holds 256 MiB; step t + 1 re-reads what step t wrote, so with input + output at or just under 256 MiB (BASELINE
// configs[4]: 2 x 128 MiB) the re-read hits only if NOTHING else displaces a line -- and whether that holds hung on the
// physical pages the two allocations got: per pair of buffers, in one process, 42.5 us or 44.4-46 us, the same virtual
// addresses on either level after a re-allocation, every pair equal (36.5 us) at 2 x 112 MiB, two levels again at 2 x 120 MiB,
// the slow level for every pair when the stores do not allocate (EXPERIMENTS.md 5.3, profiles/r05_b1_levels.txt).  So the
// first 112 MiB of each array keep the default policy -- 2 x 112 MiB fit with room to spare, their re-read always hits --
// and the grids behind them stream past the cache with the nt hint instead of fighting for it: every process gets the
// same time.  Multiple of 8 grids: the XCDs keep equal shares.  CARLE_B1_RESIDENT_BYTES overrides (0 = no limit: until round 4).
// 512-column grids only: 256-column batches of the same bytes bring 4x the grids and 4x the action records (8 MiB per step at
// N = 16384) into the cache, 2 x 112 MiB + records no longer fit and the slice thrashes (N=16384: 56.5 us against 45.0 us all
// streaming; with a 96 MiB slice 44.3: profiles/r05_b1_resident_ab.txt) -- they keep the round-4 policy.
static int64_t b1_resident_grids(int64_t n, int ww) {
    int64_t res = ww == 8 ? 112ll << 20 : 0;
    if (knobs().b1_resident_bytes >= 0) res = knobs().b1_resident_bytes;
    if (res == 0) return INT64_MAX;
    const int64_t grid_bytes = (int64_t)(64 * ww) * ww * 8;
    const int64_t g = res / grid_bytes / 8 * 8;
    return g >= n ? INT64_MAX : g;
}

template <int WW, int RULE_MODE>
static int launch_b1_mode(const uint64_t *u_in, const uint64_t *act, uint64_t *u_out, int64_t n, int ah,
                          int aw, uint32_t rule, int32_t *status, int slot, hipStream_t s, const B1EnvArgs *env = nullptr) {
    const int64_t n8 = (n + 7) / 8 * 8;         // the XCD-aware mapping walks the batch in groups of 8 grids
    int lds = b1_strip_mode(n, WW, act != nullptr && aw == 64);
    // the one-launch step keeps a resident slice and streams the rest at EVERY size from the LDS threshold on (see
    // b1_resident_grids): beyond the old streaming threshold too, the 112 MiB that do fit are re-read out of the cache
    // (512x512 N=6144 66.6-67.5 -> 63.7-65.6 us, N=8192 89.4-91.0 -> 86.7-88.9; CARLE_B1_HYBRID_ALL=0: all-streaming there)
    if (env && lds == 2 && knobs().b1_hybrid_all != 0 && (WW == 8 || knobs().b1_resident_bytes > 0)) lds = 1;
    // 64-row grids (a unit = a whole grid): persistent, just under 32 one-wave workgroups per CU (8 per SIMD).  8 x 1021: a
    // multiple of 8 (a wave stays on "its" XCD's grids) whose per-XCD stride 1021 is prime, so a wave's successive work
    // items rotate through the strips -- with a stride that is a multiple of the strip count every wave
    // kept ONE strip index, and the waves of the two strips under the action window finished last.
    // Taller grids (round 4): ONE WAVE PER STRIP, no persistence -- the dispatcher places a new wave wherever one retires,
    // which evens out what a fixed assignment cannot (the counters say the window costs the average wave 5.5 % and the
    // launch 14.5 %: profiles/r04_pmc_b1_window.txt).  Same-process sweeps of the launch cap (profiles/r04_b1_launch_cap.txt):
    // 512x512 with the window N=4096 44.2-46.7 -> 42.1-46.1 us (best 41.9 at 0.75 of the units), N=8192 96.6 -> 88.3 (0.70 ->
    // 0.77), 256x256 N=32768 98.4 -> 84.7 (0.70 -> 0.82); without a window 256x256 N=16384 42.5 -> 38.4 (0.87), 512x512 N=8192
    // 92.5 -> 87.5, 128x128 N=65536 44.2 -> 40.5.  (64-row grids lose with one wave per grid: 88 -> 95 us at N=262144.)
    // (the cap of the one-wave-per-strip launch keeps gridDim.x * blockDim.x = 64 * cap below 2^32: launch paths that validate
    // the global work size against UINT32_MAX would refuse more -- 128x128 grids beyond N = 33.5 M; the kernel's grid-stride
    // loop covers the rest)
    int64_t cap = WW > 1 ? ((1ll << 32) / 64 - 8) : 8 * 1021;
    if (const int64_t v = knobs().b1_blocks; v >= 8) cap = v / 8 * 8;   // experiment knob
    auto blocks = [&](int64_t wg) { return (unsigned)(wg < cap ? wg : cap); };
    // window launches of multi-strip grids walk half-grid units (see the kernel): 8192 waves = 32 per CU, all
    // resident at once, and the batch sizes of the BASELINE configs divide evenly over them
    int64_t ucap = 8192;
    if (const int64_t v = knobs().b1_units; v >= 8) ucap = v / 8 * 8;       // experiment knob
    const unsigned full = WW > 1 ? (unsigned)(n8 * 2 < ucap ? n8 * 2 : ucap) : blocks(n8 * WW);
    if (env && !act) {                            // one-launch step, no action
        B1EnvArgs ea = *env;
        ea.nt_from = b1_resident_grids(n, WW);
        const dim3 grid(blocks(n8 * WW)), block(64);
        if (lds == 2) hipLaunchKernelGGL((ca_step_b1_kernel<WW, 0, RULE_MODE, false, 2, true>), grid, block, 0, s, u_in, act, u_out, n, 0, 0, rule, status, slot, 0, WW, RandB1(), 0, ea);
        else if (lds == 1) hipLaunchKernelGGL((ca_step_b1_kernel<WW, 0, RULE_MODE, false, 1, true>), grid, block, 0, s, u_in, act, u_out, n, 0, 0, rule, status, slot, 0, WW, RandB1(), 0, ea);
        else hipLaunchKernelGGL((ca_step_b1_kernel<WW, 0, RULE_MODE, false, 0, true>), grid, block, 0, s, u_in, act, u_out, n, 0, 0, rule, status, slot, 0, WW, RandB1(), 0, ea);
        return check_launch("carle_env_step_b1");
    }
    if (env && aw == 64) {                        // one-launch step, the 64-column window
        B1EnvArgs ea = *env;
        ea.nt_from = b1_resident_grids(n, WW);
        // single strips -- the strips outside the window decide the master reset from a sample of the grid's record --
        // wherever the strips travel through LDS (512-column grids; 256-column grids from 64 MiB), half-grid units in the
        // row-per-lane mode.  With persistent waves the two walks tied at N=4096 and single strips won only beyond the Infinity
        // Cache (profiles/r04_kbench_b1_env.txt); with one wave per strip (above) single strips win everywhere: 512x512 N=256
        // 8.2 -> 5.7 us, N=1024 13.7 -> 12.2, N=2048 23.9 -> 22.2, N=4096 44.9-46.8 -> 42.1-46.1, 256x256 N=16384 52.3 -> 51.4,
        // N=32768 98.4 -> 84.7 (profiles/r04_b1_launch_cap.txt).  CARLE_B1_DBG bit 3 forces single strips, bit 4 half-grid units.
        int dbg = knobs().b1_dbg >= 0 ? (int)knobs().b1_dbg : 0;
        if (lds && WW > 1 && !(dbg & 16)) dbg |= 8;
        if (!lds || WW == 1 || (dbg & 16)) dbg &= ~8;
        const dim3 grid((dbg & 8) ? blocks(n8 * WW) : full), block(64);
        if (lds == 2) hipLaunchKernelGGL((ca_step_b1_kernel<WW, 1, RULE_MODE, false, 2, true>), grid, block, 0, s, u_in, act, u_out, n, ah, aw, rule, status, slot, 0, WW, RandB1(), dbg, ea);
        else if (lds == 1) hipLaunchKernelGGL((ca_step_b1_kernel<WW, 1, RULE_MODE, false, 1, true>), grid, block, 0, s, u_in, act, u_out, n, ah, aw, rule, status, slot, 0, WW, RandB1(), dbg, ea);
        else hipLaunchKernelGGL((ca_step_b1_kernel<WW, 1, RULE_MODE, false, 0, true>), grid, block, 0, s, u_in, act, u_out, n, ah, aw, rule, status, slot, 0, WW, RandB1(), dbg, ea);
        return check_launch("carle_env_step_b1");
    }
    if (env) return -1000;                        // wider windows: step kernel(s) + epilogue
    if (!act) {
        { if (lds == 2) { hipLaunchKernelGGL((ca_step_b1_kernel<WW, 0, RULE_MODE, false, 2>), dim3(blocks(n8 * WW)), dim3(64), 0,
                           s, u_in, act, u_out, n, 0, 0, rule, status, slot, 0, WW); } else if (lds == 1) { hipLaunchKernelGGL((ca_step_b1_kernel<WW, 0, RULE_MODE, false, 1>), dim3(blocks(n8 * WW)), dim3(64), 0,
                           s, u_in, act, u_out, n, 0, 0, rule, status, slot, 0, WW); } else { hipLaunchKernelGGL((ca_step_b1_kernel<WW, 0, RULE_MODE>), dim3(blocks(n8 * WW)), dim3(64), 0,
                           s, u_in, act, u_out, n, 0, 0, rule, status, slot, 0, WW); } }
        return check_launch("carle_ca_step_b1");
    }
    if (aw == 64) {
        // the usual 64x64 window: its toggle fits the full-occupancy instantiation -> one launch
        int dbg = 0;
        if (knobs().b1_dbg >= 0) dbg = (int)knobs().b1_dbg;       // experiment knob: 1 = window words fetched at their use
        const unsigned wl = (dbg & 8) && lds != 0 ? blocks(n8 * WW) : full;
        if (!lds) dbg &= ~8;
        { if (lds == 2) { hipLaunchKernelGGL((ca_step_b1_kernel<WW, 1, RULE_MODE, false, 2>), dim3(wl), dim3(64), 0,
                           s, u_in, act, u_out, n, ah, aw, rule, status, slot, 0, WW, RandB1(), dbg); } else if (lds == 1) { hipLaunchKernelGGL((ca_step_b1_kernel<WW, 1, RULE_MODE, false, 1>), dim3(wl), dim3(64), 0,
                           s, u_in, act, u_out, n, ah, aw, rule, status, slot, 0, WW, RandB1(), dbg); } else { hipLaunchKernelGGL((ca_step_b1_kernel<WW, 1, RULE_MODE>), dim3(wl), dim3(64), 0,
                           s, u_in, act, u_out, n, ah, aw, rule, status, slot, 0, WW, RandB1(), dbg); } }
        return check_launch("carle_ca_step_b1");
    }
    // wider windows: the strips whose rows, or whose two bordering rows, intersect the window rows
    // [oy, oy+ah) run the register-heavier instantiation, the others the lean one
    constexpr int H = 64 * WW;
    const int oy = (H - ah) / 2;
    int first = (oy > 0 ? oy - 1 : H - 1) / 64;              // strip of the row just above the window
    int last = ((oy + ah) % H) / 64;                          // strip of the row just below it
    if (ah == H) { first = 0; last = WW - 1; }
    int count = ((last - first + WW) % WW) + 1;
    if (count > WW) count = WW;
    { if (lds == 2) { hipLaunchKernelGGL((ca_step_b1_kernel<WW, 2, RULE_MODE, false, 2>), dim3(blocks(n8 * count)), dim3(64), 0, s,
                       u_in, act, u_out, n, ah, aw, rule, status, slot, first, count); } else if (lds == 1) { hipLaunchKernelGGL((ca_step_b1_kernel<WW, 2, RULE_MODE, false, 1>), dim3(blocks(n8 * count)), dim3(64), 0, s,
                       u_in, act, u_out, n, ah, aw, rule, status, slot, first, count); } else { hipLaunchKernelGGL((ca_step_b1_kernel<WW, 2, RULE_MODE>), dim3(blocks(n8 * count)), dim3(64), 0, s,
                       u_in, act, u_out, n, ah, aw, rule, status, slot, first, count); } }
    if (count < WW)
        { if (lds == 2) { hipLaunchKernelGGL((ca_step_b1_kernel<WW, 0, RULE_MODE, false, 2>), dim3(blocks(n8 * (WW - count))),
                           dim3(64), 0, s, u_in, act, u_out, n, 0, 0, rule, status, slot,
                           (first + count) % WW, WW - count); } else if (lds == 1) { hipLaunchKernelGGL((ca_step_b1_kernel<WW, 0, RULE_MODE, false, 1>), dim3(blocks(n8 * (WW - count))),
                           dim3(64), 0, s, u_in, act, u_out, n, 0, 0, rule, status, slot,
                           (first + count) % WW, WW - count); } else { hipLaunchKernelGGL((ca_step_b1_kernel<WW, 0, RULE_MODE>), dim3(blocks(n8 * (WW - count))),
                           dim3(64), 0, s, u_in, act, u_out, n, 0, 0, rule, status, slot,
                           (first + count) % WW, WW - count); } }
    return check_launch("carle_ca_step_b1");
}

template <int WW>
static int launch_b1_rand(const uint64_t *u_in, uint64_t *u_out, int64_t n, uint32_t rule, const RandB1 &ra,
                          int32_t *status, int slot, hipStream_t s) {
    const int64_t n8 = (n + 7) / 8 * 8;
    if (n8 * WW > 0x7FFFFFFFll) return fail_arg(CARLE_E_DIMS, "carle_ca_step_b1_rand: batch too large");
    const int lds = b1_strip_mode(n, WW);
    const int64_t wg = WW > 1 ? n8 * 2 : n8, cap = WW > 1 ? 8192 : 8 * 1021;     // half-grid units (see the kernel)
    const dim3 grid((unsigned)(wg < cap ? wg : cap)), block(64);
    const uint64_t *none = nullptr;
    if (rule == kLifeMask)
        { if (lds == 2) { hipLaunchKernelGGL((ca_step_b1_kernel<WW, 1, kRuleLife, true, 2>), grid, block, 0, s, u_in, none, u_out, n, 64, 64,
                           rule, status, slot, 0, WW, ra); } else if (lds == 1) { hipLaunchKernelGGL((ca_step_b1_kernel<WW, 1, kRuleLife, true, 1>), grid, block, 0, s, u_in, none, u_out, n, 64, 64,
                           rule, status, slot, 0, WW, ra); } else { hipLaunchKernelGGL((ca_step_b1_kernel<WW, 1, kRuleLife, true>), grid, block, 0, s, u_in, none, u_out, n, 64, 64,
                           rule, status, slot, 0, WW, ra); } }
    else if (rule == kHighLifeMask)
        { if (lds == 2) { hipLaunchKernelGGL((ca_step_b1_kernel<WW, 1, kRuleHighLife, true, 2>), grid, block, 0, s, u_in, none, u_out, n, 64,
                           64, rule, status, slot, 0, WW, ra); } else if (lds == 1) { hipLaunchKernelGGL((ca_step_b1_kernel<WW, 1, kRuleHighLife, true, 1>), grid, block, 0, s, u_in, none, u_out, n, 64,
                           64, rule, status, slot, 0, WW, ra); } else { hipLaunchKernelGGL((ca_step_b1_kernel<WW, 1, kRuleHighLife, true>), grid, block, 0, s, u_in, none, u_out, n, 64,
                           64, rule, status, slot, 0, WW, ra); } }
    else
        { if (lds == 2) { hipLaunchKernelGGL((ca_step_b1_kernel<WW, 1, kRuleGeneric, true, 2>), grid, block, 0, s, u_in, none, u_out, n, 64, 64,
                           rule, status, slot, 0, WW, ra); } else if (lds == 1) { hipLaunchKernelGGL((ca_step_b1_kernel<WW, 1, kRuleGeneric, true, 1>), grid, block, 0, s, u_in, none, u_out, n, 64, 64,
                           rule, status, slot, 0, WW, ra); } else { hipLaunchKernelGGL((ca_step_b1_kernel<WW, 1, kRuleGeneric, true>), grid, block, 0, s, u_in, none, u_out, n, 64, 64,
                           rule, status, slot, 0, WW, ra); } }
    return check_launch("carle_ca_step_b1_rand");
}

template <int WW>
static int launch_b1(const uint64_t *u_in, const uint64_t *act, uint64_t *u_out, int64_t n, int ah,
                     int aw, uint32_t rule, int32_t *status, int slot, hipStream_t s, const B1EnvArgs *env = nullptr) {
    if (rule == kLifeMask)
        return launch_b1_mode<WW, kRuleLife>(u_in, act, u_out, n, ah, aw, rule, status, slot, s, env);
    if (rule == kHighLifeMask)
        return launch_b1_mode<WW, kRuleHighLife>(u_in, act, u_out, n, ah, aw, rule, status, slot, s, env);
    return launch_b1_mode<WW, kRuleGeneric>(u_in, act, u_out, n, ah, aw, rule, status, slot, s, env);
}

}  // namespace carle

using namespace carle;

namespace carle {
static int step_b1_checked(const uint64_t *u_in, const uint64_t *action, uint64_t *u_out, int64_t n, int h, int w, int ah,
                           int aw, uint32_t rule, int32_t *status, int slot, void *stream, const B1EnvArgs *env);

// CARLE_STEP_DEFERRED on packed state: the whole step in one launch; -1000 where that form does not exist
int env_step_one_launch_b1(const uint64_t *u_in, const uint64_t *action, uint64_t *u_out, float *reward, int64_t n, int h,
                           int w, int ah, int aw, uint32_t rule, int32_t *status, int slot, int step_flags, void *stream) {
    if (n < 1 || !status || !aligned16(status) || (action && aw != 64) || knobs().two_launch_step > 0) return -1000;
    B1EnvArgs ea;
    ea.reward = reward;
    ea.publish = (step_flags & CARLE_STEP_PUBLISH_FLAGS) ? 1 : 0;
    ea.merge_moments = (step_flags & CARLE_STEP_MERGE_MOMENTS) ? 1 : 0;
    return step_b1_checked(u_in, action, u_out, n, h, w, ah, aw, rule, status, slot, stream, &ea);
}
}  // namespace carle

extern "C" int carle_ca_step_b1(const uint64_t *u_in, const uint64_t *action, uint64_t *u_out,
                                int64_t n, int h, int w, int ah, int aw, uint32_t rule,
                                int32_t *status, int slot, void *stream) {
    return step_b1_checked(u_in, action, u_out, n, h, w, ah, aw, rule, status, slot, stream, nullptr);
}

static int carle::step_b1_checked(const uint64_t *u_in, const uint64_t *action, uint64_t *u_out, int64_t n, int h, int w,
                                  int ah, int aw, uint32_t rule, int32_t *status, int slot, void *stream,
                                  const B1EnvArgs *env) {
    if (n == 0) return 0;
    if (!u_in || !u_out) return fail_arg(CARLE_E_NULL, "carle_ca_step_b1: u_in/u_out is NULL");
    if (int e = check_dims(n, h, w, action != nullptr, ah, aw)) return e;
    if (w != 64 && w != 128 && w != 256 && w != 512)
        return fail_arg(CARLE_E_DIMS, "carle_ca_step_b1: W must be 64, 128, 256 or 512");
    if (action && (aw % 64) != 0)
        return fail_arg(CARLE_E_DIMS, "carle_ca_step_b1: packed action needs AW % 64 == 0");
    if (rule >> 18) return fail_arg(CARLE_E_RULE, "carle_ca_step_b1: rule mask exceeds 18 bits");
    if (slot != 0 && slot != 1) return fail_arg(CARLE_E_DIMS, "carle_ca_step_b1: slot must be 0 or 1");
    if (!aligned16(u_in) || !aligned16(u_out) || (action && !aligned16(action)))
        return fail_arg(CARLE_E_ALIGN, "carle_ca_step_b1: pointers must be 16-byte aligned");
    const size_t bytes = (size_t)n * h * (w / 8);
    if (overlap(u_in, bytes, u_out, bytes))
        return fail_arg(CARLE_E_ALIAS, "carle_ca_step_b1: u_in and u_out overlap");
    if (n == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (w) {
        case 64: return launch_b1<1>(u_in, action, u_out, n, ah, aw, rule, status, slot, s, env);
        case 128: return launch_b1<2>(u_in, action, u_out, n, ah, aw, rule, status, slot, s, env);
        case 256: return launch_b1<4>(u_in, action, u_out, n, ah, aw, rule, status, slot, s, env);
        default: return launch_b1<8>(u_in, action, u_out, n, ah, aw, rule, status, slot, s, env);
    }
}

// carle_ca_step_b1 with the 64x64 action window drawn in the kernel (device-side RandomAgent,
// carle/agents.py:35-42): identical to packing carle_bernoulli_f32(seed, offset) and passing it in.
extern "C" int carle_ca_step_b1_rand(const uint64_t *u_in, uint64_t *u_out, int64_t n, int h, int w, int ah, int aw,
                                     uint32_t rule, float p, uint64_t seed, uint64_t offset, int32_t *status,
                                     int slot, void *stream) {
    if (n == 0) return 0;
    if (!u_in || !u_out) return fail_arg(CARLE_E_NULL, "carle_ca_step_b1_rand: u_in/u_out is NULL");
    if (int e = check_dims(n, h, w, true, ah, aw)) return e;
    if (w != 64 && w != 128 && w != 256 && w != 512)
        return fail_arg(CARLE_E_DIMS, "carle_ca_step_b1_rand: W must be 64, 128, 256 or 512");
    if (aw != 64) return fail_arg(CARLE_E_DIMS, "carle_ca_step_b1_rand: the fused random agent covers the 64x64 window");
    if (rule >> 18) return fail_arg(CARLE_E_RULE, "carle_ca_step_b1_rand: rule mask exceeds 18 bits");
    if (slot != 0 && slot != 1) return fail_arg(CARLE_E_DIMS, "carle_ca_step_b1_rand: slot must be 0 or 1");
    if (!aligned16(u_in) || !aligned16(u_out))
        return fail_arg(CARLE_E_ALIGN, "carle_ca_step_b1_rand: pointers must be 16-byte aligned");
    const size_t bytes = (size_t)n * h * (w / 8);
    if (overlap(u_in, bytes, u_out, bytes)) return fail_arg(CARLE_E_ALIAS, "carle_ca_step_b1_rand: u_in and u_out overlap");
    RandB1 ra;
    ra.p = p; ra.seed = seed; ra.offset = offset;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (w) {
        case 64: return launch_b1_rand<1>(u_in, u_out, n, rule, ra, status, slot, s);
        case 128: return launch_b1_rand<2>(u_in, u_out, n, rule, ra, status, slot, s);
        case 256: return launch_b1_rand<4>(u_in, u_out, n, rule, ra, status, slot, s);
        default: return launch_b1_rand<8>(u_in, u_out, n, rule, ra, status, slot, s);
    }
}

extern "C" int carle_ca_multistep_b1(const uint64_t *u_in, uint64_t *u_out, int64_t n, int h, int w,
                                     uint32_t rule, int k, int32_t *status, void *stream) {
    if (n == 0) return 0;
    if (!u_in || !u_out) return fail_arg(CARLE_E_NULL, "carle_ca_multistep_b1: u_in/u_out is NULL");
    if (int e = check_dims(n, h, w, false, 0, 0)) return e;
    if (w != 64 && w != 128 && w != 256 && w != 512)
        return fail_arg(CARLE_E_DIMS, "carle_ca_multistep_b1: W must be 64, 128, 256 or 512");
    if (k < 0) return fail_arg(CARLE_E_DIMS, "carle_ca_multistep_b1: k must be >= 0");
    if (rule >> 18) return fail_arg(CARLE_E_RULE, "carle_ca_multistep_b1: rule mask exceeds 18 bits");
    if (!aligned16(u_in) || !aligned16(u_out) || (status && !aligned16(status)))
        return fail_arg(CARLE_E_ALIGN, "carle_ca_multistep_b1: pointers must be 16-byte aligned");
    const size_t bytes = (size_t)n * h * (w / 8);
    if (u_in != u_out && overlap(u_in, bytes, u_out, bytes))
        return fail_arg(CARLE_E_ALIAS, "carle_ca_multistep_b1: u_in and u_out overlap partially");
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (w) {
        case 64: return launch_multistep<1>(u_in, u_out, n, rule, k, status, s);
        case 128: return launch_multistep<2>(u_in, u_out, n, rule, k, status, s);
        case 256: return launch_multistep<4>(u_in, u_out, n, rule, k, status, s);
        default: return launch_multistep<8>(u_in, u_out, n, rule, k, status, s);
    }
}

extern "C" int carle_pack_f32_b1(const float *src, uint64_t *dst, int64_t n, int h, int w,
                                 int32_t *status, int slot, void *stream) {
    if (!src || !dst) return fail_arg(CARLE_E_NULL, "carle_pack_f32_b1: NULL pointer");
    if (n < 0 || h < 1 || w < 64 || (w % 64) != 0)
        return fail_arg(CARLE_E_DIMS, "carle_pack_f32_b1: W must be a positive multiple of 64");
    if (slot != 0 && slot != 1) return fail_arg(CARLE_E_DIMS, "carle_pack_f32_b1: slot must be 0 or 1");
    const int64_t words = n * (int64_t)h * (w / 64);
    if (words == 0) return 0;
    const int64_t blocks = (words + 63) / 64;
    if (blocks > 0x7FFFFFFFll) return fail_arg(CARLE_E_DIMS, "carle_pack_f32_b1: batch too large");
    hipLaunchKernelGGL(pack_f32_b1_kernel, dim3((unsigned)blocks), dim3(64), 0,
                       static_cast<hipStream_t>(stream), src, dst, words, status, slot);
    return check_launch("carle_pack_f32_b1");
}

extern "C" int carle_unpack_b1_f32(const uint64_t *src, float *dst, int64_t n, int h, int w,
                                   void *stream) {
    if (!src || !dst) return fail_arg(CARLE_E_NULL, "carle_unpack_b1_f32: NULL pointer");
    if (n < 0 || h < 1 || w < 64 || (w % 64) != 0)
        return fail_arg(CARLE_E_DIMS, "carle_unpack_b1_f32: W must be a positive multiple of 64");
    if (!aligned16(dst)) return fail_arg(CARLE_E_ALIGN, "carle_unpack_b1_f32: dst must be 16-byte aligned");
    const int64_t quads = n * (int64_t)h * (w / 4);
    if (quads == 0) return 0;
    const int64_t blocks = (quads + 255) / 256;
    if (blocks > 0x7FFFFFFFll) return fail_arg(CARLE_E_DIMS, "carle_unpack_b1_f32: batch too large");
    hipLaunchKernelGGL(unpack_b1_f32_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), src, reinterpret_cast<float4 *>(dst), quads);
    return check_launch("carle_unpack_b1_f32");
}

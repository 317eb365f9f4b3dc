// ca_cells.hpp -- device code shared by the cell-per-element CA kernels (ca_step_cells.hip) and the
// kernel that fuses the RND2D bonus into the CA step (ca_rnd_fused.hip): cell traits (fp32 /
// uint8), the whole-batch action predicates, and the "patch" generation of one 64x64 grid per
// wavefront (see ca_step_cells.hip for the layout).
#pragma once
#include "ca_bits.hpp"
#include "common.hpp"

namespace carle {

static constexpr uint32_t kOneBits = 0x3F800000u;   // 1.0f

// 1 if the fp32 bit pattern is non-zero as a number (NaN counts, -0.0 does not): the
// truthiness torch.logical_xor applies to its operands (carle/env.py:182).
__device__ __forceinline__ uint32_t nonzero_bit(uint32_t bits) {
    return min(bits & 0x7FFFFFFFu, 1u);
}
__device__ __forceinline__ uint32_t nibble_of(const uint4 &v) {
    return nonzero_bit(v.x) | (nonzero_bit(v.y) << 1) | (nonzero_bit(v.z) << 2) |
           (nonzero_bit(v.w) << 3);
}
__device__ __forceinline__ uint64_t shfl64(uint64_t x, int src_lane) {
    uint32_t lo = __shfl((uint32_t)x, src_lane, 64);
    uint32_t hi = __shfl((uint32_t)(x >> 32), src_lane, 64);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t rotl64(uint64_t x, int s) { return (x << s) | (x >> (64 - s)); }
__device__ __forceinline__ uint64_t rotr64(uint64_t x, int s) { return (x >> s) | (x << (64 - s)); }

template <bool NT>
__device__ __forceinline__ uint4 ld16(const uint4 *p) {
    if (NT) {
        uint4 v;
        v.x = __builtin_nontemporal_load(&p->x);
        v.y = __builtin_nontemporal_load(&p->y);
        v.z = __builtin_nontemporal_load(&p->z);
        v.w = __builtin_nontemporal_load(&p->w);
        return v;
    }
    return *p;
}
template <bool NT>
__device__ __forceinline__ void st16(uint4 *p, const uint4 &v) {
    if (NT) {
        __builtin_nontemporal_store(v.x, &p->x);
        __builtin_nontemporal_store(v.y, &p->y);
        __builtin_nontemporal_store(v.z, &p->z);
        __builtin_nontemporal_store(v.w, &p->w);
    } else {
        *p = v;
    }
}

// Per-lane summary of the action elements a lane has seen, for the two whole-batch predicates
// of CARLE.step: any element non-zero (env.py:191), every element == 1 (env.py:208).  Raw bit
// patterns are folded with AND / OR; the cell traits say what "non-zero" and "one" look like.
struct ActionSeen {
    uint32_t all_and = 0xFFFFFFFFu, all_or = 0u;
    __device__ __forceinline__ void add(const uint4 &a) {
        all_and &= a.x & a.y & a.z & a.w;
        all_or |= a.x | a.y | a.z | a.w;
    }
    __device__ __forceinline__ void add(uint32_t a) { all_and &= a; all_or |= a; }
    template <class A>
    __device__ __forceinline__ bool any_nonzero() const { return (all_or & A::kNonzeroMask) != 0; }
    template <class A>
    __device__ __forceinline__ bool not_all_ones() const {
        const bool seen = !(all_and == 0xFFFFFFFFu && all_or == 0u);
        return seen && (all_and != A::kOnes || all_or != A::kOnes);
    }
};

// Cell traits: how four horizontally adjacent cells travel between memory and a 4-bit nibble.
struct F32Cells {
    using T = float;
    using Quad = uint4;
    static constexpr uint32_t kOnes = kOneBits;             // raw pattern of an all-ones word
    static constexpr uint32_t kNonzeroMask = 0x7FFFFFFFu;   // -0.0 is zero (env.py:182 truthiness)
    template <bool NT>
    static __device__ __forceinline__ Quad load(const T *p) { return ld16<NT>(reinterpret_cast<const uint4 *>(p)); }
    static __device__ __forceinline__ uint32_t nibble(const Quad &v) { return nibble_of(v); }
    static __device__ __forceinline__ Quad splat(uint32_t w) { return make_uint4(w, w, w, w); }
    template <bool NT>
    static __device__ __forceinline__ void store(T *p, uint32_t nib) {
        uint4 o4;
        o4.x = (nib & 1u) ? kOneBits : 0u;
        o4.y = (nib & 2u) ? kOneBits : 0u;
        o4.z = (nib & 4u) ? kOneBits : 0u;
        o4.w = (nib & 8u) ? kOneBits : 0u;
        st16<NT>(reinterpret_cast<uint4 *>(p), o4);
    }
    static __device__ __forceinline__ uint32_t alive(T v) { return nonzero_bit(__float_as_uint(v)); }
    static __device__ __forceinline__ uint32_t raw(T v) { return __float_as_uint(v); }
    static __device__ __forceinline__ T make(uint32_t bit) { return bit ? 1.0f : 0.0f; }
};

struct U8Cells {
    using T = uint8_t;
    using Quad = uint32_t;
    static constexpr uint32_t kOnes = 0x01010101u;
    static constexpr uint32_t kNonzeroMask = 0xFFFFFFFFu;
    template <bool NT>
    static __device__ __forceinline__ Quad load(const T *p) {
        const uint32_t *q = reinterpret_cast<const uint32_t *>(p);
        return NT ? __builtin_nontemporal_load(q) : *q;
    }
    static __device__ __forceinline__ Quad splat(uint32_t w) { return w; }
    // byte j non-zero -> bit j: high bit of every non-zero byte, then a multiply gathers them
    static __device__ __forceinline__ uint32_t nibble(Quad x) {
        const uint32_t t = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
        return (((t >> 7) * 0x01020408u) >> 24) & 0xFu;
    }
    template <bool NT>
    static __device__ __forceinline__ void store(T *p, uint32_t nib) {
        const uint32_t v = ((nib & 0xFu) * 0x00204081u) & 0x01010101u;     // bit j -> byte j
        uint32_t *q = reinterpret_cast<uint32_t *>(p);
        if (NT) __builtin_nontemporal_store(v, q);
        else *q = v;
    }
    static __device__ __forceinline__ uint32_t alive(T v) { return v != 0; }
    static __device__ __forceinline__ uint32_t raw(T v) { return (uint32_t)v * 0x01010101u; }
    static __device__ __forceinline__ T make(uint32_t bit) { return (T)bit; }
};
__device__ __forceinline__ void see(ActionSeen &s, const uint4 &a) { s.add(a); }
__device__ __forceinline__ void see(ActionSeen &s, uint32_t a) { s.add(a); }

// Returns (wave-uniform) whether some action element seen by this wavefront differs from one.
template <class A>
__device__ __forceinline__ bool publish_flags(const ActionSeen &s, int32_t *status, int slot) {
    if (status == nullptr) return true;
    const bool any = __ballot(s.any_nonzero<A>()) != 0ull;
    const bool notone = __ballot(s.not_all_ones<A>()) != 0ull;
    if ((threadIdx.x & 63) == 0) {      // every writer stores the same value: benign
        if (any) status[CARLE_ST_ANY_NONZERO(slot)] = 1;
        if (notone) status[CARLE_ST_NOT_ALL_ONES(slot)] = 1;
    }
    return notone;
}

// A CARLE_STEP_DEFERRED step (one launch: ca_step_64_kernel<..., ENV>) leaves its counter update pending in
// word CARLE_ST_PENDING; whoever touches the counters next applies it first.  Works on the register copy of the
// first three 16-byte words of the status block; returns true if something was pending.
__device__ __forceinline__ bool settle_pending(uint4 &w0, uint4 &w1, uint4 &w2) {
    const uint32_t pend = w1.y;                                           // CARLE_ST_PENDING (index 5)
    if (!(pend & 1u)) return false;
    const bool ps = (pend & 4u) != 0u, has_action = (pend & 2u) != 0u;
    // (masks, not `ps ? w0.y : w0.x`: the compiler turns that select into a dynamically indexed copy of w0 in scratch -- or,
    // promoted, into 16 bytes of LDS per thread of every kernel that settles: 4 KB of the fused CA + RND2D kernel's 37.5)
    const uint32_t pm = 0u - (uint32_t)ps;
    const bool any = has_action && ((w0.y & pm) | (w0.x & ~pm)) != 0u;
    const bool reset = has_action && ((w0.w & pm) | (w0.z & ~pm)) == 0u;
    int64_t step_number = (int64_t)(((uint64_t)w1.w << 32) | w1.z);
    int64_t since = (int64_t)(((uint64_t)w2.y << 32) | w2.x);
    int64_t resets = (int64_t)(((uint64_t)w2.w << 32) | w2.z);
    if (!any) since += 1;                       // env.py:199-200
    if (reset) { step_number = 0; since = 0; resets += 1; } else { step_number += 1; }   // env.py:216,142-145 / 230
    if (ps) { w0.y = 0u; w0.w = 0u; } else { w0.x = 0u; w0.z = 0u; }      // its flags are consumed
    w1.x = reset ? 1u : 0u;
    w1.y = 0u;
    w1.z = (uint32_t)step_number; w1.w = (uint32_t)((uint64_t)step_number >> 32);
    w2.x = (uint32_t)since; w2.y = (uint32_t)((uint64_t)since >> 32);
    w2.z = (uint32_t)resets; w2.w = (uint32_t)((uint64_t)resets >> 32);
    return true;
}

// The same on the block in memory, by ONE thread (kernels that update the counters in place).
__device__ __forceinline__ void settle_status_in_place(int32_t *st) {
    uint4 *s4 = reinterpret_cast<uint4 *>(st);
    uint4 w0 = s4[0], w1 = s4[1], w2 = s4[2];
    if (settle_pending(w0, w1, w2)) {
        s4[0] = w0; s4[1] = w1; s4[2] = w2;
        s4[4] = make_uint4(0u, 0u, 0u, 0u);                               // SCAN_TICKET / SCAN_DONE
    }
}

// Reward moments of a one-launch CA + RND2D step (carle_env_step_f32_rnd with CARLE_STEP_DEFERRED | MERGE_MOMENTS): the
// rewards only exist when the launch ends, so its workgroups leave (sum, sum of squares) pairs behind and the NEXT call on
// the status block -- or carle_env_settle -- reduces them (fixed order: lane l adds pairs l, l + 64, ...; then a xor
// tree) and Chan-merges the triple (count, mean, M2) exactly as the epilogue kernel does.  Called by all 64 lanes of ONE
// wavefront of the launch, before anything else of the launch touches the moment accumulators.
__device__ __forceinline__ void settle_moments_wave(int32_t *st, int lane) {
    const uint4 w5 = reinterpret_cast<const uint4 *>(st)[5];             // words 20..23
    const uint32_t npart = w5.z;
    if (npart == 0u) return;                                             // wave-uniform
    const uint4 w3 = reinterpret_cast<const uint4 *>(st)[3];
    const double2 *part = reinterpret_cast<const double2 *>(((uint64_t)w5.y << 32) | w5.x);
    double s = 0.0, q = 0.0;
    for (uint32_t base = 0; base < npart; base += 256u) {               // four pairs per lane in flight
        double2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = base + 64u * u + (uint32_t)lane;
            v[u] = part[i < npart ? i : 0u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (base + 64u * u + (uint32_t)lane < npart) { s += v[u].x; q += v[u].y; }
    }
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) { s += __shfl_xor(s, k, 64); q += __shfl_xor(q, k, 64); }
    if (lane == 0) {
        double *mom_out = moments_out_slot(st, reinterpret_cast<double *>(((uint64_t)w3.y << 32) | w3.x));
        double *mom_acc = reinterpret_cast<double *>(((uint64_t)w3.w << 32) | w3.z);
        if (mom_out && mom_acc) {
            const double nb = (double)w5.w, mean = nb > 0.0 ? s / nb : 0.0;
            double m2 = q - s * mean;
            if (m2 < 0.0) m2 = 0.0;
            const double na = mom_acc[0], ma = mom_acc[1], qa = mom_acc[2], tot = na + nb;
            mom_out[0] = nb; mom_out[1] = mean; mom_out[2] = m2;
            if (tot > 0.0) {
                const double delta = mean - ma;
                mom_acc[0] = tot;
                mom_acc[1] = ma + delta * nb / tot;
                mom_acc[2] = qa + m2 + delta * delta * na * nb / tot;
            }
        }
        st[CARLE_ST_MOM_COUNT] = 0;
    }
}

// Bookkeeping of a one-launch step of the BARE environment (ENV instantiations of the step kernels), by all 64 lanes of ONE
// wavefront of the launch, out of line (it runs once per launch, behind everything else of its wavefront): pending reward
// moments and counters of the previous step are applied, the moments of this step's all-zero reward merged, its flag slot's
// neighbour recycled, and this step's counter update left pending (has_action) or applied at once (no action: no
// predicates, no reset, env.py:191,208).
static __device__ __noinline__ void env_step_bookkeeping(int32_t *status, int slot, int64_t n, int has_action, int merge_moments) {
    const int lane = threadIdx.x & 63;
    settle_moments_wave(status, lane);
    if (lane != 0) return;
    uint4 *s4 = reinterpret_cast<uint4 *>(status);
    uint4 w0 = s4[0], w1 = s4[1], w2 = s4[2];
    const uint4 w3 = s4[3];
    settle_pending(w0, w1, w2);
    if (merge_moments) {
        double *mom_out = moments_out_slot(status, reinterpret_cast<double *>(((uint64_t)w3.y << 32) | w3.x));
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
    if (!has_action) {
        int64_t step_number = (int64_t)(((uint64_t)w1.w << 32) | w1.z) + 1;
        int64_t since = (int64_t)(((uint64_t)w2.y << 32) | w2.x) + 1;
        w1.x = 0u;
        w1.z = (uint32_t)step_number; w1.w = (uint32_t)((uint64_t)step_number >> 32);
        w2.x = (uint32_t)since; w2.y = (uint32_t)((uint64_t)since >> 32);
    } else {
        w1.y = 1u | 2u | (slot ? 4u : 0u);                     // CARLE_ST_PENDING: this step
    }
    // the words of THIS step's slot are written by the other wavefronts meanwhile: dword stores where a 16-byte store
    // would cover them
    s4[1] = w1; s4[2] = w2;
    const int other = slot ^ 1;
    status[CARLE_ST_ANY_NONZERO(other)] = 0;
    status[CARLE_ST_NOT_ALL_ONES(other)] = 0;
    status[CARLE_ST_SCAN_TICKET(other)] = 0;
    status[CARLE_ST_SCAN_DONE(other)] = 0;
}

// Counter / flag bookkeeping at the end of a step (env.py:142-145, 199-200, 216, 230), by ONE thread:
// w0..w2 = the first three 16-byte words of the status block as read at the start of the epilogue (a pending
// deferred step is applied first).
__device__ __forceinline__ void finish_status(int32_t *st, uint4 w0, uint4 w1, uint4 w2,
                                              int slot, bool reset, bool any) {
    settle_pending(w0, w1, w2);
    int64_t step_number = (int64_t)(((uint64_t)w1.w << 32) | w1.z);       // int32 index 6..7
    int64_t since = (int64_t)(((uint64_t)w2.y << 32) | w2.x);             // index 8..9
    int64_t resets = (int64_t)(((uint64_t)w2.w << 32) | w2.z);            // index 10..11
    if (!any) since += 1;                       // env.py:199-200
    if (reset) {                                // env.py:216 -> reset(): env.py:142-145
        step_number = 0;
        since = 0;
        resets += 1;
    } else {
        step_number += 1;                       // env.py:230
    }
    uint4 o0 = w0, o1 = w1, o2;
    if (slot) { o0.x = 0u; o0.z = 0u; } else { o0.y = 0u; o0.w = 0u; }   // recycle slot^1
    o1.x = reset ? 1u : 0u;                     // CARLE_ST_LAST_RESET (index 4)
    o1.y = 0u;                                  // CARLE_ST_PENDING
    o1.z = (uint32_t)step_number; o1.w = (uint32_t)((uint64_t)step_number >> 32);
    o2.x = (uint32_t)since; o2.y = (uint32_t)((uint64_t)since >> 32);
    o2.z = (uint32_t)resets; o2.w = (uint32_t)((uint64_t)resets >> 32);
    uint4 *so = reinterpret_cast<uint4 *>(st);
    so[0] = o0; so[1] = o1; so[2] = o2;              // words 12..15 (moment pointers) stay
    so[4] = make_uint4(0u, 0u, 0u, 0u);              // SCAN_TICKET / SCAN_DONE of both slots
}

// ---------------------------------------------------------------------------------------------------------------
// Whole-batch master reset decided INSIDE the step kernel (CARLE_STEP_DEFERRED): `torch.mean(action) == 1.0`
// (env.py:208) depends on every action element of the batch, and every output cell depends on it.  A wavefront
// whose own grid's action is NOT all ones knows the answer (no reset) -- that is every wavefront of every step an
// agent drives.  A wavefront that finds its own action all ones calls this: the wavefronts in that situation verify
// the N action grids together -- a ticket per grid from SCAN_TICKET, one count per all-ones grid in SCAN_DONE,
// bit 31 of SCAN_DONE as soon as a grid is not -- with agent-scope atomics (coherent across the XCDs' L2s).  A
// wavefront only ever waits for wavefronts that hold a ticket, i.e. that are running: nothing waits for a wavefront
// that has not been dispatched (N may exceed the resident wavefronts).  The action tensor is read-only in the kernel.
// Returns true iff every element of the batch's action equals one.  Cost when it does: the action is read twice.
// ---------------------------------------------------------------------------------------------------------------
#ifdef CARLE_SCAN_INLINE
#define CARLE_SCAN_ATTR __forceinline__
#else
#define CARLE_SCAN_ATTR __noinline__
#endif
// `grid_quads`: 16-byte quads of one grid's action; `ones`: the dword pattern of four / thirty-two all-one elements
static __device__ __forceinline__ bool batch_action_all_ones_inline(const void *act, int64_t n, int64_t grid_quads,
                                                                    uint32_t ones, int32_t *st, int slot) {
    constexpr uint32_t kAbort = 0x80000000u;
    uint32_t *ticket = reinterpret_cast<uint32_t *>(st + CARLE_ST_SCAN_TICKET(slot));
    uint32_t *done = reinterpret_cast<uint32_t *>(st + CARLE_ST_SCAN_DONE(slot));
    const int lane = threadIdx.x & 63;
    for (;;) {
        uint32_t h = 0;
        if (lane == 0) h = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        h = __builtin_amdgcn_readfirstlane(h);
        if ((int64_t)h >= n) break;
        const uint4 *p = reinterpret_cast<const uint4 *>(act) + (int64_t)h * grid_quads;
        bool all1 = true;
        for (int64_t q = lane; q < grid_quads; q += 64) {
            const uint4 a = p[q];
            all1 = all1 && a.x == ones && a.y == ones && a.z == ones && a.w == ones;
        }
        const bool all = __ballot(!all1) == 0ull;
        uint32_t d = 0;
        if (lane == 0)
            d = all ? __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u
                    : __hip_atomic_fetch_or(done, kAbort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) | kAbort;
        d = __builtin_amdgcn_readfirstlane(d);
        if (d & kAbort) return false;
    }
    for (;;) {                                   // every grid has a verifier; wait for those still reading
        uint32_t d = 0;
        if (lane == 0) d = __hip_atomic_fetch_or(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        d = __builtin_amdgcn_readfirstlane(d);
        if (d & kAbort) return false;
        if ((int64_t)d >= n) return true;
        __builtin_amdgcn_s_sleep(16);
    }
}

// Out of line for the kernels that call it where many registers are live (a call in a cold branch costs them nothing
// on the hot path; inlined, its loop would take part in their register allocation): the fp32 / uint8 step kernels.
static __device__ CARLE_SCAN_ATTR bool batch_action_all_ones_raw(const void *act, int64_t n, int64_t grid_quads, uint32_t ones,
                                                          int32_t *st, int slot) {
    return batch_action_all_ones_inline(act, n, grid_quads, ones, st, slot);
}

template <class A>
__device__ __forceinline__ bool batch_action_all_ones(const typename A::T *act, int64_t n, int aw, int32_t *st, int slot) {
    // AW % 8 == 0: a grid's action is a whole number of uint4s
    return batch_action_all_ones_raw(act, n, (int64_t)aw * aw * (int64_t)sizeof(typename A::T) / 16, A::kOnes, st, slot);
}

// One generation of a 64x64 torus held as one 64-bit "patch" word per lane: lane (q = L/16,
// k = L%16), nibble i = row 4i+q, columns 4k..4k+3.
__device__ __forceinline__ uint64_t patch_generation64(uint64_t c, int lane, uint32_t rule) {
    const int q = lane >> 4, k = lane & 15;
    // horizontal: west/east neighbour of every cell
    const int grp = lane & 48;
    const uint64_t cw = shfl64(c, grp | ((k + 15) & 15));
    const uint64_t ce = shfl64(c, grp | ((k + 1) & 15));
    const uint64_t wst = ((c << 1) & 0xEEEEEEEEEEEEEEEEull) | ((cw >> 3) & 0x1111111111111111ull);
    const uint64_t est = ((c >> 1) & 0x7777777777777777ull) | ((ce << 3) & 0x8888888888888888ull);
    uint64_t h0, h1, p0, p1;
    row_sums(wst, c, est, h0, h1, p0, p1);
    // vertical: triple sums of the rows above and below
    const int lu = (lane + 48) & 63, ld = (lane + 16) & 63;
    uint64_t a0 = shfl64(h0, lu), a1 = shfl64(h1, lu);
    uint64_t b0 = shfl64(h0, ld), b1 = shfl64(h1, ld);
    if (q == 0) { a0 = rotl64(a0, 4); a1 = rotl64(a1, 4); }
    if (q == 3) { b0 = rotr64(b0, 4); b1 = rotr64(b1, 4); }
    return apply_rule(neighbour_count(a0, a1, b0, b1, p0, p1), c, rule);
}

}  // namespace carle

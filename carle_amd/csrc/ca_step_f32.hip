// ca_step_f32.hip -- one CA generation on fp32 (N,1,H,W) grids: the reference's own format.
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
// Generic path (any even square size): one thread per cell, neighbours through L1/L2.
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

// Per-lane summary of the action elements a lane has seen, for the two whole-batch predicates
// of CARLE.step: any element non-zero (env.py:191), every element == 1.0 (env.py:208).
struct ActionSeen {
    uint32_t all_and = 0xFFFFFFFFu, all_or = 0u;
    __device__ __forceinline__ void add(const uint4 &a) {
        all_and &= a.x & a.y & a.z & a.w;
        all_or |= a.x | a.y | a.z | a.w;
    }
    __device__ __forceinline__ void add(uint32_t a) { all_and &= a; all_or |= a; }
    __device__ __forceinline__ bool any_nonzero() const { return (all_or & 0x7FFFFFFFu) != 0; }
    __device__ __forceinline__ bool not_all_ones() const {
        const bool seen = !(all_and == 0xFFFFFFFFu && all_or == 0u);
        return seen && (all_and != kOneBits || all_or != kOneBits);
    }
};

__device__ __forceinline__ void publish_flags(const ActionSeen &s, int32_t *status, int slot) {
    if (status == nullptr) return;
    const bool any = __ballot(s.any_nonzero()) != 0ull;
    const bool notone = __ballot(s.not_all_ones()) != 0ull;
    if ((threadIdx.x & 63) == 0) {      // every writer stores the same value: benign
        if (any) status[CARLE_ST_ANY_NONZERO(slot)] = 1;
        if (notone) status[CARLE_ST_NOT_ALL_ONES(slot)] = 1;
    }
}

// ---------------------------------------------------------------------------------------------
// 64x64 fast path: one wavefront per grid
// ---------------------------------------------------------------------------------------------
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

template <bool HAS_ACT, bool NT_LD = false, bool NT_ST = false, int BLOCK = 256>
__global__ __launch_bounds__(BLOCK) void ca_step_f32_64_kernel(
    const uint4 *__restrict__ u_in, const float *__restrict__ act, uint4 *__restrict__ u_out,
    int64_t n, int aw, uint32_t rule, int32_t *__restrict__ status, int slot) {
    const int lane = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    if (g >= n) return;                              // wave-uniform
    const int q = lane >> 4, k = lane & 15;
    const uint4 *gin = u_in + g * 1024 + lane;       // 4096 floats = 1024 x 16 B per grid

    uint4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = ld16<NT_LD>(gin + i * 64);
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) c |= (uint64_t)nibble_of(v[i]) << (4 * i);

    if (HAS_ACT) {
        // window offset o = (64-aw)/2 is a multiple of 4 (host guarantees aw % 8 == 0), so a
        // lane's four columns are all inside or all outside the window
        const int o = (64 - aw) >> 1;
        const int ax = 4 * k - o;
        const bool col_in = ax >= 0 && ax < aw;
        const float *abase = act + g * (int64_t)aw * aw + ax;
        ActionSeen seen;
        uint64_t tog = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ay = 4 * i + q - o;
            if (col_in && ay >= 0 && ay < aw) {
                const uint4 a = ld16<NT_LD>(reinterpret_cast<const uint4 *>(abase + (int64_t)ay * aw));
                seen.add(a);
                tog |= (uint64_t)nibble_of(a) << (4 * i);
            }
        }
        c ^= tog;                                    // toggle, env.py:182
        publish_flags(seen, status, slot);
    }

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

    const uint64_t nxt = apply_rule(neighbour_count(a0, a1, b0, b1, p0, p1), c, rule);

    uint4 *gout = u_out + g * 1024 + lane;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t nib = (uint32_t)(nxt >> (4 * i));
        uint4 o4;
        o4.x = (nib & 1u) ? kOneBits : 0u;
        o4.y = (nib & 2u) ? kOneBits : 0u;
        o4.z = (nib & 4u) ? kOneBits : 0u;
        o4.w = (nib & 8u) ? kOneBits : 0u;
        st16<NT_ST>(gout + i * 64, o4);
    }
}

// ---------------------------------------------------------------------------------------------
// generic path: any even square grid, one thread per cell
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ca_step_f32_generic_kernel(
    const float *__restrict__ u_in, const float *__restrict__ act, float *__restrict__ u_out,
    int64_t n, int h, int w, int ah, int aw, uint32_t rule, int32_t *__restrict__ status,
    int slot) {
    const int64_t cells = n * (int64_t)h * w;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int oy = (h - ah) >> 1, ox = (w - aw) >> 1;
    ActionSeen seen;
    if (idx < cells) {
        const int x = (int)(idx % w);
        const int y = (int)((idx / w) % h);
        const int64_t g = idx / ((int64_t)h * w);
        const float *gu = u_in + g * (int64_t)h * w;
        const float *ga = act ? act + g * (int64_t)ah * aw : nullptr;
        auto alive = [&](int yy, int xx) -> uint32_t {
            uint32_t a = nonzero_bit(__float_as_uint(gu[yy * w + xx]));
            const int ay = yy - oy, ax = xx - ox;
            if (ga && ay >= 0 && ay < ah && ax >= 0 && ax < aw)
                a ^= nonzero_bit(__float_as_uint(ga[ay * aw + ax]));
            return a;
        };
        const int ym = y == 0 ? h - 1 : y - 1, yp = y == h - 1 ? 0 : y + 1;
        const int xm = x == 0 ? w - 1 : x - 1, xp = x == w - 1 ? 0 : x + 1;
        const uint32_t cnt = alive(ym, xm) + alive(ym, x) + alive(ym, xp) + alive(y, xm) +
                             alive(y, xp) + alive(yp, xm) + alive(yp, x) + alive(yp, xp);
        const uint32_t c = alive(y, x);
        const uint32_t nxt = (rule >> (c ? cnt : 9 + cnt)) & 1u;
        u_out[idx] = nxt ? 1.0f : 0.0f;
        const int ay = y - oy, ax = x - ox;           // this cell owns its own action element
        if (ga && ay >= 0 && ay < ah && ax >= 0 && ax < aw)
            seen.add(__float_as_uint(ga[ay * aw + ax]));
    }
    if (act) publish_flags(seen, status, slot);
}

// ---------------------------------------------------------------------------------------------
// toggle only (CARLE.apply_action called on its own, carle/env.py:150-182): elementwise
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ca_toggle_f32_kernel(
    const float *__restrict__ u_in, const float *__restrict__ act, float *__restrict__ u_out,
    int64_t n, int h, int w, int ah, int aw) {
    const int64_t cells = n * (int64_t)h * w;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= cells) return;
    const int x = (int)(idx % w);
    const int y = (int)((idx / w) % h);
    const int64_t g = idx / ((int64_t)h * w);
    uint32_t a = nonzero_bit(__float_as_uint(u_in[idx]));
    const int ay = y - ((h - ah) >> 1), ax = x - ((w - aw) >> 1);
    if (ay >= 0 && ay < ah && ax >= 0 && ax < aw)
        a ^= nonzero_bit(__float_as_uint(act[(g * ah + ay) * (int64_t)aw + ax]));
    u_out[idx] = a ? 1.0f : 0.0f;
}

}  // namespace carle

using namespace carle;

extern "C" int carle_ca_step_f32(const float *u_in, const float *action, float *u_out, int64_t n,
                                 int h, int w, int ah, int aw, uint32_t rule, int32_t *status,
                                 int slot, void *stream) {
    if (!u_in || !u_out) return fail_arg(CARLE_E_NULL, "carle_ca_step_f32: u_in/u_out is NULL");
    if (int e = check_dims(n, h, w, action != nullptr, ah, aw)) return e;
    if (rule >> 18) return fail_arg(CARLE_E_RULE, "carle_ca_step_f32: rule mask exceeds 18 bits");
    if (slot != 0 && slot != 1) return fail_arg(CARLE_E_DIMS, "carle_ca_step_f32: slot must be 0 or 1");
    if (!aligned16(u_in) || !aligned16(u_out) || (action && !aligned16(action)))
        return fail_arg(CARLE_E_ALIGN, "carle_ca_step_f32: pointers must be 16-byte aligned");
    const size_t bytes = (size_t)n * h * w * sizeof(float);
    if (overlap(u_in, bytes, u_out, bytes))
        return fail_arg(CARLE_E_ALIAS, "carle_ca_step_f32: u_in and u_out overlap");
    if (n == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);

    if (h == 64 && w == 64 && (!action || (aw % 8) == 0)) {
        // one wavefront per workgroup, non-temporal loads and stores: measured best of the
        // {plain, nt} x {64,128,256,512}-thread variants (tools/kbench, profiles/r01_kbench.txt)
        if (n > 0x7FFFFFFFll) return fail_arg(CARLE_E_DIMS, "carle_ca_step_f32: batch too large");
        const unsigned blocks = (unsigned)n;
        if (action)
            hipLaunchKernelGGL((ca_step_f32_64_kernel<true, true, true, 64>), dim3(blocks), dim3(64),
                               0, s, reinterpret_cast<const uint4 *>(u_in), action,
                               reinterpret_cast<uint4 *>(u_out), n, aw, rule, status, slot);
        else
            hipLaunchKernelGGL((ca_step_f32_64_kernel<false, true, true, 64>), dim3(blocks), dim3(64),
                               0, s, reinterpret_cast<const uint4 *>(u_in), action,
                               reinterpret_cast<uint4 *>(u_out), n, 0, rule, status, slot);
        return check_launch("carle_ca_step_f32(64x64)");
    }
    const int64_t cells = n * (int64_t)h * w;
    const int64_t blocks = (cells + 255) / 256;
    if (blocks > 0x7FFFFFFFll) return fail_arg(CARLE_E_DIMS, "carle_ca_step_f32: batch too large");
    hipLaunchKernelGGL(ca_step_f32_generic_kernel, dim3((unsigned)blocks), dim3(256), 0, s, u_in,
                       action, u_out, n, h, w, action ? ah : 0, action ? aw : 0, rule, status, slot);
    return check_launch("carle_ca_step_f32(generic)");
}

extern "C" int carle_ca_toggle_f32(const float *u_in, const float *action, float *u_out,
                                   int64_t n, int h, int w, int ah, int aw, void *stream) {
    if (!u_in || !u_out || !action)
        return fail_arg(CARLE_E_NULL, "carle_ca_toggle_f32: NULL pointer");
    if (int e = check_dims(n, h, w, true, ah, aw)) return e;
    if (n == 0) return 0;
    const int64_t cells = n * (int64_t)h * w;
    const int64_t blocks = (cells + 255) / 256;
    if (blocks > 0x7FFFFFFFll) return fail_arg(CARLE_E_DIMS, "carle_ca_toggle_f32: batch too large");
    hipLaunchKernelGGL(ca_toggle_f32_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), u_in, action, u_out, n, h, w, ah, aw);
    return check_launch("carle_ca_toggle_f32");
}

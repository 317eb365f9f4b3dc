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
#include "ca_bits.hpp"
#include "common.hpp"

namespace carle {

__device__ __forceinline__ uint64_t dpp64_ror1(uint64_t x) {      // lane i <- lane i-1 (0 <- 63)
    const uint32_t lo = __builtin_amdgcn_update_dpp(0u, (uint32_t)x, 0x13C, 0xF, 0xF, false);
    const uint32_t hi = __builtin_amdgcn_update_dpp(0u, (uint32_t)(x >> 32), 0x13C, 0xF, 0xF, false);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t dpp64_rol1(uint64_t x) {      // lane i <- lane i+1 (63 <- 0)
    const uint32_t lo = __builtin_amdgcn_update_dpp(0u, (uint32_t)x, 0x134, 0xF, 0xF, false);
    const uint32_t hi = __builtin_amdgcn_update_dpp(0u, (uint32_t)(x >> 32), 0x134, 0xF, 0xF, false);
    return ((uint64_t)hi << 32) | lo;
}

// lane i <- lane i-1, lane 0 keeps `old` (wave_shr:1); lane i <- lane i+1, lane 63 keeps `old`
__device__ __forceinline__ uint64_t dpp64_shr1(uint64_t x, uint64_t old) {
    const uint32_t lo = __builtin_amdgcn_update_dpp((uint32_t)old, (uint32_t)x, 0x138, 0xF, 0xF, false);
    const uint32_t hi = __builtin_amdgcn_update_dpp((uint32_t)(old >> 32), (uint32_t)(x >> 32), 0x138, 0xF, 0xF, false);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t dpp64_shl1(uint64_t x, uint64_t old) {
    const uint32_t lo = __builtin_amdgcn_update_dpp((uint32_t)old, (uint32_t)x, 0x130, 0xF, 0xF, false);
    const uint32_t hi = __builtin_amdgcn_update_dpp((uint32_t)(old >> 32), (uint32_t)(x >> 32), 0x130, 0xF, 0xF, false);
    return ((uint64_t)hi << 32) | lo;
}

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

// XOR the packed action row (AWW words at column offset ox = 32*(WW-AWW)) into a grid row.
// `ok` = this lane's row is inside the window.  Register-light on purpose: each target word
// re-reads the (at most two) action words that overlap it (L1 hits) instead of staging the action
// row in a register array, so the action path does not raise the kernel's VGPR allocation.
template <int WW>
__device__ __forceinline__ void toggle_row(uint64_t (&c)[WW], const uint64_t *__restrict__ arow,
                                           int aww, bool ok, uint64_t &seen_or, uint64_t &seen_and) {
    const int wo = (WW - aww) >> 1;                         // uniform word offset of the window
    const bool half = ((WW - aww) & 1) != 0;                // ox % 64 == 32
#pragma unroll
    for (int j = 0; j < WW; ++j) {
        const int i = j - wo;                               // action word aligned with grid word j
        uint64_t cur = 0ull, prv = 0ull;
        if (ok && i >= 0 && i < aww) {
            cur = arow[i];
            seen_or |= cur;
            seen_and &= cur;
        }
        if (half && ok && i - 1 >= 0 && i - 1 < aww) prv = arow[i - 1];
        c[j] ^= half ? ((cur << 32) | (prv >> 32)) : cur;
    }
}

template <int WW, bool HAS_ACT>
__global__ __launch_bounds__(64) void ca_step_b1_kernel(
    const uint64_t *__restrict__ u_in, const uint64_t *__restrict__ act, uint64_t *__restrict__ u_out,
    int64_t n, int ah, int aw, uint32_t rule, int32_t *__restrict__ status, int slot,
    int strip_first, int strip_count) {
    // the launch covers strips [strip_first, strip_first + strip_count) (mod WW) of every grid:
    // the host launches the strips that can meet the action window with HAS_ACT = true and the
    // others with the leaner HAS_ACT = false instantiation (half the registers)
    constexpr int H = 64 * WW;
    constexpr bool MULTI = WW > 1;                          // more than one 64-row strip per grid
    const int lane = threadIdx.x;
    const int64_t wid = blockIdx.x;
    const int64_t g = wid / strip_count;
    const int r0 = ((strip_first + (int)(wid % strip_count)) % WW) * 64;
    if (g >= n) return;
    const int r = r0 + lane;
    const uint64_t *gin = u_in + g * (int64_t)H * WW;
    const int rt = (r0 + H - 1) % H, rb = (r0 + 64) % H;   // rows bordering the strip (y torus)
    const int hr = lane == 0 ? rt : (lane == 63 ? rb : -1);

    uint64_t c[WW], hal[WW];
    load_row<WW>(gin + (int64_t)r * WW, c);
#pragma unroll
    for (int j = 0; j < WW; ++j) hal[j] = 0ull;
    if (MULTI && hr >= 0) load_row<WW>(gin + (int64_t)hr * WW, hal);

    if (HAS_ACT) {
        const int oy = (H - ah) >> 1, aww = aw >> 6;
        // wave-uniform: does the strip or one of its halo rows meet the action window at all?
        const bool touches = (r0 < oy + ah && r0 + 64 > oy) ||
                             (MULTI && ((rt >= oy && rt < oy + ah) || (rb >= oy && rb < oy + ah)));
        if (touches) {
            const uint64_t *ga = act + g * (int64_t)ah * aww;
            uint64_t s_or = 0ull, s_and = ~0ull, d0 = 0ull, d1 = ~0ull;
            const bool in = r >= oy && r < oy + ah;
            toggle_row<WW>(c, ga + (int64_t)(in ? r - oy : 0) * aww, aww, in, s_or, s_and);
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

    // horizontal sums of the own row and (lanes 0/63) of the halo row
    uint64_t h0[WW], h1[WW], p0[WW], p1[WW], g0[WW], g1[WW];
#pragma unroll
    for (int j = 0; j < WW; ++j) {
        const uint64_t wst = (c[j] << 1) | (c[(j + WW - 1) % WW] >> 63);
        const uint64_t est = (c[j] >> 1) | (c[(j + 1) % WW] << 63);
        row_sums(wst, c[j], est, h0[j], h1[j], p0[j], p1[j]);
        if (MULTI) {
            const uint64_t hw = (hal[j] << 1) | (hal[(j + WW - 1) % WW] >> 63);
            const uint64_t he = (hal[j] >> 1) | (hal[(j + 1) % WW] << 63);
            uint64_t q0, q1;
            row_sums(hw, hal[j], he, g0[j], g1[j], q0, q1);
        }
    }

    uint64_t nx[WW];
#pragma unroll
    for (int j = 0; j < WW; ++j) {
        uint64_t a0, a1, b0, b1;
        if (MULTI) {
            // lanes with no source keep `old` = the sums of the halo row they loaded themselves
            a0 = dpp64_shr1(h0[j], g0[j]); a1 = dpp64_shr1(h1[j], g1[j]);     // row above
            b0 = dpp64_shl1(h0[j], g0[j]); b1 = dpp64_shl1(h1[j], g1[j]);     // row below
        } else {
            a0 = dpp64_ror1(h0[j]); a1 = dpp64_ror1(h1[j]);                   // rotate = y torus seam
            b0 = dpp64_rol1(h0[j]); b1 = dpp64_rol1(h1[j]);
        }
        nx[j] = apply_rule_auto(neighbour_count(a0, a1, b0, b1, p0[j], p1[j]), c[j], rule);
    }
    store_row<WW>(u_out + (g * (int64_t)H + r) * WW, nx);
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

template <int WW>
static int launch_b1(const uint64_t *u_in, const uint64_t *act, uint64_t *u_out, int64_t n, int ah,
                     int aw, uint32_t rule, int32_t *status, int slot, hipStream_t s) {
    if (n * WW > 0x7FFFFFFFll) return fail_arg(CARLE_E_DIMS, "carle_ca_step_b1: batch too large");
    if (!act) {
        hipLaunchKernelGGL((ca_step_b1_kernel<WW, false>), dim3((unsigned)(n * WW)), dim3(64), 0, s,
                           u_in, act, u_out, n, 0, 0, rule, status, slot, 0, WW);
        return check_launch("carle_ca_step_b1");
    }
    // strips whose rows, or whose two bordering rows, intersect the window rows [oy, oy+ah)
    constexpr int H = 64 * WW;
    const int oy = (H - ah) / 2;
    int first = (oy > 0 ? oy - 1 : H - 1) / 64;              // strip of the row just above the window
    int last = ((oy + ah) % H) / 64;                          // strip of the row just below it
    if (ah == H) { first = 0; last = WW - 1; }
    int count = ((last - first + WW) % WW) + 1;
    if (count > WW) count = WW;
    hipLaunchKernelGGL((ca_step_b1_kernel<WW, true>), dim3((unsigned)(n * count)), dim3(64), 0, s,
                       u_in, act, u_out, n, ah, aw, rule, status, slot, first, count);
    if (count < WW)
        hipLaunchKernelGGL((ca_step_b1_kernel<WW, false>), dim3((unsigned)(n * (WW - count))), dim3(64),
                           0, s, u_in, act, u_out, n, 0, 0, rule, status, slot, (first + count) % WW,
                           WW - count);
    return check_launch("carle_ca_step_b1");
}

}  // namespace carle

using namespace carle;

extern "C" int carle_ca_step_b1(const uint64_t *u_in, const uint64_t *action, uint64_t *u_out,
                                int64_t n, int h, int w, int ah, int aw, uint32_t rule,
                                int32_t *status, int slot, void *stream) {
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
        case 64: return launch_b1<1>(u_in, action, u_out, n, ah, aw, rule, status, slot, s);
        case 128: return launch_b1<2>(u_in, action, u_out, n, ah, aw, rule, status, slot, s);
        case 256: return launch_b1<4>(u_in, action, u_out, n, ah, aw, rule, status, slot, s);
        default: return launch_b1<8>(u_in, action, u_out, n, ah, aw, rule, status, slot, s);
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

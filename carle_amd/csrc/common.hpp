// common.hpp -- host-side helpers shared by the C-ABI entry points of libcarle_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/carle_hip.h"

// gfx950 only.  Besides the instruction selection (v_bitop3, DPP, v_permlane32_swap, LDS-DMA), the last-arrival finish
// protocols of nets_tiled.hip / reduce.hip / ca_cells.hpp hand data between workgroups WITHOUT agent-scope fences:
// write-through (sc1) stores, every storing wave's s_waitcnt vmcnt(0), a workgroup barrier, ONE relaxed agent-scope
// ticket, sc1 loads by the workgroup whose ticket came last.  That is a hand-off form validated on gfx950 / ROCm 7.2
// (MI355X_MICROARCH.md, "Valid forms" of inter-workgroup visibility), not an architectural guarantee: a build for any
// other architecture must not compile these paths silently.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libcarle_hip is written for gfx950 (MI355X): its fence-free inter-workgroup hand-offs are validated there only"
#endif

namespace carle {

// thread-local error text behind carle_last_error()
std::string &last_error();

inline int fail_arg(int code, const char *msg) {
    last_error() = msg;
    return code;
}

inline int fail_hip(hipError_t e, const char *where) {
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", where, hipGetErrorString(e));
    last_error() = buf;
    return -(int)e;
}

// call after every kernel launch (launch-configuration errors only; no sync)
inline int check_launch(const char *where) {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail_hip(e, where);
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline bool overlap(const void *a, size_t na, const void *b, size_t nb) {
    auto pa = reinterpret_cast<uintptr_t>(a), pb = reinterpret_cast<uintptr_t>(b);
    return pa < pb + nb && pb < pa + na;
}

// The reference's consistent domain (carle/env.py:119-132): square, even, centred window.
inline int check_dims(int64_t n, int h, int w, bool has_action, int ah, int aw) {
    if (n < 0 || h < 2 || w < 2 || h != w || (h & 1))
        return fail_arg(CARLE_E_DIMS, "grid must be square with even H == W >= 2 and N >= 0");
    if (has_action && (ah != aw || ah < 1 || ah > h || ((h - ah) & 1)))
        return fail_arg(CARLE_E_DIMS, "action window must be square, AH == AW <= H, (H-AH) even");
    return 0;
}

// Experiment knobs (environment variables CARLE_*): read ONCE, when the first entry point that needs them runs
// (a function-local static in abi.hip), never per launch -- getenv() costs a linear scan of the environment on a path
// whose problem is host time, and is not safe against a concurrent setenv().  carle_reload_knobs() re-reads them
// (tests and the A/B tools that switch a knob inside one process); a value < 0 means "not set: library default".
struct Knobs {
    int64_t stream_store_min_bytes;      // CARLE_STREAM_STORE_MIN_BYTES: fp32 state arrays from this size use streaming stores
    int64_t b1_lds_min_bytes;            // CARLE_B1_LDS_MIN_BYTES: packed strips go through the LDS image from this size
    int64_t b1_nt_min_bytes;             // CARLE_B1_NT_MIN_BYTES: ... with streaming hints from this size
    int64_t b1_resident_bytes;           // CARLE_B1_RESIDENT_BYTES: bytes per packed state array kept cacheable (0 = all)
    int64_t b1_hybrid_all;               // CARLE_B1_HYBRID_ALL: 0 = all-streaming beyond the old threshold (until round 4) instead of resident slice + streamed rest
    int64_t b1_blocks, b1_units;         // CARLE_B1_BLOCKS / CARLE_B1_UNITS: launch caps of the packed kernel
    int64_t wide_blocks;                 // CARLE_WIDE_BLOCKS: launch cap of the wavefront-per-strip kernels
    int64_t rnd_blocks;                  // CARLE_RND_BLOCKS: workgroups of the fused CA + RND2D kernel
    int64_t packed_grids;                // CARLE_PACKED_GRIDS: grids per wavefront of the packed-input 64x64 step (1 / 2; unset: by n)
    int64_t rnd_cached_load_max_bytes;   // CARLE_RND_CACHED_LOAD_MAX_BYTES
    int64_t net_blocks;                  // CARLE_NET_BLOCKS: workgroups of the 64x64 reward-net kernels
    int64_t b1_dbg;                      // CARLE_B1_DBG
    int64_t two_launch_step;             // CARLE_TWO_LAUNCH_STEP: 1 = step kernel + epilogue kernel everywhere (A/B)
};
const Knobs &knobs();

// reward += scale * bonus as torch evaluates it (carle/mcl.py:403-405): the product and the sum are rounded separately.
// (HIP's __fmul_rn / __fadd_rn are plain operators and contract into an fma like any other; the pragma does not.)
__device__ __forceinline__ float reward_accumulate(float reward, float scale, float bonus) {
#pragma clang fp contract(off)
    const float prod = scale * bonus;
    return reward + prod;
}

// Where the reward moments (count, mean, M2) of a step go: the triple at CARLE_ST_MOMENTS_OUT, or -- while a trace is
// installed (CARLE_ST_TRACE_CAP > 1: MOMENTS_OUT then points at that many rows) -- the next row of the ring.  Called by the
// ONE thread that writes a step's moments, once per step.
__device__ __forceinline__ double *moments_out_slot(int32_t *st, double *base) {
    const uint32_t cap = (uint32_t)st[CARLE_ST_TRACE_CAP];
    if (cap <= 1u || base == nullptr) return base;
    const uint32_t i = (uint32_t)st[CARLE_ST_TRACE_IDX];
    st[CARLE_ST_TRACE_IDX] = (int32_t)(i + 1u);
    return base + 3 * (size_t)(i % cap);
}

// k steps of a bare environment in one launch (all-zero rewards): k rows; the first k - 1 are written here as (n, 0, 0),
// the last one is returned for the caller to write
__device__ __forceinline__ double *moments_out_slot_steps(int32_t *st, double *base, double n, int steps) {
    double *row = moments_out_slot(st, base);
    if ((uint32_t)st[CARLE_ST_TRACE_CAP] <= 1u || base == nullptr) return row;
    for (int t = 1; t < steps; ++t) {
        row[0] = n; row[1] = 0.0; row[2] = 0.0;
        row = moments_out_slot(st, base);
    }
    return row;
}

// Arguments of the one-launch CA + RND2D step (ca_step_rnd64_kernel<..., ENV>, ca_rnd_fused.hip)
struct RndEnvArgs {
    float *reward = nullptr;             // N floats: reward_scale * bonus
    float reward_scale = 1.0f;
    int publish = 1, merge_moments = 0;
    double *partials = nullptr;          // gridDim.x (sum, sum of squares) pairs, this step's buffer
    uint64_t *bits_out = nullptr;        // N x 64 row words: a bit-packed copy of the new state for a reward net stacked on the
                                         // RND2D (AE2D reads 512 B instead of 16 KiB per grid); nullable
};

// Hides a uniform pointer from loop-invariant code motion: the scalar loads behind it stay in the phase
// that uses them (from the scalar cache) instead of being hoisted out of the grid / tile loop, where 124-140
// uniform weights overflow the SGPR file and spill into VGPR lanes (v_writelane / v_readlane per use).
typedef const float __attribute__((address_space(4))) kfloat;   // constant address space: s_load
typedef float f32x2 __attribute__((ext_vector_type(2)));          // an operand of v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32
__device__ __forceinline__ kfloat *opaque_uniform(const float *p) {
    asm volatile("" : "+s"(p));
    return (kfloat *)p;
}

// Global -> LDS copy of N4 float4 by NT threads with every load in flight before the first store (a plain copy loop
// compiles to load, s_waitcnt vmcnt(0), ds_write per iteration: one dependent memory round trip each).  N4 is a
// multiple of NT: no tail, hence no branch the compiler could sink a load into.
template <int N4, int NT>
__device__ __forceinline__ void lds_copy_f4(float4 *dst, const float4 *__restrict__ src, int tid) {
    static_assert(N4 % NT == 0, "no tail");
    constexpr int K = N4 / NT;
    float4 v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = src[tid + NT * k];
#pragma unroll
    for (int k = 0; k < K; ++k) dst[tid + NT * k] = v[k];
}

}  // namespace carle

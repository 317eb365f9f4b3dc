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
//   E  wave 0: conv2 + pooling, the two 64->16 linear layers, tanh, the bonus (rnd_tail).
// The bonus of the ALL-DEAD grid is written to bonus[n]: the epilogue (ca_finish_kernel) uses it
// for every instance when the step turns out to be a whole-batch master reset (env.py:208-216).
#include "ca_cells.hpp"
#include "rnd_device.hpp"

namespace carle {

// RndBlockShared (rnd_device.hpp) + the toggled input rows: 32,528 B.  The 2 x 64 pooled features
// of phase E overlay rows[1..64] (dead by then: phase D is behind a barrier, phase B of the next grid
// rewrites every row).  Four workgroups per CU: a fifth fits the LDS but needs <= 96 VGPRs, and the 16
// spilled registers cost more than the extra occupancy gives (measured 72 vs 45 us per step at N=4096).
struct CaRndShared {
    RndBlockShared net;
    uint32_t rows_in[64][2];     // toggled state, one 64-bit word per row (lo, hi)
};

template <bool HAS_ACT>
__global__ __launch_bounds__(256, 4) void ca_step_rnd64_kernel(
    const float *__restrict__ u_in, const float *__restrict__ act, float *__restrict__ u_out, int64_t n,
    int aw, uint32_t rule, int32_t *__restrict__ status, int slot, const float *__restrict__ ws,
    float *__restrict__ bonus) {
    __shared__ CaRndShared s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    rnd_block_init(s.net, ws, tid);
    __syncthreads();

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

    uint4 v[4], a[4];
    auto issue = [&](int64_t g) {
        const uint4 *gin = reinterpret_cast<const uint4 *>(u_in) + g * 1024 + 256 * wv + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = ld16<true>(gin + i * 64);
        if (HAS_ACT) {
            const float *abase = act + g * (int64_t)aw * aw + ax;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = make_uint4(0u, 0u, 0u, 0u);
                if ((rows_in_win >> i) & 1u)
                    a[i] = ld16<true>(reinterpret_cast<const uint4 *>(abase + (int64_t)(16 * wv + 4 * i + q - o) * aw));
            }
        }
    };

    const int64_t first = blockIdx.x, stride = gridDim.x;
    if (first < n) issue(first);
    ActionSeen seen;
    for (int64_t g = first; g <= n; g += stride) {
        const bool live = g < n;                      // g == n: the all-dead grid
        if (live) {
            // ---- A: nibbles of this lane's four rows, toggled; row words by OR over the 16 lanes
            uint32_t nib[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                nib[i] = nibble_of(v[i]);
                if (HAS_ACT) {
                    if ((rows_in_win >> i) & 1u) seen.add(a[i]);
                    nib[i] ^= nibble_of(a[i]);        // toggle, env.py:182
                }
            }
            if (g + stride < n) issue(g + stride);    // the next grid's quads travel during B..E
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t x = row_nibbles_to_dword(nib[i], k);
                if ((k & 7) == 7) s.rows_in[16 * wv + 4 * i + q][k >> 3] = x;
            }
        }
        __syncthreads();

        // ---- B: one generation, lane = row (wave 0)
        if (live && wv == 0) {
            const uint64_t c = ((uint64_t)s.rows_in[lane][1] << 32) | s.rows_in[lane][0];
            const uint64_t wst = rotl64(c, 1), est = rotr64(c, 1);      // torus seam in x
            uint64_t h0, h1, p0, p1;
            row_sums(wst, c, est, h0, h1, p0, p1);
            const int lu = (lane + 63) & 63, ld = (lane + 1) & 63;      // torus seam in y
            const uint64_t a0 = shfl64(h0, lu), a1 = shfl64(h1, lu);
            const uint64_t b0 = shfl64(h0, ld), b1 = shfl64(h1, ld);
            s.net.rows[1 + lane] = apply_rule(neighbour_count(a0, a1, b0, b1, p0, p1), c, rule);
        } else if (wv == 0) {
            s.net.rows[1 + lane] = 0ull;
        }
        __syncthreads();

        // ---- C: fp32 store of the new state (the lane's own four quads)
        if (live) {
            float *gout = u_out + g * 4096 + 1024 * wv + 4 * lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint64_t row = s.net.rows[1 + 16 * wv + 4 * i + q];
                F32Cells::store<true>(gout + i * 256, (uint32_t)(row >> (4 * k)) & 0xFu);
            }
        }

        // ---- D: conv1 (table lookup) + relu + 4x4 max-pool; thread = pooling window
        rnd_block_conv1(s.net, tid);
        __syncthreads();

        // ---- E: conv2 + pool, linear layers, bonus (wave 0)
        if (wv == 0) {
            float (*feat)[64] = reinterpret_cast<float (*)[64]>(&s.net.rows[1]);
            const float b = rnd_tail(&s.net.planes[0][0], feat, s.net.lin, ws, lane, g, nullptr, nullptr);
            if (lane == 0) bonus[g] = b;
        }
        // the next iteration's first barrier orders E's plane reads before D's next plane writes;
        // rows_in / rows_out are rewritten only after that barrier and the one behind B
    }
    if (HAS_ACT) publish_flags<F32Cells>(seen, status, slot);
}

void rnd2d_prepare_tables(const float *pp, const float *rp, float *ws, hipStream_t s);

// validated by the caller (carle_env_step_f32_rnd); bonus has n + 1 elements
int ca_step_rnd64(const float *u_in, const float *action, float *u_out, int64_t n, int aw, uint32_t rule,
                  int32_t *status, int slot, const float *pp, const float *rp, float *ws, float *bonus,
                  bool tables_valid, hipStream_t s) {
    if (!tables_valid) rnd2d_prepare_tables(pp, rp, ws, s);
    int64_t blocks = (n + 1 + 3) / 4;                 // persistent: 4 grids per workgroup when the
    if (blocks > 1024) blocks = 1024;                 // batch allows; 4 workgroups are resident per CU
    if (action)
        hipLaunchKernelGGL(ca_step_rnd64_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, u_in, action,
                           u_out, n, aw, rule, status, slot, ws, bonus);
    else
        hipLaunchKernelGGL(ca_step_rnd64_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, u_in, action,
                           u_out, n, 64, rule, status, slot, ws, bonus);
    return check_launch("carle_env_step_f32_rnd(fused CA + RND2D)");
}

}  // namespace carle

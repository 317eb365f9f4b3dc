// nets_tiled.hip -- RND2D forward for BINARY grids larger than 64x64 (H = W a multiple of 64: the
// reference's default 256x256, carle/env.py:21-22), fused per 64x64-cell TILE: the workgroup-per-grid kernel
// of nets_fused.hip with a halo.  Same networks and numerics (carle/mcl.py:262-296, 343-352; eval mode).
//   tile kernel   one workgroup per (grid, tile), persistent.  The tile's cells + 5 cells of halo (74x74)
//                 become 74 row words of 74 bits in LDS -- fp32 observations by wave ballots over coalesced
//                 row loads -- conv1 is the 512-entry table per channel; the 18x18 pooled windows of the
//                 tile + 1 window of halo (zero outside the grid = conv2's padding) fill the same 18x24 LDS
//                 planes the 64x64 kernel uses (there the ring is the zero border), so conv2 + pool is
//                 rnd_block_conv2 unchanged; the tile's 8x8 features per net then meet their 64 columns of
//                 the two (D -> 16) linear layers: 32 partial sums per tile to the workspace;
//   finish kernel per grid: partial sums of its tiles in a fixed order + bias, tanh, mean_k (t_k - p_k)^2.
// Two launches instead of the staged path's three, and nothing but the observation is read from HBM
// (staged: 6 pooled planes + 2 feature planes written and re-read).
#include "rnd_device.hpp"
#include "train_bodies.hpp"
#include <type_traits>
#include <utility>

namespace carle {

void rnd2d_prepare_tables(const float *pp, const float *rp, float *ws, hipStream_t s);      // nets_fused.hip

static constexpr int kTileRows = 74;                   // 64 + 2 * 5 cells
static constexpr int kTileWin = 18;                    // pooled windows per side: 16 + 2

struct RndTileShared {
    RndBlockShared net;                                // tables, planes, conv2 parameters (lin / rows unused)
    uint64_t rows[kTileRows + 2][2];                   // [r][0] = columns x0 .. x0+63, [r][1] = x0+64 .. x0+73
    float feat[2][64];
    int last;                                          // RndTileFinish: this workgroup delivered the last tile of its grid
};

// 6 bits of a 74-bit row starting at local column c (0 <= c <= 68)
__device__ __forceinline__ uint32_t row_seg6(uint64_t lo, uint64_t hi, int c) {
    const uint64_t v = c < 64 ? ((lo >> c) | (c > 0 ? hi << (64 - c) : 0ull)) : (hi >> (c - 64));
    return (uint32_t)v & 0x3Fu;
}

// conv1 (tables) + relu + 4x4 max-pool of one window of the tile: rows 4wy .. 4wy+5, columns 4wx .. 4wx+5
__device__ __forceinline__ void rnd_tile_window(RndTileShared &s, int wy, int wx, bool inside) {
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f, m5 = 0.f;
    if (inside) {
        uint32_t seg[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) seg[r] = row_seg6(s.rows[4 * wy + r][0], s.rows[4 * wy + r][1], 4 * wx);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            float4 pa[4];
            float2 pb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int dy = b, dx = j;
                const uint32_t p = ((seg[dy] >> dx) & 7u) | (((seg[dy + 1] >> dx) & 7u) << 3) |
                                   (((seg[dy + 2] >> dx) & 7u) << 6);
                pa[j] = s.net.lut_p[p];
                pb[j] = s.net.lut_r[p];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                m0 = fmaxf(m0, pa[j].x); m1 = fmaxf(m1, pa[j].y); m2 = fmaxf(m2, pa[j].z); m3 = fmaxf(m3, pa[j].w);
                m4 = fmaxf(m4, pb[j].x); m5 = fmaxf(m5, pb[j].y);
            }
        }
    }
    const int off = wy * kPlaneStride + wx;
    s.net.planes[0][off] = m0;
    s.net.planes[1][off] = m1;
    s.net.planes[2][off] = m2;
    s.net.planes[3][off] = m3;
    s.net.planes[4][off] = m4;
    s.net.planes[5][off] = m5;
}

struct __attribute__((aligned(4))) F4T { float x, y, z, w; };   // 16 bytes at 4-byte alignment (weight rows)

// sum over the 64 lanes by DPP (no LDS crossbar round trips: a chain of six ds_bpermute per value made the
// 164-value row reduction at the end of the training kernels cost more than a whole tile); valid in every lane
__device__ __forceinline__ float tile_wave_sum(float v) {
    v += dpp_quad_xor1(v);
    v += dpp_quad_xor2(v);
    v += dpp_row_ror(v, 0x124);
    v += dpp_row_ror(v, 0x128);
    const int b = __builtin_bit_cast(int, v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48)));
}

// Sums over the 64 lanes of MANY values at once (the gradient rows at the end of the training kernels): a butterfly
// reduce-scatter -- at every stage a lane keeps one half of its values, adds its partner's copies of that half and gives
// the other half away -- so N0 values cost 3 N0 (1/2 + 1/4 + ...) = ~3 N0 instructions per wave instead of ~12 N0 with one
// tile_wave_sum per value (AE2D: 164 values, ~2 000 instructions per wave, two waves per SIMD: several us per launch).
// Partners: lane ^ 7 (row_half_mirror), ^ 1, ^ 2 (quad_perm), ^ 8 (row_ror:8), ^ 16, ^ 32 (ds_bpermute); a stage's
// partners hold the same range because they agree in the lane bits earlier stages selected by (2, 0, 1, 3, 4, 5).
// get(integral_constant<int, i>) -> value i (zero beyond the last); the lane ends with the N0 / 64 sums of indices
// lo .. lo + N0 / 64 - 1 in `out`.  Fixed summation tree: deterministic.
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ float dpp_half_mirror(float x) {     // value of lane ^ 7
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));
}

template <int N0, class Get>
__device__ __forceinline__ int wave_reduce_scatter(Get get, int lane, float (&out)[N0 / 64]) {
    static_assert(N0 % 64 == 0, "64 lanes share the values evenly");
    constexpr int H0 = N0 / 2, H1 = N0 / 4, H2 = N0 / 8, H3 = N0 / 16, H4 = N0 / 32, H5 = N0 / 64;
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8, b4 = lane & 16, b5 = lane & 32;
    float k0[H0], k1[H1], k2[H2], k3[H3], k4[H4];
    static_for<H0>([&](auto i) {
        const float a = get(i), b = get(std::integral_constant<int, decltype(i)::value + H0>{});
        k0[i] = (b2 ? b : a) + dpp_half_mirror(b2 ? a : b);
    });
    static_for<H1>([&](auto i) { k1[i] = (b0 ? k0[i + H1] : k0[i]) + dpp_quad_xor1(b0 ? k0[i] : k0[i + H1]); });
    static_for<H2>([&](auto i) { k2[i] = (b1 ? k1[i + H2] : k1[i]) + dpp_quad_xor2(b1 ? k1[i] : k1[i + H2]); });
    static_for<H3>([&](auto i) { k3[i] = (b3 ? k2[i + H3] : k2[i]) + dpp_row_ror(b3 ? k2[i] : k2[i + H3], 0x128); });
    static_for<H4>([&](auto i) { k4[i] = (b4 ? k3[i + H4] : k3[i]) + __shfl_xor(b4 ? k3[i] : k3[i + H4], 16, 64); });
    static_for<H5>([&](auto i) { out[i] = (b5 ? k4[i + H5] : k4[i]) + __shfl_xor(b5 ? k4[i] : k4[i + H5], 32, 64); });
    return (b2 ? H0 : 0) + (b0 ? H1 : 0) + (b1 ? H2 : 0) + (b3 ? H3 : 0) + (b4 ? H4 : 0) + (b5 ? H5 : 0);
}

// the finish of one grid by the 16 lanes of a DPP row: lane k adds the tiles' partial sums of output k of both nets in tile
// order, tanh, squared difference, mean over the 16 outputs; bonus, outputs and reward += scale * bonus (mcl.py:343-352, 403-405)
template <bool COHERENT = false>
__device__ __forceinline__ void rnd_finish_grid(const float *__restrict__ partial, const float *__restrict__ pp,
                                                const float *__restrict__ rp, float *__restrict__ bonus,
                                                float *__restrict__ pred_out, float *__restrict__ target_out,
                                                float *__restrict__ reward, float reward_scale, int64_t g, bool valid, int k,
                                                int tiles, int dfeat) {
    // no contraction: whether (r - p)^2 and the first DPP add become one fma depends on the code around this function, and
    // its users (finish kernel, finish by the last tile; pair_finish_grid repeats it) must agree to the bit
#pragma clang fp contract(off)
    float sq = 0.0f;
    if (valid) {
        float p = 0.0f, r = 0.0f;
        // COHERENT: the sums were stored by workgroups on other XCDs (finish by the last tile): agent-scope loads, which
        // read past this XCD's L2 (a fence would invalidate all of it under the other workgroups' feet)
        auto ld = [&](const float *q) {
            return COHERENT ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q;
        };
        // sixteen tiles' sums requested at once, added in tile order (one round trip per sixteen tiles, not per tile)
        for (int t0 = 0; t0 < tiles; t0 += 16) {
            float a[16], b[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int t = t0 + i < tiles ? t0 + i : tiles - 1;
                a[i] = ld(partial + ((g * tiles + t) * 2 + 0) * 16 + k);
                b[i] = ld(partial + ((g * tiles + t) * 2 + 1) * 16 + k);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (t0 + i < tiles) { p += a[i]; r += b[i]; }
        }
        p = tanhf(p + pp[77 + (int64_t)16 * dfeat + k]);
        r = tanhf(r + rp[39 + (int64_t)16 * dfeat + k]);
        if (pred_out) pred_out[g * 16 + k] = p;
        if (target_out) target_out[g * 16 + k] = r;
        sq = (r - p) * (r - p);
    }
    sq += dpp_quad_xor1(sq);
    sq += dpp_quad_xor2(sq);
    sq += dpp_row_ror(sq, 0x124);
    sq += dpp_row_ror(sq, 0x128);
    if (valid && k == 0) {
        const float b = sq * (1.0f / 16.0f);
        bonus[g] = b;
        // mcl.py:403-405, as torch evaluates it: product and sum rounded separately (no contraction into an fma -- also
        // what keeps the paired launch, pair_finish_grid, bit-identical to the two wrappers one after the other)
        if (reward) reward[g] = reward_accumulate(reward[g], reward_scale, b);
    }
}

// Finish WITHOUT a finish launch: the workgroup that delivers the LAST tile of a grid (a counter per grid, agent scope)
// finishes that grid itself -- the same sums in the same (tile) order, whoever comes last.  counters: n ints, zero before
// the launch and zero again behind it.
struct RndTileFinish {
    int32_t *counters = nullptr;
    float *bonus = nullptr, *pred = nullptr, *targ = nullptr, *reward = nullptr;
    float reward_scale = 0.0f;
    // fp32 observations, training: every tile also leaves its own 64 x 64 cells bit-packed (H rows of W/64 words) --
    // the kernels behind this one (tile gradients, AE2D) then read 512 bytes per tile instead of 22 KB of fp32
    uint64_t *bits_out = nullptr;
};

// The forward pair AE2D(RND2D(env)) without a finish launch: both nets' tile workgroups count their arrivals per grid;
// the workgroup that delivers the LAST of a grid's 2 x tiles tiles forms both bonuses and the reward (the arithmetic of the
// two finish kernels, agent-scope loads of the partial sums) and keeps (sum, sum of squares) of the rewards it formed for
// the reward moments (one pair per workgroup of the launch, registered in the status block by workgroup 0).
struct PairFinish {
    int32_t *counters = nullptr;                     // n ints, zero before the launch and zero again behind it
    const float *rpartial = nullptr, *apartial = nullptr, *pp = nullptr, *rp = nullptr;
    float *rbonus = nullptr, *abonus = nullptr, *reward = nullptr;
    float rscale = 0.0f, ascale = 0.0f, inv_pixels = 0.0f;
    int dfeat = 0;
    double *mom_partials = nullptr;
    int32_t *status = nullptr;
};

// lanes 0..15 of wave 0; returns the final reward of grid g in lane 0
__device__ __forceinline__ float pair_finish_grid(const PairFinish &f, int64_t g, int k, int tiles) {
#pragma clang fp contract(off)                          // as rnd_finish_grid: bit-identical
    auto ld = [](const float *q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    float p = 0.0f, r = 0.0f, ae = 0.0f;
    for (int t0 = 0; t0 < tiles; t0 += 16) {
        float a[16], b[16], c[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int t = t0 + i < tiles ? t0 + i : tiles - 1;
            a[i] = ld(f.rpartial + ((g * tiles + t) * 2 + 0) * 16 + k);
            b[i] = ld(f.rpartial + ((g * tiles + t) * 2 + 1) * 16 + k);
            c[i] = k == 0 ? ld(f.apartial + g * tiles + t) : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (t0 + i < tiles) { p += a[i]; r += b[i]; if (k == 0) ae += c[i]; }
    }
    p = tanhf(p + f.pp[77 + (int64_t)16 * f.dfeat + k]);
    r = tanhf(r + f.rp[39 + (int64_t)16 * f.dfeat + k]);
    float sq = (r - p) * (r - p);
    sq += dpp_quad_xor1(sq);
    sq += dpp_quad_xor2(sq);
    sq += dpp_row_ror(sq, 0x124);
    sq += dpp_row_ror(sq, 0x128);
    float x = 0.0f;
    if (k == 0) {
        const float br = sq * (1.0f / 16.0f), ba = ae * f.inv_pixels;
        f.rbonus[g] = br;
        f.abonus[g] = ba;
        if (f.reward) {
            x = f.reward[g];
            x = reward_accumulate(x, f.rscale, br);          // RND2D.step, then AE2D.step (mcl.py:403-405)
            x = reward_accumulate(x, f.ascale, ba);
            f.reward[g] = x;
        }
    }
    return x;
}

// behind a tile of grid g whose partial sums went out as agent-scope stores: ticket, and the grid's finish by the last
// arrival.  `last` = a word of LDS; `acc` (thread 0) gathers the rewards this workgroup formed.  Every thread calls.
__device__ __forceinline__ void pair_ticket(const PairFinish &f, int &last, int64_t g, int tiles, int tid, double2 &acc) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
        last = __hip_atomic_fetch_add(f.counters + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2 * tiles - 1;
    __syncthreads();
    if (last) {                                          // workgroup-uniform
        if (tid < 16) {
            const float r = pair_finish_grid(f, g, tid, tiles);
            if (tid == 0 && f.reward) { acc.x += (double)r; acc.y = fma((double)r, (double)r, acc.y); }
        }
        if (tid == 0) __hip_atomic_store(f.counters + g, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// at the end of a workgroup of the pair launch: its pair of reward sums, and the registration by workgroup 0
__device__ __forceinline__ void pair_leave_moments(const PairFinish &f, int tid, const double2 &acc, int64_t n) {
    if (tid != 0 || !f.mom_partials || !f.status || !f.reward) return;
    reinterpret_cast<double2 *>(f.mom_partials)[blockIdx.x] = acc;
    if (blockIdx.x == 0) {
        uint4 w5;
        w5.x = (uint32_t)reinterpret_cast<uint64_t>(f.mom_partials);
        w5.y = (uint32_t)(reinterpret_cast<uint64_t>(f.mom_partials) >> 32);
        w5.z = gridDim.x;
        w5.w = (uint32_t)n;
        reinterpret_cast<uint4 *>(f.status)[5] = w5;                  // CARLE_ST_MOM_PARTIALS / _COUNT / _N
    }
}

// PACKED: obs = bit-packed grids (H rows of W/64 words); else fp32 (N,1,H,W), any non-zero number = alive
// (body as a device function: the stand-alone kernel below and the launch that runs the RND2D and AE2D tiles of a
// wrapper stack side by side, rnd_ae_tile_kernel, call it with their own workgroup index / count)
template <bool PACKED>
__device__ __forceinline__ void rnd_tile_body(RndTileShared &s, const void *__restrict__ obs_v, const float *__restrict__ ws,
                                              const float *__restrict__ pp, const float *__restrict__ rp,
                                              float *__restrict__ partial, float *__restrict__ featbuf,
                                              int64_t n, int size, unsigned bid, unsigned nblk,
                                              const RndTileFinish fin = RndTileFinish(), const PairFinish pf = PairFinish()) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double2 mom_acc = make_double2(0.0, 0.0);           // PairFinish: rewards this workgroup formed (thread 0)
    rnd_block_init(s.net, ws, tid);
    const int tps = size >> 6, tiles = tps * tps;      // tiles per side / per grid
    const int d8 = size >> 3, dfeat = d8 * d8;          // feature columns per row / per net
    const int64_t total = n * tiles;
    // loads of one tile into registers (fp32: the wave's rows, 2 values per lane and row; packed: thread r < kTileRows
    // assembles row r's two words): issued a whole iteration ahead
    constexpr int kPer = (kTileRows + 3) / 4;
    uint32_t va[kPer], vb[kPer];
    uint64_t plo = 0ull, phi = 0ull;
    auto issue = [&](int64_t t) {
        const int64_t g = t / tiles;
        const int tt = (int)(t % tiles), ty = tt / tps, tx = tt % tps;
        const int y0 = 64 * ty - 5, x0 = 64 * tx - 5;
        if (PACKED) {
            plo = 0ull; phi = 0ull;
            const int y = y0 + tid;
            if (tid < kTileRows && y >= 0 && y < size) {
                const uint64_t *row = reinterpret_cast<const uint64_t *>(obs_v) + (g * size + y) * tps;
                const uint64_t wml = row[tx > 0 ? tx - 1 : tx], w0 = row[tx], wpl = row[tx + 1 < tps ? tx + 1 : tx];   // unconditional: no wait per load
                    const uint64_t wm = tx > 0 ? wml : 0ull, wp = tx + 1 < tps ? wpl : 0ull;
                plo = (wm >> 59) | (w0 << 5);
                phi = (w0 >> 59) | ((wp & 0x1Full) << 5);
            }
        } else {
            // UNCONDITIONAL loads of clamped addresses, validity applied in the ballot below (as rnd_tile_bwd_kernel): the
            // conditional form cost ~14 instructions and a branch per load (38 loads per wave at the head of the tile)
            const float *gin = reinterpret_cast<const float *>(obs_v) + g * (int64_t)size * size;
            const int xac = min(max(x0 + lane, 0), size - 1), xbc = min(x0 + 64 + min(lane, 9), size - 1);
#pragma unroll
            for (int i = 0; i < kPer; ++i) {
                const int yc = min(max(y0 + wv + 4 * i, 0), size - 1);
                va[i] = __float_as_uint(gin[(int64_t)yc * size + xac]);
                vb[i] = __float_as_uint(gin[(int64_t)yc * size + xbc]);
            }
        }
    };
    if ((int64_t)bid < total) issue(bid);
    __syncthreads();
    for (int64_t t = bid; t < total; t += nblk) {
        const int64_t g = t / tiles;
        const int tt = (int)(t % tiles), ty = tt / tps, tx = tt % tps;
        // ---- the tile's kTileRows rows as bit words: from the registers the previous iteration (or the prologue) filled
        if (PACKED) {
            if (tid < kTileRows) { s.rows[tid][0] = plo; s.rows[tid][1] = phi; }
        } else {
            const int y0 = 64 * ty - 5, x0 = 64 * tx - 5;
            const int xa = x0 + lane, xb = x0 + 64 + lane;
#pragma unroll
            for (int i = 0; i < kPer; ++i) {
                const int r = wv + 4 * i, y = y0 + r;
                const bool yin = r < kTileRows && y >= 0 && y < size;
                const uint64_t lo = __ballot(yin && xa >= 0 && xa < size && (va[i] & 0x7FFFFFFFu) != 0u);
                const uint64_t hi = __ballot(yin && lane < 10 && xb < size && (vb[i] & 0x7FFFFFFFu) != 0u);
                if (lane == 0 && r < kTileRows) { s.rows[r][0] = lo; s.rows[r][1] = hi; }
            }
        }
        if (t + nblk < total) issue(t + nblk);          // the next tile travels during this one's layers
        // this tile's 8 weights per thread of the linear layers: requested now, used after conv2
        const int lfy = tid & 7, lk = (tid >> 3) & 15, lnet = tid >> 7;
        const float *wrow = (lnet == 0 ? pp + 77 : rp + 39) + (int64_t)lk * dfeat + (int64_t)(8 * ty + lfy) * d8 + 8 * tx;
        const F4T wa = *reinterpret_cast<const F4T *>(wrow), wb = *reinterpret_cast<const F4T *>(wrow + 4);
        __syncthreads();
        if (!PACKED && fin.bits_out && tid < 64)        // the tile's own cells: bits 5 .. 68 of rows 5 .. 68 of the window
            fin.bits_out[(g * size + 64 * ty + tid) * tps + tx] = (s.rows[tid + 5][0] >> 5) | (s.rows[tid + 5][1] << 59);
        // ---- conv1 + pool of the 18x18 windows (window (wy, wx) = pooled position (16ty - 1 + wy, 16tx - 1 + wx))
        {
            const int q4 = size >> 2;
            for (int i = tid; i < kTileWin * kTileWin; i += 256) {
                const int wy = i / kTileWin, wx = i % kTileWin;
                const int Y = 16 * ty - 1 + wy, X = 16 * tx - 1 + wx;
                rnd_tile_window(s, wy, wx, Y >= 0 && Y < q4 && X >= 0 && X < q4);
            }
        }
        __syncthreads();
        rnd_block_conv2(s.net, s.feat, tid);            // the tile's 8x8 features of both nets
        __syncthreads();
        if (featbuf && tid < 128) {                     // training: the features, (N, 2, D) as the staged path keeps them
            const int net = tid >> 6, f = tid & 63;
            featbuf[(g * 2 + net) * (int64_t)dfeat + (int64_t)(8 * ty + (f >> 3)) * d8 + 8 * tx + (f & 7)] = s.feat[net][f];
        }
        // ---- the tile's share of the linear layers: thread = (net, k, feature row fy of the tile)
        {
            const int fy = lfy, k = lk, net = lnet;
            const F4T a = wa, b = wb;
            const float *f = &s.feat[net][8 * fy];
            float acc = a.x * f[0];
            acc = fmaf(a.y, f[1], acc); acc = fmaf(a.z, f[2], acc); acc = fmaf(a.w, f[3], acc);
            acc = fmaf(b.x, f[4], acc); acc = fmaf(b.y, f[5], acc); acc = fmaf(b.z, f[6], acc); acc = fmaf(b.w, f[7], acc);
            acc += dpp_quad_xor1(acc);
            acc += dpp_quad_xor2(acc);
            acc += __shfl_xor(acc, 4, 64);
            if (fy == 0) {
                if (fin.counters || pf.counters)
                    __hip_atomic_store(partial + (t * 2 + net) * 16 + k, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else partial[(t * 2 + net) * 16 + k] = acc;
            }
        }
        if (pf.counters) pair_ticket(pf, s.last, g, tiles, tid, mom_acc);     // kernel argument: workgroup-uniform
        if (fin.counters) {                             // kernel argument: the same for every workgroup
            // The tile's partial sums went out as agent-scope stores (written through to where every XCD reads them);
            // once they are acknowledged the workgroup takes its ticket.  No fences: an agent-scope fence writes back and
            // invalidates the XCD's WHOLE L2 -- 25 us per step at 16 x 256x256 with 256 workgroups doing it.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0)
                s.last = __hip_atomic_fetch_add(fin.counters + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tiles - 1;
            __syncthreads();
            if (s.last) {                               // workgroup-uniform
                if (tid < 16)
                    rnd_finish_grid<true>(partial, pp, rp, fin.bonus, fin.pred, fin.targ, fin.reward, fin.reward_scale, g, true,
                                          tid, tiles, dfeat);
                if (tid == 0) __hip_atomic_store(fin.counters + g, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // rows / planes / feat are rewritten behind the next iteration's barriers
    }
    if (pf.counters) pair_leave_moments(pf, tid, mom_acc, n);
}

template <bool PACKED>
__global__ __launch_bounds__(256, 4) void rnd_tile_kernel(const void *__restrict__ obs_v, const float *__restrict__ ws,
                                                          const float *__restrict__ pp, const float *__restrict__ rp,
                                                          float *__restrict__ partial, float *__restrict__ featbuf,
                                                          int64_t n, int size, const RndTileFinish fin = RndTileFinish()) {
    __shared__ RndTileShared s;
    rnd_tile_body<PACKED>(s, obs_v, ws, pp, rp, partial, featbuf, n, size, blockIdx.x, gridDim.x, fin);
}

// 16 threads per grid: thread k adds the tiles' partial sums of output k of both nets in tile order
__global__ __launch_bounds__(256) void rnd_tile_finish_kernel(const float *__restrict__ partial, const float *__restrict__ pp,
                                                              const float *__restrict__ rp, float *__restrict__ bonus,
                                                              float *__restrict__ pred_out, float *__restrict__ target_out,
                                                              float *__restrict__ reward, float reward_scale, int64_t n,
                                                              int tiles, int dfeat) {
    const int64_t g = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    rnd_finish_grid(partial, pp, rp, bonus, pred_out, target_out, reward, reward_scale, g, g < n, threadIdx.x & 15, tiles, dfeat);
}

// ---------------------------------------------------------------------------------------------
// RND2D predictor GRADIENT, tiled (carle/mcl.py:343-386; eval mode): conv1 / conv2 weight and bias gradients of
//   gs/2 * sum_n sum_k (target - prediction)^2
// by the decomposition of the staged backward (nets_staged_bwd.hpp) on LDS tiles.  A tile owns the gradient paths
// that START at its own 8x8 features: it recomputes its activations with their halos (as the forward kernel), so
// no gradient crosses a tile border, and the sum over tiles is the whole gradient.  The linear layer's gradient
// comes from the features the forward kernel stored (rnd_bwd_linear_kernel of the staged path).  The arg-max of a
// 4x4 pooling window is the first maximum in scan order among its 16 table values (ties: equal neighbourhood
// patterns or the relu floor -- the routing is immaterial, as in the fused 64x64 kernel).  One partial row of 77
// sums per workgroup; staged_grad_reduce_kernel adds the rows in a fixed order.
// ---------------------------------------------------------------------------------------------
struct RndTileBwdShared {
    RndTileShared f;
    uint16_t pat[4][kTileWin * kTileWin];              // arg-max neighbourhood pattern per window and predictor channel
    float dmap[20][20];                                // d loss / d conv2 output (tile-local position + 2), zero ring
    float red[4][80];
};

// (body as a device function: the stand-alone kernel and rnd_bwd_pair_kernel -- the tile gradients and the linear layer's
// gradient of a training step side by side in one launch -- call it with their own workgroup index / count)
template <bool PACKED>
__device__ __forceinline__ void rnd_tile_bwd_body(RndTileBwdShared &s, const void *__restrict__ obs_v, const float *__restrict__ ws,
                                                  const float *__restrict__ pp, const float *__restrict__ pred,
                                                  const float *__restrict__ targ, float *__restrict__ rows_out,
                                                  float gs, int64_t n, int size, const int bidx, const int bcount) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    rnd_block_init(s.f.net, ws, tid);
    for (int i = tid; i < 400; i += 256) (&s.dmap[0][0])[i] = 0.0f;
    const int tps = size >> 6, tiles = tps * tps;
    const int d8 = size >> 3, dfeat = d8 * d8, q4 = size >> 2;
    const int64_t total = n * tiles;
    float acc1[36], accb1[4] = {0.f, 0.f, 0.f, 0.f}, acc2[9], accb2 = 0.0f;
#pragma unroll
    for (int i = 0; i < 36; ++i) acc1[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 9; ++i) acc2[i] = 0.0f;
    __syncthreads();
    for (int64_t t = bidx; t < total; t += bcount) {
        const int64_t g = t / tiles;
        const int tt = (int)(t % tiles), ty = tt / tps, tx = tt % tps;
        const int y0 = 64 * ty - 5, x0 = 64 * tx - 5;
        // ---- rows (as the forward kernel, not pipelined)
        if (PACKED) {
            if (tid < kTileRows) {
                const int y = y0 + tid;
                uint64_t lo = 0ull, hi = 0ull;
                if (y >= 0 && y < size) {
                    const uint64_t *row = reinterpret_cast<const uint64_t *>(obs_v) + (g * size + y) * tps;
                    const uint64_t wml = row[tx > 0 ? tx - 1 : tx], w0 = row[tx], wpl = row[tx + 1 < tps ? tx + 1 : tx];   // unconditional: no wait per load
                    const uint64_t wm = tx > 0 ? wml : 0ull, wp = tx + 1 < tps ? wpl : 0ull;
                    lo = (wm >> 59) | (w0 << 5);
                    hi = (w0 >> 59) | ((wp & 0x1Full) << 5);
                }
                s.f.rows[tid][0] = lo;
                s.f.rows[tid][1] = hi;
            }
        } else {
            const float *gin = reinterpret_cast<const float *>(obs_v) + g * (int64_t)size * size;
            constexpr int kPer = (kTileRows + 3) / 4;
            uint32_t va[kPer], vb[kPer];
            // UNCONDITIONAL loads of clamped addresses, validity applied in the ballot: behind a load inside an `if`
            // whose only use follows directly, the compiler folds the other path, sinks the compare into the branch
            // and waits there -- one dependent memory round trip per row (19 per tile; seen in the disassembly)
            const int xa = x0 + lane, xb = x0 + 64 + lane;
            const int xac = min(max(xa, 0), size - 1), xbc = min(x0 + 64 + min(lane, 9), size - 1);
#pragma unroll
            for (int i = 0; i < kPer; ++i) {
                const int yc = min(max(y0 + wv + 4 * i, 0), size - 1);
                va[i] = __float_as_uint(gin[(int64_t)yc * size + xac]);
                vb[i] = __float_as_uint(gin[(int64_t)yc * size + xbc]);
            }
#pragma unroll
            for (int i = 0; i < kPer; ++i) {
                const int r = wv + 4 * i, y = y0 + r;
                const bool yin = r < kTileRows && y >= 0 && y < size;
                const uint64_t lo = __ballot(yin && xa >= 0 && xa < size && (va[i] & 0x7FFFFFFFu) != 0u);
                const uint64_t hi = __ballot(yin && lane < 10 && xb < size && (vb[i] & 0x7FFFFFFFu) != 0u);
                if (lane == 0 && r < kTileRows) { s.f.rows[r][0] = lo; s.f.rows[r][1] = hi; }
            }
        }
        __syncthreads();
        // ---- conv1 (predictor table) + relu + 4x4 pool with the arg-max pattern of every channel
        for (int i = tid; i < kTileWin * kTileWin; i += 256) {
            const int wy = i / kTileWin, wx = i % kTileWin;
            const int Y = 16 * ty - 1 + wy, X = 16 * tx - 1 + wx;
            float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
            uint32_t p0 = 0, p1 = 0, p2 = 0, p3 = 0;
            if (Y >= 0 && Y < q4 && X >= 0 && X < q4) {
                uint32_t seg[6];
#pragma unroll
                for (int r = 0; r < 6; ++r) seg[r] = row_seg6(s.f.rows[4 * wy + r][0], s.f.rows[4 * wy + r][1], 4 * wx);
#pragma unroll
                for (int dy = 0; dy < 4; ++dy) {
                    float4 e[4];
                    uint32_t pt[4];
#pragma unroll
                    for (int dx = 0; dx < 4; ++dx) {
                        pt[dx] = ((seg[dy] >> dx) & 7u) | (((seg[dy + 1] >> dx) & 7u) << 3) | (((seg[dy + 2] >> dx) & 7u) << 6);
                        e[dx] = s.f.net.lut_p[pt[dx]];
                    }
#pragma unroll
                    for (int dx = 0; dx < 4; ++dx) {
                        if (e[dx].x > m0) { m0 = e[dx].x; p0 = pt[dx]; }
                        if (e[dx].y > m1) { m1 = e[dx].y; p1 = pt[dx]; }
                        if (e[dx].z > m2) { m2 = e[dx].z; p2 = pt[dx]; }
                        if (e[dx].w > m3) { m3 = e[dx].w; p3 = pt[dx]; }
                    }
                }
            }
            const int off = wy * kPlaneStride + wx;
            s.f.net.planes[0][off] = m0; s.f.net.planes[1][off] = m1; s.f.net.planes[2][off] = m2; s.f.net.planes[3][off] = m3;
            s.pat[0][i] = (uint16_t)p0; s.pat[1][i] = (uint16_t)p1; s.pat[2][i] = (uint16_t)p2; s.pat[3][i] = (uint16_t)p3;
        }
        __syncthreads();
        // ---- conv2 forward (predictor), its gradient and weight gradients: thread = (feature f, channel c)
        {
            const int f = tid >> 2, c = tid & 3, fy = f >> 3, fx = f & 7;
            const float *pl = &s.f.net.planes[c][(2 * fy) * kPlaneStride + 2 * fx];
            float patch[4][4], w[9];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float2 lo = *reinterpret_cast<const float2 *>(pl + r * kPlaneStride);
                const float2 hi = *reinterpret_cast<const float2 *>(pl + r * kPlaneStride + 2);
                patch[r][0] = lo.x; patch[r][1] = lo.y; patch[r][2] = hi.x; patch[r][3] = hi.y;
            }
#pragma unroll
            for (int i = 0; i < 9; ++i) w[i] = s.f.net.w2[0][c][i];
            float o[2][2];
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    float a = 0.f;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) a = fmaf(w[ky * 3 + kx], patch[dy + ky][dx + kx], a);
                    a += dpp_quad_xor1(a);
                    a += dpp_quad_xor2(a);
                    o[dy][dx] = a;
                }
            int sel = 0;
            float mx = o[0][0];
            if (o[0][1] > mx) { mx = o[0][1]; sel = 1; }
            if (o[1][0] > mx) { mx = o[1][0]; sel = 2; }
            if (o[1][1] > mx) { mx = o[1][1]; sel = 3; }
            const float featv = mx + s.f.net.b2[0];
            // d loss / d feature through the linear layer: lane c takes outputs 4c .. 4c+3
            const int64_t col = (int64_t)(8 * ty + fy) * d8 + 8 * tx + fx;
            float df = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 4 * c + j;
                const float pk = pred[g * 16 + k], tk = targ[g * 16 + k];
                const float dz = gs * (pk - tk) * (1.0f - pk * pk);
                df = fmaf(pp[77 + (int64_t)k * dfeat + col], dz, df);
            }
            df += dpp_quad_xor1(df);
            df += dpp_quad_xor2(df);
            const float dd = featv > 0.0f ? df : 0.0f;
            const int ay = sel >> 1, ax = sel & 1;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float p00 = patch[ky][kx], p01 = patch[ky][kx + 1], p10 = patch[ky + 1][kx], p11 = patch[ky + 1][kx + 1];
                    acc2[ky * 3 + kx] = fmaf(dd, ay ? (ax ? p11 : p10) : (ax ? p01 : p00), acc2[ky * 3 + kx]);
                }
            if (c == 0) {
                accb2 += dd;
#pragma unroll
                for (int e = 0; e < 4; ++e) s.dmap[2 * fy + (e >> 1) + 2][2 * fx + (e & 1) + 2] = e == sel ? dd : 0.0f;
            }
        }
        __syncthreads();
        // ---- conv1 weight / bias gradients: thread = window; the gradient of the pooled value is gathered from dmap
        for (int i = tid; i < kTileWin * kTileWin; i += 256) {
            const int wy = i / kTileWin, wx = i % kTileWin;
            float dm[3][3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) dm[ky][kx] = s.dmap[wy - ky + 2][wx - kx + 2];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float dp = 0.0f;
#pragma unroll
                for (int tq = 0; tq < 9; ++tq) dp = fmaf(s.f.net.w2[0][c][tq], dm[tq / 3][tq % 3], dp);
                const float g1 = s.f.net.planes[c][wy * kPlaneStride + wx] > 0.0f ? dp : 0.0f;
                const uint32_t pt = s.pat[c][i];
                accb1[c] += g1;
#pragma unroll
                for (int tq = 0; tq < 9; ++tq) acc1[c * 9 + tq] = fmaf(g1, (float)((pt >> tq) & 1u), acc1[c * 9 + tq]);
            }
        }
        // the next tile's barriers order its writes to rows / planes / pat / dmap behind these reads
    }
    // ---- one partial row per workgroup: conv1.w[36] conv1.b[4] conv2.w[36] conv2.b[1]
    __syncthreads();
    const int cme = tid & 3;
    {
        auto get = [&](auto ic) -> float {
            constexpr int i = decltype(ic)::value;
            if constexpr (i < 36) return acc1[i];
            else if constexpr (i < 40) return accb1[i - 36];
            else if constexpr (i < 76) return (i - 40) / 9 == cme ? acc2[(i - 40) % 9] : 0.0f;
            else if constexpr (i == 76) return accb2;
            else return 0.0f;
        };
        float sums[2];
        const int lo = wave_reduce_scatter<128>(get, lane, sums);      // (one tile_wave_sum per value: 2.4 x the instructions)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (lo + j < 77) s.red[wv][lo + j] = sums[j];
    }
    __syncthreads();
    if (tid < 77) rows_out[(int64_t)bidx * 77 + tid] = (s.red[0][tid] + s.red[1][tid]) + (s.red[2][tid] + s.red[3][tid]);
}

template <bool PACKED>
__global__ __launch_bounds__(256, 2) void rnd_tile_bwd_kernel(const void *__restrict__ obs_v, const float *__restrict__ ws,
                                                              const float *__restrict__ pp, const float *__restrict__ pred,
                                                              const float *__restrict__ targ, float *__restrict__ rows_out,
                                                              float gs, int64_t n, int size) {
    __shared__ RndTileBwdShared s;
    rnd_tile_bwd_body<PACKED>(s, obs_v, ws, pp, pred, targ, rows_out, gs, n, size, (int)blockIdx.x, (int)gridDim.x);
}

// workgroups [0, brows): the tile gradients; [brows, brows + lin_blocks): the linear layer's gradient from the stored
// features (nets_staged_bwd.hpp) -- independent of each other, both behind rnd_tile_finish_kernel
template <bool PACKED>
__global__ __launch_bounds__(256, 2) void rnd_bwd_pair_kernel(const void *__restrict__ obs_v, const float *__restrict__ ws,
                                                              const float *__restrict__ pp, const float *__restrict__ pred,
                                                              const float *__restrict__ targ, float *__restrict__ rows_out,
                                                              const float *__restrict__ featbuf, float *__restrict__ grad,
                                                              float gs, int64_t n, int size, int brows) {
    __shared__ RndTileBwdShared s;
    if ((int)blockIdx.x < brows)
        rnd_tile_bwd_body<PACKED>(s, obs_v, ws, pp, pred, targ, rows_out, gs, n, size, (int)blockIdx.x, brows);
    else
        rnd_bwd_linear_body(featbuf, pred, targ, grad, gs, n, (size >> 3) * (size >> 3), (int64_t)blockIdx.x - brows);
}

// floats of the RND2D training scratch in front of the bit-packed copy of fp32 observations (rnd_ae_tiled_train): the
// layout of the staged path's scratch (the larger of the two), rounded to an even count (8-byte words behind it)
int64_t rnd2d_tiled_bits_offset_floats(int64_t n, int size) {
    const int64_t d = (int64_t)(size >> 3) * (size >> 3);
    const int64_t staged = 2 * n * 16 + 2 * n * d + 1024 * (37 + 40) + 64;      // = rnd2d_train_staged_scratch_floats
    return (staged + 1) & ~1ll;
}

// number of partial rows (= workgroups) of the tiled backward kernel
int rnd2d_tiled_bwd_rows(int64_t n, int size) {
    const int64_t total = n * (int64_t)(size >> 6) * (size >> 6);
    return (int)(total < 512 ? total : 512);
}

// forward with the features kept, outputs -> pred / targ, then the conv gradients -> rows (the caller runs the
// linear-layer gradient and the row reduction of the staged path)
int rnd2d_tiled_train(const void *obs, bool packed, int64_t n, int size, const float *pp, const float *rp, float *ws,
                      float *featbuf, float *pred, float *targ, float *rows, float *bonus, float *reward,
                      float reward_scale, float grad_scale, bool tables_valid, hipStream_t s) {
    if (!tables_valid) rnd2d_prepare_tables(pp, rp, ws, s);
    const int tiles = (size >> 6) * (size >> 6);
    const int64_t total = n * tiles;
    if ((n + 15) / 16 > 0x7FFFFFFFll) return fail_arg(CARLE_E_DIMS, "carle_rnd2d_train: batch too large");
    const unsigned blocks = (unsigned)(total < 1024 ? total : 1024);
    float *partial = ws + kFusedWsTotal;
    const int d8 = size >> 3;
    if (packed)
        hipLaunchKernelGGL(rnd_tile_kernel<true>, dim3(blocks), dim3(256), 0, s, obs, ws, pp, rp, partial, featbuf, n, size);
    else
        hipLaunchKernelGGL(rnd_tile_kernel<false>, dim3(blocks), dim3(256), 0, s, obs, ws, pp, rp, partial, featbuf, n, size);
    hipLaunchKernelGGL(rnd_tile_finish_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, s, partial, pp, rp, bonus,
                       pred, targ, reward, reward_scale, n, tiles, d8 * d8);
    const unsigned brows = (unsigned)rnd2d_tiled_bwd_rows(n, size);
    if (packed)
        hipLaunchKernelGGL(rnd_tile_bwd_kernel<true>, dim3(brows), dim3(256), 0, s, obs, ws, pp, pred, targ, rows,
                           grad_scale, n, size);
    else
        hipLaunchKernelGGL(rnd_tile_bwd_kernel<false>, dim3(brows), dim3(256), 0, s, obs, ws, pp, pred, targ, rows,
                           grad_scale, n, size);
    return check_launch("carle_rnd2d_train(tiled)");
}

// =============================================================================================
// AE2D forward, tiled (carle/mcl.py:452-467, 504-512; eval mode): the 64x64 kernel of nets_fused.hip with the
// halos its four layers need -- cells +7, A1 (pooled conv1) +3, A2 (pooled conv2) +1, A3 (convT1) +1 -- every
// halo value computed from the tile's own cells (27-41 % more conv work than the interior alone), values
// outside the grid zero (each layer's padding).  One partial sum of squared errors per tile; the staged
// path's finish kernel adds a grid's tiles in a fixed order.
// =============================================================================================
static constexpr int kAeLutFloatsT = 512 * 4;          // = kAeLutFloats of nets_fused.hip (workspace layout)
static constexpr int kAeRows = 78;
static constexpr int kT1 = 38, kT1S = 40, kT1Plane = kT1 * kT1S;
static constexpr int kT2S = 20, kT2Plane = 20 * kT2S;  // A2 with a zero ring: local index + 1
static constexpr int kT3S = 36;

struct AeTileShared {
    float4 lut[512];
    float a1[4 * kT1Plane];            // A3 (34 x 36) overlays plane 0 once conv2 has read A1
    float a2[2 * kT2Plane];
    uint64_t rows[kAeRows + 2][2];
    uint64_t trows[64];                // TGT: the tile's own cells of the frame the reconstruction is compared with
    float red[4];
    int last;                          // PairFinish: this workgroup delivered the last tile of its grid
};

// 4 bits of a 78-bit row starting at local column c (0 <= c <= 74)
__device__ __forceinline__ uint32_t row_seg4(uint64_t lo, uint64_t hi, int c) {
    const uint64_t v = c < 64 ? ((lo >> c) | (c > 0 ? hi << (64 - c) : 0ull)) : (hi >> (c - 64));
    return (uint32_t)v & 0xFu;
}

__device__ __forceinline__ float tile_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }   // as nets_fused.hip

// TGT (PredictionBonus, mcl.py:581-605): the squared error of a tile is taken against the same cells of another bit-packed
// frame `tgt` (one aligned word per tile row) instead of the input
template <bool PACKED, bool TGT = false>
__device__ __forceinline__ void ae_tile_body(AeTileShared &s, const void *__restrict__ obs_v, const float *__restrict__ ws,
                                             float *__restrict__ partial, float *__restrict__ recon,
                                             int64_t n, int size, unsigned bid, unsigned nblk,
                                             const PairFinish pf = PairFinish(), const uint64_t *__restrict__ tgt = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double2 mom_acc = make_double2(0.0, 0.0);           // PairFinish: rewards this workgroup formed (thread 0)
    {
        const float4 *src = reinterpret_cast<const float4 *>(ws);
        lds_copy_f4<512, 256>(s.lut, src, tid);
        for (int i = tid; i < 2 * kT2Plane; i += 256) s.a2[i] = 0.0f;       // the ring stays zero
    }
    const float *prm = ws + kAeLutFloatsT;             // conv2.w[72] b[2] | convT1.w[32] b[1] | convT2.w[16] b[1]
    // the 122 uniform weights are re-derived from an opaque pointer behind every barrier of the tile loop (AE_W_HERE):
    // their scalar loads stay inside the phase that uses them (common.hpp: opaque_uniform)
    kfloat *w2 = opaque_uniform(prm), *b2 = w2 + 72, *wt1 = w2 + 74, *wt2 = w2 + 107;
    const float bt1 = prm[106], bt2 = prm[123];
#define AE_W_HERE() do { w2 = opaque_uniform(prm); b2 = w2 + 72; wt1 = w2 + 74; wt2 = w2 + 107; } while (0)
    const int tps = size >> 6, tiles = tps * tps;
    const int h2 = size >> 1, h4 = size >> 2;
    const int64_t total = n * tiles;
    const int py = tid >> 4, px = tid & 15;
    // loads of one tile into registers (fp32: the wave's rows, 2 values per lane and row; packed: thread r < kAeRows
    // assembles row r's two words): issued a whole iteration ahead
    constexpr int kPer = (kAeRows + 3) / 4;
    uint32_t va[kPer], vb[kPer];
    uint64_t plo = 0ull, phi = 0ull, tword = 0ull;
    auto issue = [&](int64_t t) {
        const int64_t g = t / tiles;
        const int tt = (int)(t % tiles), ty = tt / tps, tx = tt % tps;
        const int y0 = 64 * ty - 7, x0 = 64 * tx - 7;
        if (TGT) {
            if (tid < 64) tword = tgt[(g * size + 64 * ty + tid) * tps + tx];
        }
        if (PACKED) {
            plo = 0ull; phi = 0ull;
            const int y = y0 + tid;
            if (tid < kAeRows && y >= 0 && y < size) {
                const uint64_t *row = reinterpret_cast<const uint64_t *>(obs_v) + (g * size + y) * tps;
                const uint64_t wml = row[tx > 0 ? tx - 1 : tx], w0 = row[tx], wpl = row[tx + 1 < tps ? tx + 1 : tx];   // unconditional: no wait per load
                    const uint64_t wm = tx > 0 ? wml : 0ull, wp = tx + 1 < tps ? wpl : 0ull;
                plo = (wm >> 57) | (w0 << 7);
                phi = (w0 >> 57) | ((wp & 0x7Full) << 7);
            }
        } else {
            // unconditional loads of clamped addresses, validity applied in the ballot below (see rnd_tile_body)
            const float *gin = reinterpret_cast<const float *>(obs_v) + g * (int64_t)size * size;
            const int xac = min(max(x0 + lane, 0), size - 1), xbc = min(x0 + 64 + min(lane, 13), size - 1);
#pragma unroll
            for (int i = 0; i < kPer; ++i) {
                const int yc = min(max(y0 + wv + 4 * i, 0), size - 1);
                va[i] = __float_as_uint(gin[(int64_t)yc * size + xac]);
                vb[i] = __float_as_uint(gin[(int64_t)yc * size + xbc]);
            }
        }
    };
    if ((int64_t)bid < total) issue(bid);
    __syncthreads();
    for (int64_t t = bid; t < total; t += nblk) {
        const int64_t g = t / tiles;
        const int tt = (int)(t % tiles), ty = tt / tps, tx = tt % tps;
        // ---- the tile's kAeRows rows as bit words: from the registers the previous iteration (or the prologue) filled
        if (TGT) {
            if (tid < 64) s.trows[tid] = tword;
        }
        if (PACKED) {
            if (tid < kAeRows) { s.rows[tid][0] = plo; s.rows[tid][1] = phi; }
        } else {
            const int y0 = 64 * ty - 7, x0 = 64 * tx - 7;
            const int xa = x0 + lane, xb = x0 + 64 + lane;
#pragma unroll
            for (int i = 0; i < kPer; ++i) {
                const int r = wv + 4 * i, y = y0 + r;
                const bool yin = r < kAeRows && y >= 0 && y < size;
                const uint64_t lo = __ballot(yin && xa >= 0 && xa < size && (va[i] & 0x7FFFFFFFu) != 0u);
                const uint64_t hi = __ballot(yin && lane < 14 && xb < size && (vb[i] & 0x7FFFFFFFu) != 0u);
                if (lane == 0 && r < kAeRows) { s.rows[r][0] = lo; s.rows[r][1] = hi; }
            }
        }
        if (t + nblk < total) issue(t + nblk);          // the next tile travels during this one's layers
        __syncthreads();
        AE_W_HERE();

        // ---- conv1 (table) + relu + 2x2 pool -> A1, positions (32ty-3+pl, 32tx-3+ql), pl, ql in [0, 38)
        for (int i = tid; i < kT1 * kT1; i += 256) {
            const int pl = i / kT1, ql = i % kT1;
            const int P = 32 * ty - 3 + pl, Q = 32 * tx - 3 + ql;
            float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
            if (P >= 0 && P < h2 && Q >= 0 && Q < h2) {
                uint32_t seg[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) seg[r] = row_seg4(s.rows[2 * pl + r][0], s.rows[2 * pl + r][1], 2 * ql);
                float4 e[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int dy = d >> 1, dx = d & 1;
                    const uint32_t p = ((seg[dy] >> dx) & 7u) | (((seg[dy + 1] >> dx) & 7u) << 3) |
                                       (((seg[dy + 2] >> dx) & 7u) << 6);
                    e[d] = s.lut[p];
                }
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    m0 = fmaxf(m0, e[d].x); m1 = fmaxf(m1, e[d].y); m2 = fmaxf(m2, e[d].z); m3 = fmaxf(m3, e[d].w);
                }
            }
            const int off = pl * kT1S + ql;
            s.a1[0 * kT1Plane + off] = m0;
            s.a1[1 * kT1Plane + off] = m1;
            s.a1[2 * kT1Plane + off] = m2;
            s.a1[3 * kT1Plane + off] = m3;
        }
        __syncthreads();
        AE_W_HERE();

        // ---- conv2 (4->2) + relu + 2x2 pool -> A2, positions (16ty-1+il, 16tx-1+jl), il, jl in [0, 18); the
        //      output channel is wave-uniform (scalar weights): passes (oc 0, first 256), (oc 1, first 256), and
        //      the remaining 68 positions of both channels on waves 0-1 / 2-3
        auto conv2_item = [&](int pos, int oc) {
            const int il = pos / 18, jl = pos % 18;
            const int I = 16 * ty - 1 + il, J = 16 * tx - 1 + jl;
            float out = 0.0f;
            if (I >= 0 && I < h4 && J >= 0 && J < h4) {
                float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const float *pl = s.a1 + cc * kT1Plane + (2 * il) * kT1S + 2 * jl;
                    float patch[4][4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float2 lo = *reinterpret_cast<const float2 *>(pl + r * kT1S);
                        const float2 hi = *reinterpret_cast<const float2 *>(pl + r * kT1S + 2);
                        patch[r][0] = lo.x; patch[r][1] = lo.y; patch[r][2] = hi.x; patch[r][3] = hi.y;
                    }
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 2; ++dx)
#pragma unroll
                            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                                for (int kx = 0; kx < 3; ++kx)
                                    acc[dy][dx] = fmaf(w2[(oc * 4 + cc) * 9 + ky * 3 + kx], patch[dy + ky][dx + kx], acc[dy][dx]);
                }
                const float mx = fmaxf(fmaxf(acc[0][0], acc[0][1]), fmaxf(acc[1][0], acc[1][1])) + b2[oc];
                out = fmaxf(mx, 0.0f);
            }
            s.a2[oc * kT2Plane + (il + 1) * kT2S + jl + 1] = out;
        };
        conv2_item(tid, 0);
        conv2_item(tid, 1);
        {
            const int ocw = __builtin_amdgcn_readfirstlane(wv >> 1), rem = tid & 127;
            if (rem < 18 * 18 - 256) conv2_item(256 + rem, ocw);
        }
        __syncthreads();                               // A1 reads done: A3 may overlay plane 0
        AE_W_HERE();

        // ---- convT1 (2->1, k4 s2 p1) + relu -> A3 rows 32ty-1 .. 32ty+32: thread = the 2x2 outputs (2B, 2B+1) of
        //      A2 index B = 16ty-1+bl (as the 64x64 kernel: compile-time taps); outputs outside the range are dropped
        for (int i = tid; i < 18 * 18; i += 256) {
            const int bl = i / 18, cl = i % 18;
            float r2[2][3][3];                         // A2 rows B-1 .. B+1 (zero ring), cols C-1 .. C+1
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int x = 0; x < 3; ++x) r2[cc][r][x] = s.a2[cc * kT2Plane + (bl + r) * kT2S + cl + x];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    const int ry0 = 1, ky0 = j ? 2 : 1;
                    const int ry1 = j ? 2 : 0, ky1 = j ? 0 : 3;
                    const int cx0 = 1, kx0 = ii ? 2 : 1;
                    const int cx1 = ii ? 2 : 0, kx1 = ii ? 0 : 3;
                    float acc = 0.0f;
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc) {
                        acc = fmaf(r2[cc][ry0][cx0], wt1[cc * 16 + ky0 * 4 + kx0], acc);
                        acc = fmaf(r2[cc][ry0][cx1], wt1[cc * 16 + ky0 * 4 + kx1], acc);
                        acc = fmaf(r2[cc][ry1][cx0], wt1[cc * 16 + ky1 * 4 + kx0], acc);
                        acc = fmaf(r2[cc][ry1][cx1], wt1[cc * 16 + ky1 * 4 + kx1], acc);
                    }
                    const int lr = 2 * bl - 1 + j, lc = 2 * cl - 1 + ii;          // local A3 index (0 = global 32t - 1)
                    const int gr = 32 * ty - 1 + lr, gc = 32 * tx - 1 + lc;
                    if (lr >= 0 && lr < 34 && lc >= 0 && lc < 34)
                        s.a1[lr * kT3S + lc] = (gr >= 0 && gr < h2 && gc >= 0 && gc < h2) ? fmaxf(acc + bt1, 0.0f) : 0.0f;
                }
        }
        __syncthreads();
        AE_W_HERE();

        // ---- convT2 + sigmoid + squared error on the thread's own 4x4 cells (as the 64x64 kernel)
        float sq = 0.0f;
        {
            float t3[4][4];                            // A3 local rows 2py .. 2py+3, cols 2px .. 2px+3
            const float *pl = s.a1 + (2 * py) * kT3S + 2 * px;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float2 lo = *reinterpret_cast<const float2 *>(pl + r * kT3S);
                const float2 hi = *reinterpret_cast<const float2 *>(pl + r * kT3S + 2);
                t3[r][0] = lo.x; t3[r][1] = lo.y; t3[r][2] = hi.x; t3[r][3] = hi.y;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t bits = TGT ? (uint32_t)(s.trows[4 * py + r] >> (4 * px)) & 0xFu
                                          : row_seg4(s.rows[7 + 4 * py + r][0], s.rows[7 + 4 * py + r][1], 7 + 4 * px);
                float out4[4];
#pragma unroll
                for (int cx = 0; cx < 4; ++cx) {
                    const int my = r >> 1, mx = cx >> 1;
                    const int ry0 = my + 1, ky0 = (r & 1) ? 2 : 1;
                    const int ry1 = (r & 1) ? my + 2 : my, ky1 = (r & 1) ? 0 : 3;
                    const int cx0 = mx + 1, kx0 = (cx & 1) ? 2 : 1;
                    const int cx1 = (cx & 1) ? mx + 2 : mx, kx1 = (cx & 1) ? 0 : 3;
                    float acc = 0.0f;
                    acc = fmaf(t3[ry0][cx0], wt2[ky0 * 4 + kx0], acc);
                    acc = fmaf(t3[ry0][cx1], wt2[ky0 * 4 + kx1], acc);
                    acc = fmaf(t3[ry1][cx0], wt2[ky1 * 4 + kx0], acc);
                    acc = fmaf(t3[ry1][cx1], wt2[ky1 * 4 + kx1], acc);
                    const float pr = tile_sigmoid(acc + bt2);
                    out4[cx] = pr;
                    const float df = (float)((bits >> cx) & 1u) - pr;
                    sq = fmaf(df, df, sq);
                }
                if (recon)
                    *reinterpret_cast<float4 *>(recon + g * (int64_t)size * size + (int64_t)(64 * ty + 4 * py + r) * size +
                                                64 * tx + 4 * px) = make_float4(out4[0], out4[1], out4[2], out4[3]);
            }
        }
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) sq += __shfl_xor(sq, sft, 64);
        if (lane == 0) s.red[wv] = sq;
        __syncthreads();                               // also: rows / A3 reads done before the next tile
        AE_W_HERE();
        if (tid == 0) {
            const float sum = (s.red[0] + s.red[1]) + (s.red[2] + s.red[3]);
            if (pf.counters) __hip_atomic_store(partial + t, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else partial[t] = sum;
        }
        if (pf.counters) pair_ticket(pf, s.last, g, tiles, tid, mom_acc);     // kernel argument: workgroup-uniform
    }
    if (pf.counters) pair_leave_moments(pf, tid, mom_acc, n);
}

template <bool PACKED, bool TGT = false>
__global__ __launch_bounds__(256, 4) void ae_tile_kernel(const void *__restrict__ obs_v, const float *__restrict__ ws,
                                                         float *__restrict__ partial, float *__restrict__ recon,
                                                         int64_t n, int size, const uint64_t *__restrict__ tgt) {
    __shared__ AeTileShared s;
    ae_tile_body<PACKED, TGT>(s, obs_v, ws, partial, recon, n, size, blockIdx.x, gridDim.x, PairFinish(), tgt);
}

// ---------------------------------------------------------------------------------------------------------------
// RND2D + AE2D forward of a wrapper stack (AE2D(RND2D(env)), carle/train_mcl.py:34-38, evaluation/eval.py:83-85) in
// ONE launch instead of four: both nets read the same observation, so the tile workgroups of both run side by side
// in one launch -- workgroups [0, rnd_blocks) walk the RND2D tiles, the others the AE2D tiles, each with its own
// barriers and LDS image -- and one finish kernel forms both bonuses and adds them to the reward in the wrappers'
// order (reward += s_rnd * b_rnd, then += s_ae * b_ae: mcl.py:403-405 twice).  At the reference's default batch
// (16 x 256x256: 256 tiles per net, a tile per CU) a step is bound by its dependent dispatches, not by the kernels.
// ---------------------------------------------------------------------------------------------------------------
union RndAeTileShared {
    RndTileShared r;
    AeTileShared a;
};

template <bool PACKED>
__global__ __launch_bounds__(256, 4) void rnd_ae_tile_kernel(const void *__restrict__ obs_v, const float *__restrict__ rws,
                                                             const float *__restrict__ pp, const float *__restrict__ rp,
                                                             float *__restrict__ rpartial, const float *__restrict__ aws,
                                                             float *__restrict__ apartial, int64_t n, int size,
                                                             unsigned rnd_blocks, const PairFinish pf = PairFinish()) {
    __shared__ RndAeTileShared s;
    if (blockIdx.x < rnd_blocks)
        rnd_tile_body<PACKED>(s.r, obs_v, rws, pp, rp, rpartial, nullptr, n, size, blockIdx.x, rnd_blocks, RndTileFinish(), pf);
    else
        ae_tile_body<PACKED>(s.a, obs_v, aws, apartial, nullptr, n, size, blockIdx.x - rnd_blocks, gridDim.x - rnd_blocks, pf);
}

void ae2d_prepare_tables(const float *ap, float *ws, hipStream_t s);

// both wrappers' bonuses of one observation batch; workspaces as rnd2d_tiled / ae2d_tiled use them
// workgroups of the forward pair launch (= reward-moment pairs it leaves)
int64_t rnd_ae_tiled_blocks(int64_t n, int size) {
    const int64_t total = n * (int64_t)(size >> 6) * (size >> 6);
    return (total < 1024 ? total : 1024) + (total < 16384 ? total : 16384);
}

int rnd_ae_tiled(const void *obs, bool packed, int64_t n, int size, const float *pp, const float *rp, float *rws, float *rbonus,
                 float rscale, bool rnd_tables_valid, const float *ap, float *aws, float *abonus, float ascale,
                 bool ae_tables_valid, float *reward, double *mom_partials, int32_t *status, hipStream_t s) {
    if (!rnd_tables_valid) rnd2d_prepare_tables(pp, rp, rws, s);
    if (!ae_tables_valid) ae2d_prepare_tables(ap, aws, s);
    const int tiles = (size >> 6) * (size >> 6);
    const int64_t total = n * tiles;
    if ((n + 15) / 16 > 0x7FFFFFFFll) return fail_arg(CARLE_E_DIMS, "carle_rnd_ae_bonus: batch too large");
    const unsigned rb = (unsigned)(total < 1024 ? total : 1024), ab = (unsigned)(total < 16384 ? total : 16384);
    float *rpartial = rws + kFusedWsTotal, *apartial = aws + kAeLutFloatsT + 128;
    const int d8 = size >> 3;
    // ONE launch: the workgroup that delivers the last of a grid's 2 x tiles tiles finishes the grid (PairFinish).  The
    // arrival counters live behind the RND2D partial sums: zero whenever the RND2D tables are (re)built -- TABLES_VALID
    // promises the workspace of a finished call with the same parameters AND the same (n, size) -- and zero again
    // behind every launch.
    PairFinish pf;
    pf.counters = reinterpret_cast<int32_t *>(rpartial + total * 32);
    pf.rpartial = rpartial; pf.apartial = apartial; pf.pp = pp; pf.rp = rp;
    pf.rbonus = rbonus; pf.abonus = abonus; pf.reward = reward;
    pf.rscale = rscale; pf.ascale = ascale; pf.inv_pixels = 1.0f / ((float)size * (float)size);
    pf.dfeat = d8 * d8;
    pf.mom_partials = mom_partials; pf.status = status;
    if (!rnd_tables_valid) {
        if (hipError_t e = hipMemsetAsync(pf.counters, 0, (size_t)n * sizeof(int32_t), s); e != hipSuccess)
            return fail_hip(e, "carle_rnd_ae_bonus: hipMemsetAsync");
    }
    if (packed)
        hipLaunchKernelGGL(rnd_ae_tile_kernel<true>, dim3(rb + ab), dim3(256), 0, s, obs, rws, pp, rp, rpartial, aws, apartial, n, size, rb, pf);
    else
        hipLaunchKernelGGL(rnd_ae_tile_kernel<false>, dim3(rb + ab), dim3(256), 0, s, obs, rws, pp, rp, rpartial, aws, apartial, n, size, rb, pf);
    return check_launch("carle_rnd_ae_bonus(tiled)");
}

// ---------------------------------------------------------------------------------------------
// AE2D forward + BACKWARD, tiled (carle/mcl.py:504-547; eval mode): the bonus partial AND the gradient of
//   gs/2 * sum (obs - decoder(obs))^2  with respect to the 164 predictor parameters, per 64x64 tile in ONE kernel.
// A tile owns the gradient paths that start at its own 64x64 output cells; all activations on those paths lie
// inside the halos the forward needs (A3 +1, A2 +1, A1 +3, cells +7), so nothing crosses a tile border.  The
// layer-by-layer decomposition is the staged backward's (nets_staged_bwd.hpp: gathers, no atomics, first maximum
// in scan order) on LDS planes; transposed-convolution taps are compile-time constants because a thread handles
// 2x2 blocks aligned to even coordinates.  164 accumulators per thread over all tiles of the workgroup, one
// partial row per workgroup, staged_grad_reduce_kernel adds the rows in a fixed order.  77 KB of LDS: 2 per CU.
// ---------------------------------------------------------------------------------------------
struct AeTileTrainShared {
    float4 lut[512];
    float a1[4 * kT1Plane];
    float a2[2 * kT2Plane];            // zero ring (local index + 1)
    float a3[34 * kT3S];               // A3, later its gated gradient in place
    float g2[2 * 18 * 18];             // gated d loss / d A2
    float dpre[64][64];                // d loss / d pre-sigmoid of the tile's own cells
    float dmap[2][40][40];             // d loss / d conv2 output (local position + 2), zero ring
    uint64_t rows[kAeRows + 2][2];
    uint64_t trows[64];                // TGT: the tile's own cells of the target frame (PredictionBonus)
    uint16_t nb[kT1 * kT1];            // 4x4 cell neighbourhood of every A1 position
    uint8_t am[kT1 * kT1];             // 2-bit arg-max index of the four conv1 channels
    uint8_t sel2[2][18 * 18];          // arg-max of the 2x2 conv2 outputs under every A2 position
    float red_sq[8];                   // per-tile squared-error sums of the waves (the 164-wide rows at the end overlay dpre)
};

// NT threads per tile: 512 (one workgroup per CU) for small batches -- a tile is a chain of dependent LDS round trips,
// twice the waves halve it (N=16: 67 -> 41 us per call) -- 256 (two workgroups per CU) for large ones (N=256: 374
// against 399 us)
template <bool PACKED, int NT, bool TGT = false>
__global__ __launch_bounds__(NT, NT == 512 ? 1 : 2) void ae_tile_train_kernel(const void *__restrict__ obs_v, const float *__restrict__ ws,
                                                               float *__restrict__ partial, float *__restrict__ rows_out,
                                                               float gs, int64_t n, int size,
                                                               const uint64_t *__restrict__ tgt) {
    __shared__ AeTileTrainShared s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    {
        const float4 *src = reinterpret_cast<const float4 *>(ws);
        lds_copy_f4<512, NT>(s.lut, src, tid);
        for (int i = tid; i < 2 * kT2Plane; i += NT) s.a2[i] = 0.0f;
        for (int i = tid; i < 2 * 40 * 40; i += NT) (&s.dmap[0][0][0])[i] = 0.0f;
    }
    const float *prm = ws + kAeLutFloatsT;
    // the 122 uniform weights are re-derived from an opaque pointer behind every barrier of the tile loop (AE_W_HERE):
    // their scalar loads stay inside the phase that uses them (common.hpp: opaque_uniform)
    kfloat *w2 = opaque_uniform(prm), *b2 = w2 + 72, *wt1 = w2 + 74, *wt2 = w2 + 107;
    const float bt1 = prm[106], bt2 = prm[123];
#define AE_W_HERE() do { w2 = opaque_uniform(prm); b2 = w2 + 72; wt1 = w2 + 74; wt2 = w2 + 107; } while (0)
    const int tps = size >> 6, tiles = tps * tps;
    const int h2 = size >> 1, h4 = size >> 2;
    const int64_t total = n * tiles;
    const int py = (tid & 255) >> 4, px = tid & 15, hf = tid >> 8;      // P4: rows 2 hf, 2 hf + 1 of the 4x4 cell block (py, px)
    // the conv2 gradients are CHANNEL-SPLIT over the threads: thread tid accumulates the (output, input) channel pair
    // tid & 7 -- 9 + 1 accumulators instead of 72 + 2 (all 164 per thread: 256 VGPRs + 132 spilled, 134 us per tile);
    // conv1 stays per position (the gather of the map is shared by the four channels: 1 400 instead of 4 000
    // instructions per tile)
    float gw1[36], gb1[4] = {0.f, 0.f, 0.f, 0.f}, gw2[9], gb2 = 0.0f, gwt1[32], gbt1 = 0.0f, gwt2[16], gbt2 = 0.0f;
#pragma unroll
    for (int i = 0; i < 36; ++i) gw1[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 9; ++i) gw2[i] = 0.0f;
    const int oc7 = (tid >> 2) & 1, cc7 = tid & 3;     // conv2 (oc, cc) pair of this thread (P7)
#pragma unroll
    for (int i = 0; i < 32; ++i) gwt1[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) gwt2[i] = 0.0f;
    __syncthreads();
    for (int64_t t = blockIdx.x; t < total; t += gridDim.x) {
        const int64_t g = t / tiles;
        const int tt = (int)(t % tiles), ty = tt / tps, tx = tt % tps;
        const int y0 = 64 * ty - 7, x0 = 64 * tx - 7;
        // ---- P0: rows
        if (TGT) {                                     // the tile's own cells of the target frame: one aligned word per row
            if (tid >= NT - 64) s.trows[tid - (NT - 64)] = tgt[(g * size + 64 * ty + (tid - (NT - 64))) * tps + tx];
        }
        if (PACKED) {
            if (tid < kAeRows) {
                const int y = y0 + tid;
                uint64_t lo = 0ull, hi = 0ull;
                if (y >= 0 && y < size) {
                    const uint64_t *row = reinterpret_cast<const uint64_t *>(obs_v) + (g * size + y) * tps;
                    const uint64_t wml = row[tx > 0 ? tx - 1 : tx], w0 = row[tx], wpl = row[tx + 1 < tps ? tx + 1 : tx];   // unconditional: no wait per load
                    const uint64_t wm = tx > 0 ? wml : 0ull, wp = tx + 1 < tps ? wpl : 0ull;
                    lo = (wm >> 57) | (w0 << 7);
                    hi = (w0 >> 57) | ((wp & 0x7Full) << 7);
                }
                s.rows[tid][0] = lo;
                s.rows[tid][1] = hi;
            }
        } else {
            const float *gin = reinterpret_cast<const float *>(obs_v) + g * (int64_t)size * size;
            constexpr int kWaves = NT / 64, kPer = (kAeRows + kWaves - 1) / kWaves;
            uint32_t va[kPer], vb[kPer];
            // unconditional loads of clamped addresses, validity applied in the ballot (see rnd_tile_bwd_kernel)
            const int xa = x0 + lane, xb = x0 + 64 + lane;
            const int xac = min(max(xa, 0), size - 1), xbc = min(x0 + 64 + min(lane, 13), size - 1);
#pragma unroll
            for (int i = 0; i < kPer; ++i) {
                const int yc = min(max(y0 + wv + kWaves * i, 0), size - 1);
                va[i] = __float_as_uint(gin[(int64_t)yc * size + xac]);
                vb[i] = __float_as_uint(gin[(int64_t)yc * size + xbc]);
            }
#pragma unroll
            for (int i = 0; i < kPer; ++i) {
                const int r = wv + kWaves * i, y = y0 + r;
                const bool yin = r < kAeRows && y >= 0 && y < size;
                const uint64_t lo = __ballot(yin && xa >= 0 && xa < size && (va[i] & 0x7FFFFFFFu) != 0u);
                const uint64_t hi = __ballot(yin && lane < 14 && xb < size && (vb[i] & 0x7FFFFFFFu) != 0u);
                if (lane == 0 && r < kAeRows) { s.rows[r][0] = lo; s.rows[r][1] = hi; }
            }
        }
        __syncthreads();
        AE_W_HERE();

        // ---- P1: conv1 (table) + relu + 2x2 pool -> A1 with arg-max index and 4x4 cell neighbourhood
        for (int i = tid; i < kT1 * kT1; i += NT) {
            const int pl = i / kT1, ql = i % kT1;
            const int P = 32 * ty - 3 + pl, Q = 32 * tx - 3 + ql;
            float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
            uint32_t i0 = 0, i1 = 0, i2 = 0, i3 = 0, nbv = 0;
            if (P >= 0 && P < h2 && Q >= 0 && Q < h2) {
                uint32_t seg[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) seg[r] = row_seg4(s.rows[2 * pl + r][0], s.rows[2 * pl + r][1], 2 * ql);
                nbv = seg[0] | (seg[1] << 4) | (seg[2] << 8) | (seg[3] << 12);
                float4 e[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int dy = d >> 1, dx = d & 1;
                    const uint32_t p = ((seg[dy] >> dx) & 7u) | (((seg[dy + 1] >> dx) & 7u) << 3) |
                                       (((seg[dy + 2] >> dx) & 7u) << 6);
                    e[d] = s.lut[p];
                }
#pragma unroll
                for (int d = 0; d < 4; ++d) {          // strict >: the first maximum in scan order
                    if (e[d].x > m0) { m0 = e[d].x; i0 = d; }
                    if (e[d].y > m1) { m1 = e[d].y; i1 = d; }
                    if (e[d].z > m2) { m2 = e[d].z; i2 = d; }
                    if (e[d].w > m3) { m3 = e[d].w; i3 = d; }
                }
            }
            const int off = pl * kT1S + ql;
            s.a1[0 * kT1Plane + off] = m0; s.a1[1 * kT1Plane + off] = m1;
            s.a1[2 * kT1Plane + off] = m2; s.a1[3 * kT1Plane + off] = m3;
            s.nb[i] = (uint16_t)nbv;
            s.am[i] = (uint8_t)(i0 | (i1 << 2) | (i2 << 4) | (i3 << 6));
        }
        __syncthreads();
        AE_W_HERE();

        // ---- P2: conv2 (4->2) + relu + 2x2 pool -> A2 (zero ring) with the arg-max of the window
        auto conv2_item = [&](int pos, int oc) {
            const int il = pos / 18, jl = pos % 18;
            const int I = 16 * ty - 1 + il, J = 16 * tx - 1 + jl;
            float out = 0.0f;
            int sel = 0;
            if (I >= 0 && I < h4 && J >= 0 && J < h4) {
                float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const float *pl = s.a1 + cc * kT1Plane + (2 * il) * kT1S + 2 * jl;
                    float patch[4][4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float2 lo = *reinterpret_cast<const float2 *>(pl + r * kT1S);
                        const float2 hi = *reinterpret_cast<const float2 *>(pl + r * kT1S + 2);
                        patch[r][0] = lo.x; patch[r][1] = lo.y; patch[r][2] = hi.x; patch[r][3] = hi.y;
                    }
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 2; ++dx)
#pragma unroll
                            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                                for (int kx = 0; kx < 3; ++kx)
                                    acc[dy][dx] = fmaf(w2[(oc * 4 + cc) * 9 + ky * 3 + kx], patch[dy + ky][dx + kx], acc[dy][dx]);
                }
                float mx = acc[0][0];
                if (acc[0][1] > mx) { mx = acc[0][1]; sel = 1; }
                if (acc[1][0] > mx) { mx = acc[1][0]; sel = 2; }
                if (acc[1][1] > mx) { mx = acc[1][1]; sel = 3; }
                out = fmaxf(mx + b2[oc], 0.0f);
            }
            s.a2[oc * kT2Plane + (il + 1) * kT2S + jl + 1] = out;
            s.sel2[oc][pos] = (uint8_t)sel;
        };
        if (NT == 512) {
            const int ocw = __builtin_amdgcn_readfirstlane(wv >> 2), rem = tid & 255;      // waves 0-3: oc 0, waves 4-7: oc 1
            conv2_item(rem, ocw);
            if (rem < 18 * 18 - 256) conv2_item(256 + rem, ocw);
        } else {
            conv2_item(tid, 0);
            conv2_item(tid, 1);
            const int ocw = __builtin_amdgcn_readfirstlane(wv >> 1), rem = tid & 127;
            if (rem < 18 * 18 - 256) conv2_item(256 + rem, ocw);
        }
        __syncthreads();
        AE_W_HERE();

        // ---- P3: convT1 + relu -> A3 (rows 32ty-1 .. 32ty+32), blocks of 2x2 outputs per A2 index
        for (int i = tid; i < 18 * 18; i += NT) {
            const int bl = i / 18, cl = i % 18;
            float r2[2][3][3];
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int x = 0; x < 3; ++x) r2[cc][r][x] = s.a2[cc * kT2Plane + (bl + r) * kT2S + cl + x];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    const int ry0 = 1, ky0 = j ? 2 : 1;
                    const int ry1 = j ? 2 : 0, ky1 = j ? 0 : 3;
                    const int cx0 = 1, kx0 = ii ? 2 : 1;
                    const int cx1 = ii ? 2 : 0, kx1 = ii ? 0 : 3;
                    float acc = 0.0f;
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc) {
                        acc = fmaf(r2[cc][ry0][cx0], wt1[cc * 16 + ky0 * 4 + kx0], acc);
                        acc = fmaf(r2[cc][ry0][cx1], wt1[cc * 16 + ky0 * 4 + kx1], acc);
                        acc = fmaf(r2[cc][ry1][cx0], wt1[cc * 16 + ky1 * 4 + kx0], acc);
                        acc = fmaf(r2[cc][ry1][cx1], wt1[cc * 16 + ky1 * 4 + kx1], acc);
                    }
                    const int lr = 2 * bl - 1 + j, lc = 2 * cl - 1 + ii;
                    const int gr = 32 * ty - 1 + lr, gc = 32 * tx - 1 + lc;
                    if (lr >= 0 && lr < 34 && lc >= 0 && lc < 34)
                        s.a3[lr * kT3S + lc] = (gr >= 0 && gr < h2 && gc >= 0 && gc < h2) ? fmaxf(acc + bt1, 0.0f) : 0.0f;
                }
        }
        __syncthreads();
        AE_W_HERE();

        // ---- P4: convT2 + sigmoid + loss on the thread's own 4x4 cells; d loss / d pre-sigmoid; convT2 gradients
        float sq = 0.0f;
        {
            float t3[4][4];
            const float *pl = s.a3 + (2 * py) * kT3S + 2 * px;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float2 lo = *reinterpret_cast<const float2 *>(pl + r * kT3S);
                const float2 hi = *reinterpret_cast<const float2 *>(pl + r * kT3S + 2);
                t3[r][0] = lo.x; t3[r][1] = lo.y; t3[r][2] = hi.x; t3[r][3] = hi.y;
            }
            auto cell_row = [&](auto rc) {               // rc: compile-time row of the 4x4 block
                constexpr int r = decltype(rc)::value;
                const uint32_t bits = TGT ? (uint32_t)(s.trows[4 * py + r] >> (4 * px)) & 0xFu
                                          : row_seg4(s.rows[7 + 4 * py + r][0], s.rows[7 + 4 * py + r][1], 7 + 4 * px);
                float d4[4];
#pragma unroll
                for (int cx = 0; cx < 4; ++cx) {
                    const int my = r >> 1, mx = cx >> 1;
                    const int ry0 = my + 1, ky0 = (r & 1) ? 2 : 1;
                    const int ry1 = (r & 1) ? my + 2 : my, ky1 = (r & 1) ? 0 : 3;
                    const int cx0 = mx + 1, kx0 = (cx & 1) ? 2 : 1;
                    const int cx1 = (cx & 1) ? mx + 2 : mx, kx1 = (cx & 1) ? 0 : 3;
                    float acc = 0.0f;
                    acc = fmaf(t3[ry0][cx0], wt2[ky0 * 4 + kx0], acc);
                    acc = fmaf(t3[ry0][cx1], wt2[ky0 * 4 + kx1], acc);
                    acc = fmaf(t3[ry1][cx0], wt2[ky1 * 4 + kx0], acc);
                    acc = fmaf(t3[ry1][cx1], wt2[ky1 * 4 + kx1], acc);
                    const float pr = tile_sigmoid(acc + bt2);
                    const float ob = (float)((bits >> cx) & 1u);
                    const float df = ob - pr;
                    sq = fmaf(df, df, sq);
                    const float dz = gs * (pr - ob) * pr * (1.0f - pr);
                    d4[cx] = dz;
                    gbt2 += dz;
                    gwt2[ky0 * 4 + kx0] = fmaf(dz, t3[ry0][cx0], gwt2[ky0 * 4 + kx0]);
                    gwt2[ky0 * 4 + kx1] = fmaf(dz, t3[ry0][cx1], gwt2[ky0 * 4 + kx1]);
                    gwt2[ky1 * 4 + kx0] = fmaf(dz, t3[ry1][cx0], gwt2[ky1 * 4 + kx0]);
                    gwt2[ky1 * 4 + kx1] = fmaf(dz, t3[ry1][cx1], gwt2[ky1 * 4 + kx1]);
                }
                *reinterpret_cast<float4 *>(&s.dpre[4 * py + r][4 * px]) = make_float4(d4[0], d4[1], d4[2], d4[3]);
            };
            if (NT == 256 || hf == 0) {                  // wave-uniform (hf = wave / 4)
                cell_row(std::integral_constant<int, 0>());
                cell_row(std::integral_constant<int, 1>());
            }
            if (NT == 256 || hf == 1) {
                cell_row(std::integral_constant<int, 2>());
                cell_row(std::integral_constant<int, 3>());
            }
        }
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) sq += __shfl_xor(sq, sft, 64);
        if (lane == 0) s.red_sq[wv] = sq;
        __syncthreads();
        AE_W_HERE();
        if (tid == 0)
            partial[t] = ((s.red_sq[0] + s.red_sq[1]) + (s.red_sq[2] + s.red_sq[3])) +
                         (NT == 512 ? ((s.red_sq[4] + s.red_sq[5]) + (s.red_sq[6] + s.red_sq[7])) : 0.0f);

        // ---- P5: d loss / d A3 gathered from the own cells each A3 position feeds, relu gate, in place
        for (int i = tid; i < 34 * 34; i += NT) {
            const int jl = i / 34, kl = i % 34;
            float da = 0.0f;
#pragma unroll
            for (int ky = 0; ky < 4; ++ky)
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) {
                    const int oy = 2 * jl - 3 + ky, ox = 2 * kl - 3 + kx;      // local output cell fed through tap (ky, kx)
                    if (oy >= 0 && oy < 64 && ox >= 0 && ox < 64) da = fmaf(s.dpre[oy][ox], wt2[ky * 4 + kx], da);
                }
            float *a = &s.a3[jl * kT3S + kl];
            *a = *a > 0.0f ? da : 0.0f;
        }
        __syncthreads();
        AE_W_HERE();

        // ---- P6: convT1 weight gradients (2x2 blocks, compile-time taps) and d loss / d A2 (gather), relu gate
        for (int i = tid; i < 18 * 18; i += NT) {
            const int bl = i / 18, cl = i % 18;
            float r2[2][3][3];
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int x = 0; x < 3; ++x) r2[cc][r][x] = s.a2[cc * kT2Plane + (bl + r) * kT2S + cl + x];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    const int lr = 2 * bl - 1 + j, lc = 2 * cl - 1 + ii;
                    const float gz = (lr >= 0 && lr < 34 && lc >= 0 && lc < 34) ? s.a3[lr * kT3S + lc] : 0.0f;
                    gbt1 += gz;
                    const int ry0 = 1, ky0 = j ? 2 : 1;
                    const int ry1 = j ? 2 : 0, ky1 = j ? 0 : 3;
                    const int cx0 = 1, kx0 = ii ? 2 : 1;
                    const int cx1 = ii ? 2 : 0, kx1 = ii ? 0 : 3;
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc) {
                        gwt1[cc * 16 + ky0 * 4 + kx0] = fmaf(gz, r2[cc][ry0][cx0], gwt1[cc * 16 + ky0 * 4 + kx0]);
                        gwt1[cc * 16 + ky0 * 4 + kx1] = fmaf(gz, r2[cc][ry0][cx1], gwt1[cc * 16 + ky0 * 4 + kx1]);
                        gwt1[cc * 16 + ky1 * 4 + kx0] = fmaf(gz, r2[cc][ry1][cx0], gwt1[cc * 16 + ky1 * 4 + kx0]);
                        gwt1[cc * 16 + ky1 * 4 + kx1] = fmaf(gz, r2[cc][ry1][cx1], gwt1[cc * 16 + ky1 * 4 + kx1]);
                    }
                }
            // A2 (il, jl2) = this block's index: it feeds A3 local rows 2 il - 2 + ky, columns 2 jl2 - 2 + kx
            float d2[2] = {0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 4; ++ky)
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) {
                    const int lr = 2 * bl - 2 + ky, lc = 2 * cl - 2 + kx;
                    if (lr >= 0 && lr < 34 && lc >= 0 && lc < 34) {
                        const float gz = s.a3[lr * kT3S + lc];
                        d2[0] = fmaf(gz, wt1[ky * 4 + kx], d2[0]);
                        d2[1] = fmaf(gz, wt1[16 + ky * 4 + kx], d2[1]);
                    }
                }
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
                s.g2[cc * 324 + i] = s.a2[cc * kT2Plane + (bl + 1) * kT2S + cl + 1] > 0.0f ? d2[cc] : 0.0f;
        }
        __syncthreads();
        AE_W_HERE();

        // ---- P7: conv2 weight gradients and the sparse conv2-output gradient map; thread = ((oc, cc) pair, 1 of 32 slots)
        for (int pos = tid >> 3; pos < 18 * 18; pos += NT / 8) {
            const int il = pos / 18, jl = pos % 18;
            const float dd = s.g2[oc7 * 324 + pos];
            const int sel = s.sel2[oc7][pos], ay = sel >> 1, ax = sel & 1;
            if (cc7 == 0) {
                gb2 += dd;
#pragma unroll
                for (int e = 0; e < 4; ++e) s.dmap[oc7][2 * il + (e >> 1) + 2][2 * jl + (e & 1) + 2] = e == sel ? dd : 0.0f;
            }
            const float *pl = s.a1 + cc7 * kT1Plane + (2 * il + ay) * kT1S + 2 * jl + ax;     // the 3x3 window under the arg-max
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) gw2[ky * 3 + kx] = fmaf(dd, pl[ky * kT1S + kx], gw2[ky * 3 + kx]);
        }
        __syncthreads();
        AE_W_HERE();

        // ---- P8: conv1 weight / bias gradients; thread = A1 position, gradient gathered from the map
        for (int i = tid; i < kT1 * kT1; i += NT) {
            const int pl = i / kT1, ql = i % kT1;
            float dm[2][3][3];
#pragma unroll
            for (int oc = 0; oc < 2; ++oc)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) dm[oc][ky][kx] = s.dmap[oc][pl - ky + 2][ql - kx + 2];
            const uint32_t nbv = s.nb[i], amv = s.am[i];
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                float da = 0.0f;
#pragma unroll
                for (int oc = 0; oc < 2; ++oc)
#pragma unroll
                    for (int tq = 0; tq < 9; ++tq) da = fmaf(w2[(oc * 4 + ch) * 9 + tq], dm[oc][tq / 3][tq % 3], da);
                const float g1 = s.a1[ch * kT1Plane + pl * kT1S + ql] > 0.0f ? da : 0.0f;
                gb1[ch] += g1;
                const uint32_t idx = (amv >> (2 * ch)) & 3u;
                const uint32_t tb = nbv >> (((idx & 2u) << 1) | (idx & 1u));
#pragma unroll
                for (int tq = 0; tq < 9; ++tq)
                    gw1[ch * 9 + tq] = fmaf(g1, (float)((tb >> (4 * (tq / 3) + tq % 3)) & 1u), gw1[ch * 9 + tq]);
            }
        }
        __syncthreads();
        AE_W_HERE();
    }
    // ---- one partial row per workgroup, blob order: conv1.w[36] b[4] | conv2.w[72] b[2] | convT1.w[32] b[1] | convT2.w[16] b[1]
    // (one loop over all 164 entries is not unrolled by the compiler: the accumulator arrays would be indexed
    // dynamically and move to LDS)
    float (*red)[164] = reinterpret_cast<float (*)[164]>(&s.dpre[0][0]);     // dpre is dead: every wave passed the last barrier
    {
        const int mine = oc7 * 4 + cc7;                  // the conv2 (output, input) channel pair this thread accumulated
        auto get = [&](auto ic) -> float {
            constexpr int i = decltype(ic)::value;
            if constexpr (i < 36) return gw1[i];
            else if constexpr (i < 40) return gb1[i - 36];
            else if constexpr (i < 112) return (i - 40) / 9 == mine ? gw2[(i - 40) % 9] : 0.0f;
            else if constexpr (i < 114) return mine == 4 * (i - 112) ? gb2 : 0.0f;
            else if constexpr (i < 146) return gwt1[i - 114];
            else if constexpr (i == 146) return gbt1;
            else if constexpr (i < 163) return gwt2[i - 147];
            else if constexpr (i == 163) return gbt2;
            else return 0.0f;
        };
        float sums[3];
        const int lo = wave_reduce_scatter<192>(get, lane, sums);
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (lo + j < 164) red[wv][lo + j] = sums[j];
    }
    __syncthreads();
    AE_W_HERE();
    if (tid < 164)
        rows_out[(int64_t)blockIdx.x * 164 + tid] = ((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid])) +
                                                    (NT == 512 ? ((red[4][tid] + red[5][tid]) + (red[6][tid] + red[7][tid])) : 0.0f);
}

int ae2d_tiled_train_rows(int64_t n, int size) {
    const int64_t total = n * (int64_t)(size >> 6) * (size >> 6);
    return (int)(total < 512 ? total : 512);
}

void ae2d_prepare_tables(const float *ap, float *ws, hipStream_t s);

// tables in ws[0 .. 2176), one squared-error partial per tile behind them; gradient rows -> rows
int ae2d_tiled_train(const void *obs, bool packed, int64_t n, int size, const float *ap, float *ws, float *rows,
                     float grad_scale, bool tables_valid, hipStream_t s, const uint64_t *tgt = nullptr) {
    if (!tables_valid) ae2d_prepare_tables(ap, ws, s);
    const unsigned blocks = (unsigned)ae2d_tiled_train_rows(n, size);
    float *partial = ws + kAeLutFloatsT + 128;
    const int64_t total = n * (int64_t)(size >> 6) * (size >> 6);
    const uint64_t *no_tgt = nullptr;
    if (total <= 512) {                                // at most two tiles per CU: 512 threads per tile
        if (tgt)                                       // (bit-packed frames only)
            hipLaunchKernelGGL((ae_tile_train_kernel<true, 512, true>), dim3(blocks), dim3(512), 0, s, obs, ws, partial, rows, grad_scale, n, size, tgt);
        else if (packed)
            hipLaunchKernelGGL((ae_tile_train_kernel<true, 512, false>), dim3(blocks), dim3(512), 0, s, obs, ws, partial, rows, grad_scale, n, size, no_tgt);
        else
            hipLaunchKernelGGL((ae_tile_train_kernel<false, 512, false>), dim3(blocks), dim3(512), 0, s, obs, ws, partial, rows, grad_scale, n, size, no_tgt);
    } else {
        if (tgt)
            hipLaunchKernelGGL((ae_tile_train_kernel<true, 256, true>), dim3(blocks), dim3(256), 0, s, obs, ws, partial, rows, grad_scale, n, size, tgt);
        else if (packed)
            hipLaunchKernelGGL((ae_tile_train_kernel<true, 256, false>), dim3(blocks), dim3(256), 0, s, obs, ws, partial, rows, grad_scale, n, size, no_tgt);
        else
            hipLaunchKernelGGL((ae_tile_train_kernel<false, 256, false>), dim3(blocks), dim3(256), 0, s, obs, ws, partial, rows, grad_scale, n, size, no_tgt);
    }
    return check_launch("carle_ae2d_train(tiled)");
}

// ---------------------------------------------------------------------------------------------
// AE2D(RND2D(env)) in training mode at tiled sizes -- the stack of carle/train_mcl.py:34-38 at the reference's default
// geometry: both wrappers' bonuses, reward updates and predictor gradients of one step in FOUR launches instead of nine
// (a step there is a chain of dependent dispatches of 4-7 us each, whatever their size):
//   rnd_tile_kernel (features kept; the last tile of a grid finishes it) -> rnd_bwd_pair_kernel (tile gradients | linear-layer gradient)
//   -> ae_tile_train_kernel -> rnd_ae_train_tail_kernel (row reduction RND2D | row reduction AE2D | AE2D bonus + reward).
// Same device bodies, same order of every sum as the two wrappers stepping one after the other: bit-identical.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void rnd_ae_train_tail_kernel(const float *__restrict__ rrows, int rrow_count,
                                                                float *__restrict__ rgrad, const float *__restrict__ arows,
                                                                int arow_count, float *__restrict__ agrad,
                                                                const float *__restrict__ apartial, float *__restrict__ abonus,
                                                                float *__restrict__ reward, float ascale, int64_t n, int tiles,
                                                                float inv_pixels, double *__restrict__ mom_partials,
                                                                int32_t *__restrict__ status) {
    __shared__ float part[32][33];
    __shared__ double2 mpart[16];
    const int b = (int)blockIdx.x;
    if (b < 3) staged_grad_reduce_body(part, rrows, rrow_count, rgrad, 77, b);                 // workgroup-uniform branches
    else if (b < 9) staged_grad_reduce_body(part, arows, arow_count, agrad, 164, b - 3);
    else {
        const int64_t g = (int64_t)(b - 9) * 1024 + threadIdx.x;
        const float r = ae_mse_finish_body(apartial, abonus, reward, ascale, n, tiles, inv_pixels, g);
        if (mom_partials && status && reward)
            reward_moment_partials(mpart, g < n, r, mom_partials, status, b - 9, (int)gridDim.x - 9, n);
    }
}

// rscratch: pred (n x 16) | targ (n x 16) | features (n x 2 x D) | rows, as carle_rnd2d_train_scratch_floats lays it out;
// arows: the AE2D gradient rows (carle_ae2d_train_scratch_floats)
int rnd_ae_tiled_train(const void *obs, bool packed, int64_t n, int size, const float *pp, const float *rp, float *rws,
                       float *rscratch, float *rgrad, float *rbonus, float rscale, float rgrad_scale, bool rvalid,
                       const float *ap, float *aws, float *arows, float *agrad, float *abonus, float ascale, float agrad_scale,
                       bool avalid, float *reward, double *mom_partials, int32_t *status, hipStream_t s) {
    if (!rvalid) rnd2d_prepare_tables(pp, rp, rws, s);
    const int tiles = (size >> 6) * (size >> 6);
    const int64_t total = n * tiles;
    if ((n + 15) / 16 > 0x7FFFFFFFll || (n + 1023) / 1024 > 0x7FFFFF00ll)
        return fail_arg(CARLE_E_DIMS, "carle_rnd_ae_train: batch too large");
    const int d8 = size >> 3, d = d8 * d8;
    float *pred = rscratch, *targ = pred + n * kRndK, *featbuf = targ + n * kRndK, *rrows = featbuf + n * 2 * (int64_t)d;
    const unsigned blocks = (unsigned)(total < 1024 ? total : 1024);
    float *rpartial = rws + kFusedWsTotal;
    // the workgroup that delivers a grid's last tile finishes the grid (bonus, outputs, reward): no finish launch.  The
    // arrival counters live behind the partial sums; zero whenever the tables are (re)built -- TABLES_VALID promises the
    // workspace of a finished call with the same parameters AND the same (n, size) -- and zero again behind every launch.
    RndTileFinish fin;
    fin.counters = reinterpret_cast<int32_t *>(rpartial + total * 32);
    fin.bonus = rbonus; fin.pred = pred; fin.targ = targ; fin.reward = reward; fin.reward_scale = rscale;
    // fp32 observations: the forward tiles leave a bit-packed copy at the end of the RND2D scratch
    // (carle_rnd2d_train_scratch_floats reserves it); the gradient tiles and the AE2D tiles read that
    const void *obs2 = obs;
    bool packed2 = packed;
    if (!packed) {
        fin.bits_out = reinterpret_cast<uint64_t *>(rscratch + rnd2d_tiled_bits_offset_floats(n, size));
        obs2 = fin.bits_out;
        packed2 = true;
    }
    if (!rvalid) {
        if (hipError_t e = hipMemsetAsync(fin.counters, 0, (size_t)n * sizeof(int32_t), s); e != hipSuccess)
            return fail_hip(e, "carle_rnd_ae_train: hipMemsetAsync");
    }
    if (packed)
        hipLaunchKernelGGL(rnd_tile_kernel<true>, dim3(blocks), dim3(256), 0, s, obs, rws, pp, rp, rpartial, featbuf, n, size, fin);
    else
        hipLaunchKernelGGL(rnd_tile_kernel<false>, dim3(blocks), dim3(256), 0, s, obs, rws, pp, rp, rpartial, featbuf, n, size, fin);
    const int brows = rnd2d_tiled_bwd_rows(n, size);
    const unsigned lin = (unsigned)(((int64_t)kRndK * d + kRndK + 255) / 256);
    if (packed2)
        hipLaunchKernelGGL(rnd_bwd_pair_kernel<true>, dim3((unsigned)brows + lin), dim3(256), 0, s, obs2, rws, pp, pred, targ,
                           rrows, featbuf, rgrad, rgrad_scale, n, size, brows);
    else
        hipLaunchKernelGGL(rnd_bwd_pair_kernel<false>, dim3((unsigned)brows + lin), dim3(256), 0, s, obs2, rws, pp, pred, targ,
                           rrows, featbuf, rgrad, rgrad_scale, n, size, brows);
    if (int e = ae2d_tiled_train(obs2, packed2, n, size, ap, aws, arows, agrad_scale, avalid, s)) return e;
    hipLaunchKernelGGL(rnd_ae_train_tail_kernel, dim3(9u + (unsigned)((n + 1023) / 1024)), dim3(1024), 0, s, rrows, brows, rgrad,
                       arows, ae2d_tiled_train_rows(n, size), agrad, aws + kAeLutFloatsT + 128, abonus, reward, ascale, n, tiles,
                       1.0f / ((float)size * (float)size), mom_partials, status);
    return check_launch("carle_rnd_ae_train(tiled)");
}

int64_t ae2d_tiled_workspace_floats(int64_t n, int size) {
    const int64_t tiles = (int64_t)(size >> 6) * (size >> 6);
    return kAeLutFloatsT + 128 + n * tiles;
}


// tables in ws[0 .. 2176), one partial per tile behind them; the caller's finish kernel sums them
int ae2d_tiled(const void *obs, bool packed, int64_t n, int size, const float *ap, float *ws, float *recon,
               bool tables_valid, hipStream_t s, const uint64_t *tgt) {
    if (!tables_valid) ae2d_prepare_tables(ap, ws, s);
    const int64_t total = n * (int64_t)(size >> 6) * (size >> 6);
    // up to 16384 workgroups (4 per CU resident, the others start out of phase as those retire): N=2048 x 256x256
    // 532 -> 475 us, N=256 75 -> 72 us against 1024 persistent ones; the RND2D tile kernel measured flat (stays at 1024)
    const unsigned blocks = (unsigned)(total < 16384 ? total : 16384);
    float *partial = ws + kAeLutFloatsT + 128;
    const uint64_t *no_tgt = nullptr;
    if (tgt)                                           // (bit-packed frames only)
        hipLaunchKernelGGL((ae_tile_kernel<true, true>), dim3(blocks), dim3(256), 0, s, obs, ws, partial, recon, n, size, tgt);
    else if (packed)
        hipLaunchKernelGGL((ae_tile_kernel<true, false>), dim3(blocks), dim3(256), 0, s, obs, ws, partial, recon, n, size, no_tgt);
    else
        hipLaunchKernelGGL((ae_tile_kernel<false, false>), dim3(blocks), dim3(256), 0, s, obs, ws, partial, recon, n, size, no_tgt);
    return check_launch("carle_ae2d_bonus(tiled)");
}


bool rnd2d_tiled_applies(int h, int w) { return h == w && h > 64 && (h & 63) == 0; }

// workspace: the fused tables (kFusedWsTotal floats) + 32 partial sums per tile
int64_t rnd2d_tiled_workspace_floats(int64_t n, int size) {
    const int64_t tiles = (int64_t)(size >> 6) * (size >> 6);
    return kFusedWsTotal + n * tiles * 32 + n;            // + one arrival counter per grid (RndTileFinish)
}

int rnd2d_tiled(const void *obs, bool packed, int64_t n, int size, const float *pp, const float *rp, float *ws,
                float *bonus, float *pred_out, float *target_out, float *reward, float reward_scale, bool tables_valid,
                hipStream_t s) {
    if (!tables_valid) rnd2d_prepare_tables(pp, rp, ws, s);
    const int tiles = (size >> 6) * (size >> 6);
    const int64_t total = n * tiles;
    if ((n + 15) / 16 > 0x7FFFFFFFll) return fail_arg(CARLE_E_DIMS, "carle_rnd2d_bonus: batch too large");
    const unsigned blocks = (unsigned)(total < 1024 ? total : 1024);      // persistent: 4 workgroups per CU
    float *partial = ws + kFusedWsTotal;
    // ONE launch: the workgroup that delivers a grid's last tile finishes the grid (RndTileFinish; counters behind the
    // partial sums, zero whenever the tables are rebuilt -- TABLES_VALID also promises the previous call's (n, size))
    RndTileFinish fin;
    fin.counters = reinterpret_cast<int32_t *>(partial + total * 32);
    fin.bonus = bonus; fin.pred = pred_out; fin.targ = target_out; fin.reward = reward; fin.reward_scale = reward_scale;
    if (!tables_valid) {
        if (hipError_t e = hipMemsetAsync(fin.counters, 0, (size_t)n * sizeof(int32_t), s); e != hipSuccess)
            return fail_hip(e, "carle_rnd2d_bonus: hipMemsetAsync");
    }
    if (packed)
        hipLaunchKernelGGL(rnd_tile_kernel<true>, dim3(blocks), dim3(256), 0, s, obs, ws, pp, rp, partial, static_cast<float *>(nullptr), n, size, fin);
    else
        hipLaunchKernelGGL(rnd_tile_kernel<false>, dim3(blocks), dim3(256), 0, s, obs, ws, pp, rp, partial, static_cast<float *>(nullptr), n, size, fin);
    return check_launch("carle_rnd2d_bonus(tiled)");
}

}  // namespace carle

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

namespace carle {

static constexpr int kTileRows = 74;                   // 64 + 2 * 5 cells
static constexpr int kTileWin = 18;                    // pooled windows per side: 16 + 2

struct RndTileShared {
    RndBlockShared net;                                // tables, planes, conv2 parameters (lin / rows unused)
    uint64_t rows[kTileRows + 2][2];                   // [r][0] = columns x0 .. x0+63, [r][1] = x0+64 .. x0+73
    float feat[2][64];
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

// PACKED: obs = bit-packed grids (H rows of W/64 words); else fp32 (N,1,H,W), any non-zero number = alive
template <bool PACKED>
__global__ __launch_bounds__(256, 4) void rnd_tile_kernel(const void *__restrict__ obs_v, const float *__restrict__ ws,
                                                          const float *__restrict__ pp, const float *__restrict__ rp,
                                                          float *__restrict__ partial, int64_t n, int size) {
    __shared__ RndTileShared s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    rnd_block_init(s.net, ws, tid);
    const int tps = size >> 6, tiles = tps * tps;      // tiles per side / per grid
    const int d8 = size >> 3, dfeat = d8 * d8;          // feature columns per row / per net
    const int64_t total = n * tiles;
    __syncthreads();
    for (int64_t t = blockIdx.x; t < total; t += gridDim.x) {
        const int64_t g = t / tiles;
        const int tt = (int)(t % tiles), ty = tt / tps, tx = tt % tps;
        const int y0 = 64 * ty - 5, x0 = 64 * tx - 5;
        // ---- the tile's 74 rows as 74-bit words
        if (PACKED) {
            if (tid < kTileRows) {
                const int y = y0 + tid;
                uint64_t lo = 0ull, hi = 0ull;
                if (y >= 0 && y < size) {
                    const uint64_t *row = reinterpret_cast<const uint64_t *>(obs_v) + (g * size + y) * tps;
                    const uint64_t wm = tx > 0 ? row[tx - 1] : 0ull, w0 = row[tx], wp = tx + 1 < tps ? row[tx + 1] : 0ull;
                    lo = (wm >> 59) | (w0 << 5);
                    hi = (w0 >> 59) | ((wp & 0x1Full) << 5);
                }
                s.rows[tid][0] = lo;
                s.rows[tid][1] = hi;
            }
        } else {
            const float *gin = reinterpret_cast<const float *>(obs_v) + g * (int64_t)size * size;
            for (int r = wv; r < kTileRows; r += 4) {   // wave = row: two coalesced loads, two ballots
                const int y = y0 + r;
                const bool yin = y >= 0 && y < size;
                const int xa = x0 + lane, xb = x0 + 64 + lane;
                uint32_t va = 0u, vb = 0u;
                if (yin && xa >= 0 && xa < size) va = __float_as_uint(gin[(int64_t)y * size + xa]);
                if (yin && lane < 10 && xb < size) vb = __float_as_uint(gin[(int64_t)y * size + xb]);
                const uint64_t lo = __ballot((va & 0x7FFFFFFFu) != 0u);
                const uint64_t hi = __ballot((vb & 0x7FFFFFFFu) != 0u);
                if (lane == 0) { s.rows[r][0] = lo; s.rows[r][1] = hi; }
            }
        }
        __syncthreads();
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
        // ---- the tile's share of the linear layers: thread = (net, k, feature row fy of the tile)
        {
            const int fy = tid & 7, k = (tid >> 3) & 15, net = tid >> 7;
            const float *wrow = (net == 0 ? pp + 77 : rp + 39) + (int64_t)k * dfeat + (int64_t)(8 * ty + fy) * d8 + 8 * tx;
            const F4T a = *reinterpret_cast<const F4T *>(wrow), b = *reinterpret_cast<const F4T *>(wrow + 4);
            const float *f = &s.feat[net][8 * fy];
            float acc = a.x * f[0];
            acc = fmaf(a.y, f[1], acc); acc = fmaf(a.z, f[2], acc); acc = fmaf(a.w, f[3], acc);
            acc = fmaf(b.x, f[4], acc); acc = fmaf(b.y, f[5], acc); acc = fmaf(b.z, f[6], acc); acc = fmaf(b.w, f[7], acc);
            acc += dpp_quad_xor1(acc);
            acc += dpp_quad_xor2(acc);
            acc += __shfl_xor(acc, 4, 64);
            if (fy == 0) partial[(t * 2 + net) * 16 + k] = acc;
        }
        // rows / planes / feat are rewritten behind the next iteration's barriers
    }
}

// 16 threads per grid: thread k adds the tiles' partial sums of output k of both nets in tile order
__global__ __launch_bounds__(256) void rnd_tile_finish_kernel(const float *__restrict__ partial, const float *__restrict__ pp,
                                                              const float *__restrict__ rp, float *__restrict__ bonus,
                                                              float *__restrict__ pred_out, float *__restrict__ target_out,
                                                              float *__restrict__ reward, float reward_scale, int64_t n,
                                                              int tiles, int dfeat) {
    const int64_t g = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int k = threadIdx.x & 15;
    float sq = 0.0f;
    if (g < n) {
        float p = 0.0f, r = 0.0f;
        for (int t = 0; t < tiles; ++t) {
            p += partial[((g * tiles + t) * 2 + 0) * 16 + k];
            r += partial[((g * tiles + t) * 2 + 1) * 16 + k];
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
    if (g < n && k == 0) {
        const float b = sq * (1.0f / 16.0f);
        bonus[g] = b;
        if (reward) reward[g] += reward_scale * b;        // mcl.py:403-405
    }
}

void rnd2d_prepare_tables(const float *pp, const float *rp, float *ws, hipStream_t s);

bool rnd2d_tiled_applies(int h, int w) { return h == w && h > 64 && (h & 63) == 0; }

// workspace: the fused tables (kFusedWsTotal floats) + 32 partial sums per tile
int64_t rnd2d_tiled_workspace_floats(int64_t n, int size) {
    const int64_t tiles = (int64_t)(size >> 6) * (size >> 6);
    return kFusedWsTotal + n * tiles * 32;
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
    if (packed)
        hipLaunchKernelGGL(rnd_tile_kernel<true>, dim3(blocks), dim3(256), 0, s, obs, ws, pp, rp, partial, n, size);
    else
        hipLaunchKernelGGL(rnd_tile_kernel<false>, dim3(blocks), dim3(256), 0, s, obs, ws, pp, rp, partial, n, size);
    const int d8 = size >> 3;
    hipLaunchKernelGGL(rnd_tile_finish_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, s, partial, pp, rp, bonus,
                       pred_out, target_out, reward, reward_scale, n, tiles, d8 * d8);
    return check_launch("carle_rnd2d_bonus(tiled)");
}

}  // namespace carle

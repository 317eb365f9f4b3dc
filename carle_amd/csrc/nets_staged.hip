// nets_staged.hip -- RND2D / AE2D forward passes (eval mode) as staged HIP kernels, any
// H == W that is a multiple of 8.  fp32 direct convolution; intermediates live in a
// caller-provided workspace.  This is the size-general path; 64x64 grids take the fused
// LDS kernels of nets_fused.hip.
//
// Networks (riveSunder/carle, carle/mcl.py; Dropout = identity in eval mode):
//   RND2D predictor  mcl.py:262-276  conv(1->4,3x3,p1) relu pool2 pool2 conv(4->1,3x3,p1) relu pool2
//                                    flatten linear(D->16) tanh,   D = (H/8)(W/8)
//   RND2D random     mcl.py:285-296  conv(1->2) relu pool2 pool2 conv(2->1) relu pool2 flatten linear tanh
//   RND2D bonus      mcl.py:343-352  mean_k (target_k - prediction_k)^2
//   AE2D predictor   mcl.py:452-467  conv(1->4) relu pool2 conv(4->2) relu pool2
//                                    convT(2->1,k4,s2,p1) relu convT(1->1,k4,s2,p1) sigmoid
//   AE2D bonus       mcl.py:504-512  mean_{h,w} (obs - prediction)^2
// relu and max-pool commute, and two stacked 2x2 pools are one 4x4 pool.
//
// Parameter blobs (flat fp32, state_dict order of each nn.Sequential):
//   RND predictor: conv1.w[4][1][3][3] conv1.b[4] conv2.w[1][4][3][3] conv2.b[1] lin.w[16][D] lin.b[16]
//   RND random   : conv1.w[2][1][3][3] conv1.b[2] conv2.w[1][2][3][3] conv2.b[1] lin.w[16][D] lin.b[16]
//   AE predictor : conv1.w[4][1][3][3] b[4] conv2.w[2][4][3][3] b[2] convT1.w[2][1][4][4] b[1]
//                  convT2.w[1][1][4][4] b[1]                                          (164 floats)
#include "common.hpp"
#include "train_bodies.hpp"

namespace carle {
// A padded read: zero outside the plane.  The load is UNCONDITIONAL (index 0 of the plane when outside): behind a
// load inside a branch or a select the compiler places the consumer in the branch and waits for that one load --
// a dependent memory round trip per tap (round 2, seen in the disassembly of the staged kernels).
__device__ __forceinline__ float ldz(const float *__restrict__ p, int64_t i, bool ok) {
    const float t = p[ok ? i : 0];
    return ok ? t : 0.0f;
}
}  // namespace carle

namespace carle {

static constexpr int kRndDim = 16;

// ---------------------------------------------------------------------------------------------
// RND2D stage 1: conv1 (6 channels: 4 predictor + 2 random) + relu + 4x4 max-pool
// one thread per (grid, pooled pixel)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rnd_conv1_pool4_kernel(
    const float *__restrict__ obs, const float *__restrict__ pp, const float *__restrict__ rp,
    float *__restrict__ ws1, int64_t n, int h, int w) {
    const int h4 = h >> 2, w4 = w >> 2;
    const int64_t total = n * (int64_t)h4 * w4;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int px = (int)(idx % w4), py = (int)((idx / w4) % h4);
    const int64_t g = idx / ((int64_t)h4 * w4);
    const float *go = obs + g * (int64_t)h * w;
    float patch[6][6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const int y = 4 * py - 1 + r;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int x = 4 * px - 1 + c;
            patch[r][c] = ldz(go, y * w + x, y >= 0 && y < h && x >= 0 && x < w);
        }
    }
#pragma unroll
    for (int ch = 0; ch < 6; ++ch) {
        const float *wt = ch < 4 ? pp + ch * 9 : rp + (ch - 4) * 9;
        const float bias = ch < 4 ? pp[36 + ch] : rp[18 + (ch - 4)];
        float k[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) k[t] = wt[t];
        float m = 0.0f;                                   // relu floor
#pragma unroll
        for (int dy = 0; dy < 4; ++dy)
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) {
                float acc = 0.0f;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
                        acc = fmaf(k[ky * 3 + kx], patch[dy + ky][dx + kx], acc);
                m = fmaxf(m, acc + bias);
            }
        ws1[((g * 6 + ch) * h4 + py) * (int64_t)w4 + px] = m;
    }
}

// RND2D stage 2: conv2 (C->1, 3x3, zero pad) + relu + 2x2 max-pool; thread per (grid, net, F pixel)
__global__ __launch_bounds__(256) void rnd_conv2_pool2_kernel(
    const float *__restrict__ ws1, const float *__restrict__ pp, const float *__restrict__ rp,
    float *__restrict__ ws2, int64_t n, int h, int w) {
    const int h4 = h >> 2, w4 = w >> 2, h8 = h >> 3, w8 = w >> 3;
    const int d = h8 * w8;
    const int64_t total = n * 2 * (int64_t)d;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int fx = (int)(idx % w8), fy = (int)((idx / w8) % h8);
    const int net = (int)((idx / d) % 2);                 // 0 = predictor, 1 = random
    const int64_t g = idx / (2 * (int64_t)d);
    const int nch = net == 0 ? 4 : 2, chbase = net == 0 ? 0 : 4;
    const float *wt = net == 0 ? pp + 40 : rp + 20;       // conv2.w
    const float bias = net == 0 ? pp[76] : rp[38];        // conv2.b
    float o[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    for (int c = 0; c < nch; ++c) {
        const float *pl = ws1 + (g * 6 + chbase + c) * (int64_t)h4 * w4;
        float patch[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int y = 2 * fy - 1 + r;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const int x = 2 * fx - 1 + cc;
                patch[r][cc] = ldz(pl, y * w4 + x, y >= 0 && y < h4 && x >= 0 && x < w4);
            }
        }
#pragma unroll
        for (int oy = 0; oy < 2; ++oy)
#pragma unroll
            for (int ox = 0; ox < 2; ++ox)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
                        o[oy][ox] = fmaf(wt[c * 9 + ky * 3 + kx], patch[oy + ky][ox + kx], o[oy][ox]);
    }
    const float m = fmaxf(fmaxf(o[0][0], o[0][1]), fmaxf(o[1][0], o[1][1])) + bias;
    ws2[(g * 2 + net) * (int64_t)d + fy * w8 + fx] = fmaxf(m, 0.0f);
}

struct __attribute__((aligned(4))) F4U { float x, y, z, w; };   // 16 bytes at 4-byte alignment: the weight rows sit
                                                                 // 77 / 39 floats into the state_dict-ordered blobs

// RND2D stage 3: linear(D->16) + tanh for both nets, bonus = mean_k (t_k - p_k)^2
// one 256-thread workgroup per grid: thread = (net, k, eighth of D); every thread streams its own
// contiguous slice of the weight row (float4 loads), the eight partial sums of an output are
// combined in a fixed order by shuffles.  (One wavefront per grid with strided weight reads took
// 63 us at N=16, 256x256: the reference's default geometry.)
__global__ __launch_bounds__(256) void rnd_linear_bonus_kernel(
    const float *__restrict__ ws2, const float *__restrict__ pp, const float *__restrict__ rp,
    float *__restrict__ bonus, float *__restrict__ pred_out, float *__restrict__ target_out,
    float *__restrict__ reward, float reward_scale, int64_t n, int d) {
    __shared__ float s_out[2][kRndDim];
    const int64_t g = blockIdx.x;
    const int t = threadIdx.x;
    const int seg = t & 7, k = (t >> 3) & 15, net = t >> 7;
    const float *f = ws2 + (g * 2 + net) * (int64_t)d;
    const float *lw = (net == 0 ? pp + 77 : rp + 39) + (int64_t)k * d;
    const float *lb = (net == 0 ? pp + 77 : rp + 39) + (int64_t)kRndDim * d;
    // d = (H / 8)^2: a multiple of 64 for H a multiple of 64, but ODD for H = 8 * odd (120, 136, 264 ...): the eight
    // float4 slices cover d only when d is a multiple of 32; every other d goes through the scalar loop, whose last
    // slice takes the remainder.  (Until round 4 the test was `len % 4 == 0` alone: at d = 225, 289, 1089 ... -- len = 28,
    // 36, 136 -- the vector path dropped the last d % 8 inputs of every output; found by tools/fuzz_nets.py.)
    const int len = d >> 3, i0 = seg * len;
    float acc = 0.0f;
    // (the weight rows sit 77 / 39 floats into the state_dict-ordered blobs -- 4-byte aligned only: F4U below, as the wide
    // kernel reads them; with a float4 the alignment test never passed for the predictor and the vector path was dead)
    if (len > 0 && (d & 31) == 0 && ((uintptr_t)(f + i0) & 15u) == 0) {
        const F4U *w4 = reinterpret_cast<const F4U *>(lw + i0);
        const float4 *f4 = reinterpret_cast<const float4 *>(f + i0);
        for (int i = 0; i < (len >> 2); ++i) {
            const F4U a = w4[i];
            const float4 b = f4[i];
            acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc);
            acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
        }
    } else {
        const int i1 = seg == 7 ? d : i0 + len;           // the last slice takes the remainder
        for (int i = i0; i < i1; ++i) acc = fmaf(lw[i], f[i], acc);
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (seg == 0) {
        const float out = tanhf(acc + lb[k]);
        s_out[net][k] = out;
        if (net == 0 && pred_out) pred_out[g * kRndDim + k] = out;
        if (net == 1 && target_out) target_out[g * kRndDim + k] = out;
    }
    __syncthreads();
    if (t == 0) {
        float sq = 0.0f;
#pragma unroll
        for (int i = 0; i < kRndDim; ++i) {
            const float df = s_out[1][i] - s_out[0][i];
            sq = fmaf(df, df, sq);
        }
        const float b = sq * (1.0f / kRndDim);
        bonus[g] = b;
        if (reward) reward[g] = reward_accumulate(reward[g], reward_scale, b);        // mcl.py:403-405
    }
}

// The same for D a multiple of 128 (H = W a multiple of 32: the reference's default 256x256 has D = 1024):
// 1024 threads per grid, thread = (net, k, one of 32 slices of D), the slice's D/128 float4 pairs all in
// flight at once -- ONE memory latency instead of D/32 dependent ones (the 256-thread kernel above: 25 us at
// N=16, 256x256; this one: see profiles/r02_fwd256.txt).  Fixed summation order.
__global__ __launch_bounds__(1024) void rnd_linear_bonus_wide_kernel(
    const float *__restrict__ ws2, const float *__restrict__ pp, const float *__restrict__ rp,
    float *__restrict__ bonus, float *__restrict__ pred_out, float *__restrict__ target_out,
    float *__restrict__ reward, float reward_scale, int64_t n, int d) {
    __shared__ float s_out[2][kRndDim];
    const int64_t g = blockIdx.x;
    const int t = threadIdx.x;
    const int seg = t & 31, k = (t >> 5) & 15, net = t >> 9;
    const int len4 = d >> 7;                              // float4 pairs per slice
    const float4 *f4 = reinterpret_cast<const float4 *>(ws2 + (g * 2 + net) * (int64_t)d) + seg * len4;
    const float *base = net == 0 ? pp + 77 : rp + 39;
    const F4U *w4 = reinterpret_cast<const F4U *>(base + (int64_t)k * d) + seg * len4;
    float acc = 0.0f;
    for (int i0 = 0; i0 < len4; i0 += 8) {
        F4U a[8];
        float4 b[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i0 + i < len4) { a[i] = w4[i0 + i]; b[i] = f4[i0 + i]; }
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i0 + i < len4) {
                acc = fmaf(a[i].x, b[i].x, acc); acc = fmaf(a[i].y, b[i].y, acc);
                acc = fmaf(a[i].z, b[i].z, acc); acc = fmaf(a[i].w, b[i].w, acc);
            }
    }
#pragma unroll
    for (int sh = 1; sh < 32; sh <<= 1) acc += __shfl_xor(acc, sh, 64);
    if (seg == 0) {
        const float out = tanhf(acc + base[(int64_t)kRndDim * d + k]);
        s_out[net][k] = out;
        if (net == 0 && pred_out) pred_out[g * kRndDim + k] = out;
        if (net == 1 && target_out) target_out[g * kRndDim + k] = out;
    }
    __syncthreads();
    if (t == 0) {
        float sq = 0.0f;
#pragma unroll
        for (int i = 0; i < kRndDim; ++i) {
            const float df = s_out[1][i] - s_out[0][i];
            sq = fmaf(df, df, sq);
        }
        const float b = sq * (1.0f / kRndDim);
        bonus[g] = b;
        if (reward) reward[g] = reward_accumulate(reward[g], reward_scale, b);        // mcl.py:403-405
    }
}

// launches the linear stage: the wide kernel where its float4 slices exist and are aligned
static inline void launch_rnd_linear(const float *ws2, const float *pp, const float *rp, float *bonus, float *pred_out,
                                     float *target_out, float *reward, float reward_scale, int64_t n, int d,
                                     hipStream_t s) {
    const bool wide = (d & 127) == 0 && ((uintptr_t)ws2 & 15u) == 0;
    if (wide)
        hipLaunchKernelGGL(rnd_linear_bonus_wide_kernel, dim3((unsigned)n), dim3(1024), 0, s, ws2, pp, rp, bonus,
                           pred_out, target_out, reward, reward_scale, n, d);
    else
        hipLaunchKernelGGL(rnd_linear_bonus_kernel, dim3((unsigned)n), dim3(256), 0, s, ws2, pp, rp, bonus, pred_out,
                           target_out, reward, reward_scale, n, d);
}

// ---------------------------------------------------------------------------------------------
// AE2D stage 1: conv1 (1->4) + relu + 2x2 pool; thread per (grid, A1 pixel)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ae_conv1_pool2_kernel(
    const float *__restrict__ obs, const float *__restrict__ ap, float *__restrict__ a1,
    int64_t n, int h, int w) {
    const int h2 = h >> 1, w2 = w >> 1;
    const int64_t total = n * (int64_t)h2 * w2;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int px = (int)(idx % w2), py = (int)((idx / w2) % h2);
    const int64_t g = idx / ((int64_t)h2 * w2);
    const float *go = obs + g * (int64_t)h * w;
    float patch[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int y = 2 * py - 1 + r;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int x = 2 * px - 1 + c;
            patch[r][c] = ldz(go, y * w + x, y >= 0 && y < h && x >= 0 && x < w);
        }
    }
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
        float m = 0.0f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                float acc = 0.0f;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
                        acc = fmaf(ap[ch * 9 + ky * 3 + kx], patch[dy + ky][dx + kx], acc);
                m = fmaxf(m, acc + ap[36 + ch]);
            }
        a1[((g * 4 + ch) * h2 + py) * (int64_t)w2 + px] = m;
    }
}

// AE2D stage 2: conv2 (4->2) + relu + 2x2 pool; thread per (grid, A2 pixel), both channels
__global__ __launch_bounds__(256) void ae_conv2_pool2_kernel(
    const float *__restrict__ a1, const float *__restrict__ ap, float *__restrict__ a2,
    int64_t n, int h, int w) {
    const int h2 = h >> 1, w2 = w >> 1, h4 = h >> 2, w4 = w >> 2;
    const int64_t total = n * (int64_t)h4 * w4;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int px = (int)(idx % w4), py = (int)((idx / w4) % h4);
    const int64_t g = idx / ((int64_t)h4 * w4);
    const float *wt = ap + 40;                            // conv2.w[2][4][3][3]
    float o[2][2][2] = {};
    for (int c = 0; c < 4; ++c) {
        const float *pl = a1 + (g * 4 + c) * (int64_t)h2 * w2;
        float patch[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int y = 2 * py - 1 + r;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const int x = 2 * px - 1 + cc;
                patch[r][cc] = ldz(pl, y * w2 + x, y >= 0 && y < h2 && x >= 0 && x < w2);
            }
        }
#pragma unroll
        for (int oc = 0; oc < 2; ++oc)
#pragma unroll
            for (int oy = 0; oy < 2; ++oy)
#pragma unroll
                for (int ox = 0; ox < 2; ++ox)
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx)
                            o[oc][oy][ox] = fmaf(wt[(oc * 4 + c) * 9 + ky * 3 + kx],
                                                 patch[oy + ky][ox + kx], o[oc][oy][ox]);
    }
#pragma unroll
    for (int oc = 0; oc < 2; ++oc) {
        const float m = fmaxf(fmaxf(o[oc][0][0], o[oc][0][1]), fmaxf(o[oc][1][0], o[oc][1][1])) +
                        ap[112 + oc];
        a2[((g * 2 + oc) * h4 + py) * (int64_t)w4 + px] = fmaxf(m, 0.0f);
    }
}

// transposed conv k4 s2 p1: output row y takes input rows iy with ky = y + 1 - 2*iy in [0,3]:
//   y even (2m):   (iy=m, ky=1), (iy=m-1, ky=3);   y odd (2m+1): (iy=m, ky=2), (iy=m+1, ky=0)
__device__ __forceinline__ void convt_taps(int y, int &i0, int &k0, int &i1, int &k1) {
    const int m = y >> 1;
    if (y & 1) { i0 = m; k0 = 2; i1 = m + 1; k1 = 0; }
    else       { i0 = m; k0 = 1; i1 = m - 1; k1 = 3; }
}

// AE2D stage 3: convT1 (2->1) + relu; thread per (grid, A3 pixel), A3 is (H/2 x W/2)
__global__ __launch_bounds__(256) void ae_convt1_kernel(
    const float *__restrict__ a2, const float *__restrict__ ap, float *__restrict__ a3,
    int64_t n, int h, int w) {
    const int h2 = h >> 1, w2 = w >> 1, h4 = h >> 2, w4 = w >> 2;
    const int64_t total = n * (int64_t)h2 * w2;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % w2), y = (int)((idx / w2) % h2);
    const int64_t g = idx / ((int64_t)h2 * w2);
    const float *wt = ap + 114;                           // convT1.w[2][1][4][4]
    int iy[2], ky[2], ix[2], kx[2];
    convt_taps(y, iy[0], ky[0], iy[1], ky[1]);
    convt_taps(x, ix[0], kx[0], ix[1], kx[1]);
    float acc = 0.0f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float *pl = a2 + (g * 2 + c) * (int64_t)h4 * w4;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                acc = fmaf(ldz(pl, iy[a] * w4 + ix[b], iy[a] >= 0 && iy[a] < h4 && ix[b] >= 0 && ix[b] < w4),
                           wt[c * 16 + ky[a] * 4 + kx[b]], acc);
    }
    a3[idx] = fmaxf(acc + ap[146], 0.0f);
}

// AE2D stage 4: convT2 (1->1) + sigmoid, squared error against obs, mean over the grid.
// Workgroup (g, c) covers pixels [c*chunk, (c+1)*chunk) of grid g and leaves one partial sum; the
// finish kernel adds the partials of a grid in a fixed order (deterministic).  (One workgroup per
// whole grid took 223 us at N=16, 256x256: the reference's default geometry.)
static constexpr int kMseChunk = 1024;                 // 4 pixels per thread (4096: 16 dependent iterations, 16 us at N=16 256x256)

__global__ __launch_bounds__(256) void ae_convt2_mse_kernel(
    const float *__restrict__ obs, const float *__restrict__ a3, const float *__restrict__ ap,
    float *__restrict__ partial, float *__restrict__ recon, int chunks, int h, int w) {
    __shared__ float red[4];
    const int h2 = h >> 1, w2 = w >> 1;
    const int64_t g = blockIdx.x / chunks;
    const int c = (int)(blockIdx.x % chunks);
    const float *wt = ap + 147;                           // convT2.w[1][1][4][4]
    const float bias = ap[163];
    const float *pl = a3 + g * (int64_t)h2 * w2;
    const float *go = obs + g * (int64_t)h * w;
    const int p_end = min((c + 1) * kMseChunk, h * w);
    float sum = 0.0f;
#pragma unroll 4
    for (int p = c * kMseChunk + threadIdx.x; p < p_end; p += 256) {
        const int x = p % w, y = p / w;
        int iy[2], ky[2], ix[2], kx[2];
        convt_taps(y, iy[0], ky[0], iy[1], ky[1]);
        convt_taps(x, ix[0], kx[0], ix[1], kx[1]);
        float acc = 0.0f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                acc = fmaf(ldz(pl, iy[a] * w2 + ix[b], iy[a] >= 0 && iy[a] < h2 && ix[b] >= 0 && ix[b] < w2),
                           wt[ky[a] * 4 + kx[b]], acc);
        const float pr = 1.0f / (1.0f + expf(-(acc + bias)));
        if (recon) recon[g * (int64_t)h * w + p] = pr;
        const float df = go[p] - pr;
        sum = fmaf(df, df, sum);
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) sum += __shfl_xor(sum, s, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void ae_mse_finish_kernel(const float *__restrict__ partial,
                                                            float *__restrict__ bonus,
                                                            float *__restrict__ reward, float reward_scale,
                                                            int64_t n, int chunks, float inv_pixels) {
    ae_mse_finish_body(partial, bonus, reward, reward_scale, n, chunks, inv_pixels,
                       (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

static inline int check_net_dims(int64_t n, int h, int w, const char *who) {
    if (n < 0 || h != w || h < 8 || (h % 8) != 0) {
        static thread_local char buf[160];
        snprintf(buf, sizeof buf, "%s: grids must be square with H == W a multiple of 8", who);
        return fail_arg(CARLE_E_DIMS, buf);
    }
    return 0;
}

static inline unsigned blocks_for(int64_t total) { return (unsigned)((total + 255) / 256); }

}  // namespace carle

#include "nets_staged_bwd.hpp"      // staged backward kernels: launch the forward kernels above

namespace carle {
int64_t rnd2d_fused_workspace_floats();
int rnd2d_fused64(const void *obs, int obs_mode, int64_t n, const float *pp, const float *rp,
                  float *ws, float *bonus, float *pred_out, float *target_out, float *reward,
                  float reward_scale, bool tables_valid, hipStream_t s);
int rnd2d_train64_rows(int64_t n);
int rnd2d_train64(const void *obs, int obs_mode, int64_t n, const float *pp, const float *rp, float *ws,
                  float *bonus, float *reward, float reward_scale, float grad_scale, float *partial,
                  float *grad, bool tables_valid, hipStream_t s);
bool rnd2d_tiled_applies(int h, int w);
int64_t rnd2d_tiled_workspace_floats(int64_t n, int size);
int rnd2d_tiled(const void *obs, bool packed, int64_t n, int size, const float *pp, const float *rp, float *ws,
                float *bonus, float *pred_out, float *target_out, float *reward, float reward_scale, bool tables_valid,
                hipStream_t s);
int rnd2d_tiled_bwd_rows(int64_t n, int size);
int rnd2d_tiled_train(const void *obs, bool packed, int64_t n, int size, const float *pp, const float *rp, float *ws,
                      float *featbuf, float *pred, float *targ, float *rows, float *bonus, float *reward,
                      float reward_scale, float grad_scale, bool tables_valid, hipStream_t s);
int ae2d_tiled_train_rows(int64_t n, int size);
int ae2d_tiled_train(const void *obs, bool packed, int64_t n, int size, const float *ap, float *ws, float *rows,
                     float grad_scale, bool tables_valid, hipStream_t s, const uint64_t *tgt = nullptr);
int64_t ae2d_tiled_workspace_floats(int64_t n, int size);
int ae2d_tiled(const void *obs, bool packed, int64_t n, int size, const float *ap, float *ws, float *recon,
               bool tables_valid, hipStream_t s, const uint64_t *tgt = nullptr);
int ae2d_train64_rows(int64_t n);
int ae2d_train64(const void *obs, bool packed, int64_t n, const float *ap, float *ws, float *bonus,
                 float *reward, float reward_scale, float grad_scale, float *partial, float *grad,
                 bool tables_valid, hipStream_t s, const uint64_t *tgt = nullptr);
int64_t ae2d_fused_workspace_floats();
int ae2d_fused64(const void *obs, bool packed, int64_t n, const float *ap, float *ws, float *bonus,
                 float *recon, float *reward, float reward_scale, bool tables_valid, hipStream_t s,
                 const uint64_t *tgt = nullptr, double *partials = nullptr, int32_t *status = nullptr);
int64_t ae2d_fused64_blocks(int64_t n);
}

using namespace carle;

extern "C" int64_t carle_rnd2d_workspace_floats(int64_t n, int h, int w) {
    if (n < 0 || h < 8 || w < 8) return 0;
    const int64_t staged = n * (6 * (int64_t)(h / 4) * (w / 4) + 2 * (int64_t)(h / 8) * (w / 8));
    int64_t fused = rnd2d_fused_workspace_floats();
    if (rnd2d_tiled_applies(h, w)) fused = rnd2d_tiled_workspace_floats(n, h);
    return staged > fused ? staged : fused;
}

namespace carle {
int rnd_ae_tiled(const void *obs, bool packed, int64_t n, int size, const float *pp, const float *rp, float *rws, float *rbonus,
                 float rscale, bool rnd_tables_valid, const float *ap, float *aws, float *abonus, float ascale,
                 bool ae_tables_valid, float *reward, double *mom_partials, int32_t *status, hipStream_t s);
int64_t rnd_ae_tiled_blocks(int64_t n, int size);
int rnd_ae_tiled_train(const void *obs, bool packed, int64_t n, int size, const float *pp, const float *rp, float *rws,
                       float *rscratch, float *rgrad, float *rbonus, float rscale, float rgrad_scale, bool rvalid,
                       const float *ap, float *aws, float *arows, float *agrad, float *abonus, float ascale, float agrad_scale,
                       bool avalid, float *reward, double *mom_partials, int32_t *status, hipStream_t s);
}

// RND2D.get_bonus AND AE2D.get_bonus of the same observations (carle/mcl.py:343-352, 504-512) + both wrappers' reward
// updates (mcl.py:403-405, RND2D first) in one launch: binary observations, H == W a multiple of 64 above 64 (the tiled
// kernels; the reference's default 256x256).  obs_fmt: CARLE_FMT_F32 (values exactly 0 / 1) or CARLE_FMT_B1.
// Workspaces, parameter blobs, bonuses and TABLES_VALID flags exactly as for carle_rnd2d_bonus_* / carle_ae2d_bonus_*.
extern "C" int carle_rnd_ae_bonus(const void *obs, int obs_fmt, int64_t n, int h, int w, const float *rnd_pred_params,
                                  const float *rnd_rand_params, float *rnd_workspace, float *rnd_bonus, float rnd_scale,
                                  int rnd_flags, const float *ae_params, float *ae_workspace, float *ae_bonus, float ae_scale,
                                  int ae_flags, float *reward_accum, double *moment_partials, int32_t *status,
                                  void *stream) {
    if (!obs || !rnd_pred_params || !rnd_rand_params || !rnd_workspace || !rnd_bonus || !ae_params || !ae_workspace || !ae_bonus)
        return fail_arg(CARLE_E_NULL, "carle_rnd_ae_bonus: NULL pointer");
    if (obs_fmt != CARLE_FMT_F32 && obs_fmt != CARLE_FMT_B1)
        return fail_arg(CARLE_E_FORMAT, "carle_rnd_ae_bonus: obs_fmt must be CARLE_FMT_F32 or CARLE_FMT_B1");
    if (n < 0 || !rnd2d_tiled_applies(h, w))
        return fail_arg(CARLE_E_DIMS, "carle_rnd_ae_bonus: binary grids with H == W a multiple of 64 above 64 (use the "
                                      "per-net entry points otherwise)");
    if ((moment_partials != nullptr) != (status != nullptr) || (status && (!reward_accum || !aligned16(status))))
        return fail_arg(CARLE_E_NULL, "carle_rnd_ae_bonus: moment_partials, status (16-byte aligned) and reward_accum go together");
    if (n == 0) return 0;
    return rnd_ae_tiled(obs, obs_fmt == CARLE_FMT_B1, n, h, rnd_pred_params, rnd_rand_params, rnd_workspace, rnd_bonus,
                        rnd_scale, (rnd_flags & CARLE_NET_TABLES_VALID) != 0, ae_params, ae_workspace, ae_bonus, ae_scale,
                        (ae_flags & CARLE_NET_TABLES_VALID) != 0, reward_accum, moment_partials, status,
                        static_cast<hipStream_t>(stream));
}

// fp64 values behind `moment_partials` of the two entry points around this one (one (sum, sum of squares) pair per
// workgroup of their finish kernels)
extern "C" int64_t carle_rnd_ae_partials_doubles(int64_t n, int h, int w) {
    if (n < 0 || !rnd2d_tiled_applies(h, w)) return 0;
    const int64_t fwd = rnd_ae_tiled_blocks(n, h), train = (n + 1023) / 1024;
    return 2 * (fwd > train ? fwd : train);
}

// The same stack in TRAINING mode (carle/train_mcl.py:34-38, 64-69): both wrappers' bonuses, reward updates and the
// accumulation of both predictors' gradients of one step -- what carle_rnd2d_train_* followed by carle_ae2d_train_* do --
// in four launches instead of nine.  Scratch and gradient buffers exactly as for those two entry points
// (carle_rnd2d_train_scratch_floats / carle_ae2d_train_scratch_floats); bit-identical results.
extern "C" int carle_rnd_ae_train(const void *obs, int obs_fmt, int64_t n, int h, int w, const float *rnd_pred_params,
                                  const float *rnd_rand_params, float *rnd_workspace, float *rnd_scratch, float *rnd_grad,
                                  float *rnd_bonus, float rnd_scale, float rnd_grad_scale, int rnd_flags,
                                  const float *ae_params, float *ae_workspace, float *ae_scratch, float *ae_grad,
                                  float *ae_bonus, float ae_scale, float ae_grad_scale, int ae_flags, float *reward_accum,
                                  double *moment_partials, int32_t *status, void *stream) {
    if (!obs || !rnd_pred_params || !rnd_rand_params || !rnd_workspace || !rnd_scratch || !rnd_grad || !rnd_bonus ||
        !ae_params || !ae_workspace || !ae_scratch || !ae_grad || !ae_bonus)
        return fail_arg(CARLE_E_NULL, "carle_rnd_ae_train: NULL pointer");
    if (obs_fmt != CARLE_FMT_F32 && obs_fmt != CARLE_FMT_B1)
        return fail_arg(CARLE_E_FORMAT, "carle_rnd_ae_train: obs_fmt must be CARLE_FMT_F32 or CARLE_FMT_B1");
    if (n < 0 || !rnd2d_tiled_applies(h, w))
        return fail_arg(CARLE_E_DIMS, "carle_rnd_ae_train: binary grids with H == W a multiple of 64 above 64 (use the "
                                      "per-net entry points otherwise)");
    if ((moment_partials != nullptr) != (status != nullptr) || (status && (!reward_accum || !aligned16(status))))
        return fail_arg(CARLE_E_NULL, "carle_rnd_ae_train: moment_partials, status (16-byte aligned) and reward_accum go together");
    if (n == 0) return 0;
    return rnd_ae_tiled_train(obs, obs_fmt == CARLE_FMT_B1, n, h, rnd_pred_params, rnd_rand_params, rnd_workspace, rnd_scratch,
                              rnd_grad, rnd_bonus, rnd_scale, rnd_grad_scale, (rnd_flags & CARLE_NET_TABLES_VALID) != 0,
                              ae_params, ae_workspace, ae_scratch, ae_grad, ae_bonus, ae_scale, ae_grad_scale,
                              (ae_flags & CARLE_NET_TABLES_VALID) != 0, reward_accum, moment_partials, status,
                              static_cast<hipStream_t>(stream));
}

extern "C" int carle_rnd2d_bonus_b1(const uint64_t *obs, int64_t n, int h, int w,
                                    const float *pred_params, const float *rand_params,
                                    float *workspace, float *bonus, float *pred_out,
                                    float *target_out, float *reward_accum, float reward_scale,
                                    int flags, void *stream) {
    if (!obs || !pred_params || !rand_params || !workspace || !bonus)
        return fail_arg(CARLE_E_NULL, "carle_rnd2d_bonus_b1: NULL pointer");
    if (n >= 0 && rnd2d_tiled_applies(h, w)) {
        if (n == 0) return 0;
        return rnd2d_tiled(obs, true, n, h, pred_params, rand_params, workspace, bonus, pred_out, target_out,
                           reward_accum, reward_scale, (flags & CARLE_NET_TABLES_VALID) != 0,
                           static_cast<hipStream_t>(stream));
    }
    if (h != 64 || w != 64 || n < 0)
        return fail_arg(CARLE_E_DIMS, "carle_rnd2d_bonus_b1: bit-packed observations are supported "
                                      "for square grids with H = W a multiple of 64");
    if (n == 0) return 0;
    return rnd2d_fused64(obs, 1, n, pred_params, rand_params, workspace, bonus, pred_out,
                         target_out, reward_accum, reward_scale, (flags & CARLE_NET_TABLES_VALID) != 0,
                         static_cast<hipStream_t>(stream));
}

extern "C" int carle_rnd2d_bonus_f32(const float *obs, int64_t n, int h, int w,
                                     const float *pred_params, const float *rand_params,
                                     float *workspace, float *bonus, float *pred_out,
                                     float *target_out, float *reward_accum, float reward_scale,
                                     int flags, void *stream) {
    if (!obs || !pred_params || !rand_params || !workspace || !bonus)
        return fail_arg(CARLE_E_NULL, "carle_rnd2d_bonus_f32: NULL pointer");
    if (int e = check_net_dims(n, h, w, "carle_rnd2d_bonus_f32")) return e;
    if (n == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (h == 64 && w == 64 && (flags & CARLE_NET_OBS_BINARY) && aligned16(obs))
        return rnd2d_fused64(obs, 0, n, pred_params, rand_params, workspace, bonus, pred_out,
                             target_out, reward_accum, reward_scale,
                             (flags & CARLE_NET_TABLES_VALID) != 0, s);
    if ((flags & CARLE_NET_OBS_BINARY) && rnd2d_tiled_applies(h, w))
        return rnd2d_tiled(obs, false, n, h, pred_params, rand_params, workspace, bonus, pred_out, target_out,
                           reward_accum, reward_scale, (flags & CARLE_NET_TABLES_VALID) != 0, s);
    const int64_t p1 = n * (int64_t)(h / 4) * (w / 4);
    const int d = (h / 8) * (w / 8);
    if (p1 * 6 / 256 > 0x7FFFFFFFll) return fail_arg(CARLE_E_DIMS, "carle_rnd2d_bonus_f32: batch too large");
    float *ws1 = workspace, *ws2 = workspace + 6 * p1;
    hipLaunchKernelGGL(rnd_conv1_pool4_kernel, dim3(blocks_for(p1)), dim3(256), 0, s, obs,
                       pred_params, rand_params, ws1, n, h, w);
    hipLaunchKernelGGL(rnd_conv2_pool2_kernel, dim3(blocks_for(n * 2 * (int64_t)d)), dim3(256), 0, s,
                       ws1, pred_params, rand_params, ws2, n, h, w);
    launch_rnd_linear(ws2, pred_params, rand_params, bonus, pred_out, target_out, reward_accum, reward_scale, n, d, s);
    return check_launch("carle_rnd2d_bonus_f32");
}

// AE2D forward through the per-layer kernels: decoder(obs), squared error against `target` (= obs for AE2D.get_bonus,
// a later frame for PredictionBonus), any H = W multiple of 8, any fp32 values
static int ae2d_bonus_staged(const float *obs, const float *target, int64_t n, int h, int w, const float *params,
                             float *workspace, float *bonus, float *recon_out, float *reward_accum, float reward_scale,
                             hipStream_t s, const char *who) {
    const int64_t q = (int64_t)(h / 2) * (w / 2);
    const int chunks = (h * w + kMseChunk - 1) / kMseChunk;
    if (n * q / 256 > 0x7FFFFFFFll || n * chunks > 0x7FFFFFFFll) {
        static thread_local char buf[96];
        snprintf(buf, sizeof buf, "%s: batch too large", who);
        return fail_arg(CARLE_E_DIMS, buf);
    }
    float *a1 = workspace, *a2 = a1 + n * 4 * q, *a3 = a2 + n * 2 * (q / 4);
    hipLaunchKernelGGL(ae_conv1_pool2_kernel, dim3(blocks_for(n * q)), dim3(256), 0, s, obs, params, a1, n, h, w);
    hipLaunchKernelGGL(ae_conv2_pool2_kernel, dim3(blocks_for(n * (q / 4))), dim3(256), 0, s, a1, params, a2, n, h, w);
    hipLaunchKernelGGL(ae_convt1_kernel, dim3(blocks_for(n * q)), dim3(256), 0, s, a2, params, a3, n, h, w);
    float *part = a3 + n * q;
    hipLaunchKernelGGL(ae_convt2_mse_kernel, dim3((unsigned)(n * chunks)), dim3(256), 0, s, target, a3, params, part,
                       recon_out, chunks, h, w);
    hipLaunchKernelGGL(ae_mse_finish_kernel, dim3(blocks_for(n)), dim3(256), 0, s, part, bonus, reward_accum,
                       reward_scale, n, chunks, 1.0f / (float)(h * w));
    return check_launch(who);
}

extern "C" int64_t carle_ae2d_workspace_floats(int64_t n, int h, int w) {
    if (n < 0 || h < 8 || w < 8) return 0;
    const int64_t q = (int64_t)(h / 2) * (w / 2);
    const int64_t chunks = ((int64_t)h * w + 1023) / 1024;           // = kMseChunk
    const int64_t staged = n * (4 * q + 2 * (q / 4) + q + chunks);   // A1 (4ch) + A2 (2ch) + A3 + MSE partials
    int64_t fused = ae2d_fused_workspace_floats();
    if (rnd2d_tiled_applies(h, w)) fused = ae2d_tiled_workspace_floats(n, h);
    return staged > fused ? staged : fused;
}

extern "C" int carle_ae2d_bonus_b1(const uint64_t *obs, int64_t n, int h, int w, const float *params,
                                   float *workspace, float *bonus, float *recon_out,
                                   float *reward_accum, float reward_scale, int flags, void *stream) {
    if (!obs || !params || !workspace || !bonus)
        return fail_arg(CARLE_E_NULL, "carle_ae2d_bonus_b1: NULL pointer");
    if (n >= 0 && rnd2d_tiled_applies(h, w)) {
        if (n == 0) return 0;
        hipStream_t s = static_cast<hipStream_t>(stream);
        if (int e = ae2d_tiled(obs, true, n, h, params, workspace, recon_out, (flags & CARLE_NET_TABLES_VALID) != 0, s))
            return e;
        const int tiles = (h >> 6) * (h >> 6);
        hipLaunchKernelGGL(ae_mse_finish_kernel, dim3(blocks_for(n)), dim3(256), 0, s, workspace + 2048 + 128, bonus,
                           reward_accum, reward_scale, n, tiles, 1.0f / ((float)h * (float)w));
        return check_launch("carle_ae2d_bonus_b1");
    }
    if (h != 64 || w != 64 || n < 0)
        return fail_arg(CARLE_E_DIMS, "carle_ae2d_bonus_b1: bit-packed observations are supported "
                                      "for square grids with H = W a multiple of 64");
    if (recon_out && !aligned16(recon_out)) return fail_arg(CARLE_E_ALIGN, "carle_ae2d_bonus_b1: recon_out alignment");
    if (n == 0) return 0;
    return ae2d_fused64(obs, true, n, params, workspace, bonus, recon_out, reward_accum, reward_scale,
                        (flags & CARLE_NET_TABLES_VALID) != 0, static_cast<hipStream_t>(stream));
}

// carle_ae2d_bonus_b1 for an AE2D that is the OUTERMOST wrapper of a 64x64 stack whose rewards feed the running reward
// statistics (CARLE_ST_MOMENTS_*): reward_accum is required, and the kernel that finishes the rewards also leaves their
// (sum, sum of squares) per workgroup in moment_partials (carle_ae2d_stats_partials_doubles(n) doubles, caller-owned,
// alive until the next call on the status block) and registers them in `status` -- merged by that next call (or
// carle_env_settle) exactly as the one-launch RND2D step's are.  No stand-alone moments launch per step.
extern "C" int64_t carle_ae2d_stats_partials_doubles(int64_t n) { return n > 0 ? 2 * ae2d_fused64_blocks(n) : 2; }

extern "C" int carle_ae2d_bonus_b1_stats(const uint64_t *obs, int64_t n, int h, int w, const float *params, float *workspace,
                                         float *bonus, float *reward_accum, float reward_scale, int flags,
                                         double *moment_partials, int32_t *status, void *stream) {
    if (!obs || !params || !workspace || !bonus || !reward_accum || !moment_partials || !status)
        return fail_arg(CARLE_E_NULL, "carle_ae2d_bonus_b1_stats: NULL pointer");
    if (h != 64 || w != 64 || n <= 0) return fail_arg(CARLE_E_DIMS, "carle_ae2d_bonus_b1_stats: 64x64 bit-packed grids, n > 0");
    if (!aligned16(moment_partials) || !aligned16(status))
        return fail_arg(CARLE_E_ALIGN, "carle_ae2d_bonus_b1_stats: moment_partials / status must be 16-byte aligned");
    return ae2d_fused64(obs, true, n, params, workspace, bonus, nullptr, reward_accum, reward_scale,
                        (flags & CARLE_NET_TABLES_VALID) != 0, static_cast<hipStream_t>(stream), nullptr, moment_partials, status);
}

extern "C" int carle_ae2d_bonus_f32(const float *obs, int64_t n, int h, int w, const float *params,
                                    float *workspace, float *bonus, float *recon_out,
                                    float *reward_accum, float reward_scale, int flags, void *stream) {
    if (!obs || !params || !workspace || !bonus)
        return fail_arg(CARLE_E_NULL, "carle_ae2d_bonus_f32: NULL pointer");
    if (int e = check_net_dims(n, h, w, "carle_ae2d_bonus_f32")) return e;
    if (n == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (h == 64 && w == 64 && (flags & CARLE_NET_OBS_BINARY) && aligned16(obs) &&
        (!recon_out || aligned16(recon_out)))
        return ae2d_fused64(obs, false, n, params, workspace, bonus, recon_out, reward_accum,
                            reward_scale, (flags & CARLE_NET_TABLES_VALID) != 0, s);
    if ((flags & CARLE_NET_OBS_BINARY) && rnd2d_tiled_applies(h, w) && (!recon_out || aligned16(recon_out))) {
        if (int e = ae2d_tiled(obs, false, n, h, params, workspace, recon_out, (flags & CARLE_NET_TABLES_VALID) != 0, s))
            return e;
        const int tiles = (h >> 6) * (h >> 6);
        hipLaunchKernelGGL(ae_mse_finish_kernel, dim3(blocks_for(n)), dim3(256), 0, s, workspace + 2048 + 128, bonus,
                           reward_accum, reward_scale, n, tiles, 1.0f / ((float)h * (float)w));
        return check_launch("carle_ae2d_bonus_f32");
    }
    return ae2d_bonus_staged(obs, obs, n, h, w, params, workspace, bonus, recon_out, reward_accum, reward_scale, s,
                             "carle_ae2d_bonus_f32");
}

// PredictionBonus.get_bonus (carle/mcl.py:581-605): decoder(obs) against `target`, a later frame of the same universes
extern "C" int carle_ae2d_predict_bonus_f32(const float *obs, const float *target, int64_t n, int h, int w,
                                            const float *params, float *workspace, float *bonus, float *recon_out,
                                            void *stream) {
    if (!obs || !target || !params || !workspace || !bonus)
        return fail_arg(CARLE_E_NULL, "carle_ae2d_predict_bonus_f32: NULL pointer");
    if (int e = check_net_dims(n, h, w, "carle_ae2d_predict_bonus_f32")) return e;
    if (n == 0) return 0;
    return ae2d_bonus_staged(obs, target, n, h, w, params, workspace, bonus, recon_out, nullptr, 0.0f,
                             static_cast<hipStream_t>(stream), "carle_ae2d_predict_bonus_f32");
}

// ... and with the gradient of grad_scale/2 * sum_i sum_hw (target_i - decoder(obs_i))^2 accumulated into grad[0..164)
// (PredictionBonus.get_bonus_accumulate / get_bonus_update, mcl.py:644-677); buffers as carle_ae2d_train_f32's staged path
extern "C" int carle_ae2d_predict_train_f32(const float *obs, const float *target, int64_t n, int h, int w,
                                            const float *params, float *workspace, float *scratch, float *grad,
                                            float *bonus, float grad_scale, void *stream) {
    if (!obs || !target || !params || !workspace || !scratch || !grad || !bonus)
        return fail_arg(CARLE_E_NULL, "carle_ae2d_predict_train_f32: NULL pointer");
    if (int e = check_net_dims(n, h, w, "carle_ae2d_predict_train_f32")) return e;
    if (n == 0) return 0;
    return ae2d_train_staged(obs, target, n, h, w, params, workspace, scratch, grad, bonus, nullptr, 0.0f, grad_scale,
                             static_cast<hipStream_t>(stream));
}

// The same on BIT-PACKED frames (binary grids, H = W a multiple of 64: what CARLE.step returns): the fused 64x64 kernels /
// the tiled kernels of AE2D with the error term (and the output layer's gradient) taken against `target`.
// workspace / partial / flags as carle_ae2d_bonus_b1 / carle_ae2d_train_b1.
extern "C" int carle_ae2d_predict_bonus_b1(const uint64_t *obs, const uint64_t *target, int64_t n, int h, int w,
                                           const float *params, float *workspace, float *bonus, float *recon_out,
                                           int flags, void *stream) {
    if (!obs || !target || !params || !workspace || !bonus)
        return fail_arg(CARLE_E_NULL, "carle_ae2d_predict_bonus_b1: NULL pointer");
    if (n < 0 || h != w || h < 64 || (h & 63))
        return fail_arg(CARLE_E_DIMS, "carle_ae2d_predict_bonus_b1: bit-packed frames need H = W a multiple of 64");
    if (recon_out && !aligned16(recon_out)) return fail_arg(CARLE_E_ALIGN, "carle_ae2d_predict_bonus_b1: recon_out alignment");
    if (n == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool valid = (flags & CARLE_NET_TABLES_VALID) != 0;
    if (h == 64) return ae2d_fused64(obs, true, n, params, workspace, bonus, recon_out, nullptr, 0.0f, valid, s, target);
    if (int e = ae2d_tiled(obs, true, n, h, params, workspace, recon_out, valid, s, target)) return e;
    const int tiles = (h >> 6) * (h >> 6);
    hipLaunchKernelGGL(ae_mse_finish_kernel, dim3(blocks_for(n)), dim3(256), 0, s, workspace + 2048 + 128, bonus,
                       static_cast<float *>(nullptr), 0.0f, n, tiles, 1.0f / ((float)h * (float)w));
    return check_launch("carle_ae2d_predict_bonus_b1");
}

extern "C" int carle_ae2d_predict_train_b1(const uint64_t *obs, const uint64_t *target, int64_t n, int h, int w,
                                           const float *params, float *workspace, float *partial, float *grad,
                                           float *bonus, float grad_scale, int flags, void *stream) {
    if (!obs || !target || !params || !workspace || !partial || !grad || !bonus)
        return fail_arg(CARLE_E_NULL, "carle_ae2d_predict_train_b1: NULL pointer");
    if (n < 0 || h != w || h < 64 || (h & 63))
        return fail_arg(CARLE_E_DIMS, "carle_ae2d_predict_train_b1: bit-packed frames need H = W a multiple of 64");
    if (n == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool valid = (flags & CARLE_NET_TABLES_VALID) != 0;
    if (h == 64)
        return ae2d_train64(obs, true, n, params, workspace, bonus, nullptr, 0.0f, grad_scale, partial, grad, valid, s, target);
    if (int e = ae2d_tiled_train(obs, true, n, h, params, workspace, partial, grad_scale, valid, s, target)) return e;
    const int tiles = (h >> 6) * (h >> 6);
    hipLaunchKernelGGL(ae_mse_finish_kernel, dim3(blocks_for(n)), dim3(256), 0, s, workspace + 2048 + 128, bonus,
                       static_cast<float *>(nullptr), 0.0f, n, tiles, 1.0f / ((float)h * (float)w));
    hipLaunchKernelGGL(staged_grad_reduce_kernel, dim3(6), dim3(1024), 0, s, partial, ae2d_tiled_train_rows(n, h), grad, 164);
    return check_launch("carle_ae2d_predict_train_b1");
}

// ---------------------------------------------------------------------------------------------
// RND2D bonus + predictor gradient in one pass (64x64 binary observations), SURVEY 8(f) row 2
// ---------------------------------------------------------------------------------------------
extern "C" int64_t carle_rnd2d_train_partial_floats(int64_t n) {
    return n < 0 ? 0 : (int64_t)rnd2d_train64_rows(n) * 1117;
}

static int rnd2d_train_common(const void *obs, int obs_mode, int64_t n, int h, int w,
                              const float *pred_params, const float *rand_params, float *workspace,
                              float *partial, float *grad, float *bonus, float *reward_accum,
                              float reward_scale, float grad_scale, int flags, void *stream) {
    if (!obs || !pred_params || !rand_params || !workspace || !partial || !grad || !bonus)
        return fail_arg(CARLE_E_NULL, "carle_rnd2d_train: NULL pointer");
    if (n < 0) return fail_arg(CARLE_E_DIMS, "carle_rnd2d_train: negative batch");
    const bool fused = h == 64 && w == 64 && (obs_mode == 1 || ((flags & CARLE_NET_OBS_BINARY) && aligned16(obs)));
    if (!fused && n >= 0 && rnd2d_tiled_applies(h, w) && (obs_mode == 1 || (flags & CARLE_NET_OBS_BINARY))) {
        // binary grids with H = W a multiple of 64: tiled forward (features kept) + tiled conv gradients
        // (nets_tiled.hip) + the staged path's linear-layer gradient and row reduction; `partial` holds
        // carle_rnd2d_train_scratch_floats(n, h, w) floats: pred | targ | features (N,2,D) | rows
        if (n == 0) return 0;
        hipStream_t s = static_cast<hipStream_t>(stream);
        const int d = (h / 8) * (w / 8);
        float *pred = partial, *targ = pred + n * kRndK, *featbuf = targ + n * kRndK, *rows = featbuf + n * 2 * (int64_t)d;
        if (int e = rnd2d_tiled_train(obs, obs_mode == 1, n, h, pred_params, rand_params, workspace, featbuf, pred, targ,
                                      rows, bonus, reward_accum, reward_scale, grad_scale,
                                      (flags & CARLE_NET_TABLES_VALID) != 0, s))
            return e;
        hipLaunchKernelGGL(rnd_bwd_linear_kernel, dim3(blocks256((int64_t)kRndK * d + kRndK)), dim3(256), 0, s, featbuf,
                           pred, targ, grad, grad_scale, n, d);
        hipLaunchKernelGGL(staged_grad_reduce_kernel, dim3(3), dim3(1024), 0, s, rows, rnd2d_tiled_bwd_rows(n, h), grad, 77);
        return check_launch("carle_rnd2d_train(tiled)");
    }
    if (!fused) {
        // any other square size / non-binary fp32 observations: staged forward + staged backward kernels
        // (nets_staged_bwd.hpp); `partial` must hold carle_rnd2d_train_scratch_floats(n, h, w) floats
        if (obs_mode == 1)
            return fail_arg(CARLE_E_DIMS, "carle_rnd2d_train_b1: bit-packed observations need H = W a multiple of 64");
        if (int e = check_net_dims(n, h, w, "carle_rnd2d_train_f32")) return e;
        if (n == 0) return 0;
        return rnd2d_train_staged(static_cast<const float *>(obs), n, h, w, pred_params, rand_params, workspace,
                                  partial, grad, bonus, reward_accum, reward_scale, grad_scale,
                                  static_cast<hipStream_t>(stream));
    }
    if (n == 0) return 0;
    return rnd2d_train64(obs, obs_mode, n, pred_params, rand_params, workspace, bonus, reward_accum,
                         reward_scale, grad_scale, partial, grad, (flags & CARLE_NET_TABLES_VALID) != 0,
                         static_cast<hipStream_t>(stream));
}

extern "C" int carle_rnd2d_train_f32(const float *obs, int64_t n, int h, int w, const float *pred_params,
                                     const float *rand_params, float *workspace, float *partial,
                                     float *grad, float *bonus, float *reward_accum, float reward_scale,
                                     float grad_scale, int flags, void *stream) {
    return rnd2d_train_common(obs, 0, n, h, w, pred_params, rand_params, workspace, partial, grad, bonus,
                              reward_accum, reward_scale, grad_scale, flags, stream);
}

extern "C" int carle_rnd2d_train_b1(const uint64_t *obs, int64_t n, int h, int w, const float *pred_params,
                                    const float *rand_params, float *workspace, float *partial,
                                    float *grad, float *bonus, float *reward_accum, float reward_scale,
                                    float grad_scale, int flags, void *stream) {
    return rnd2d_train_common(obs, 1, n, h, w, pred_params, rand_params, workspace, partial, grad, bonus,
                              reward_accum, reward_scale, grad_scale, flags, stream);
}

// ---------------------------------------------------------------------------------------------
// AE2D bonus + predictor gradient in one pass (64x64 binary observations), SURVEY 8(f) row 2
// ---------------------------------------------------------------------------------------------
extern "C" int64_t carle_ae2d_train_partial_floats(int64_t n) {
    return n < 0 ? 0 : (int64_t)ae2d_train64_rows(n) * 164;
}

static int ae2d_train_common(const void *obs, bool packed, int64_t n, int h, int w, const float *params,
                             float *workspace, float *partial, float *grad, float *bonus,
                             float *reward_accum, float reward_scale, float grad_scale, int flags,
                             void *stream) {
    if (!obs || !params || !workspace || !partial || !grad || !bonus)
        return fail_arg(CARLE_E_NULL, "carle_ae2d_train: NULL pointer");
    if (n < 0) return fail_arg(CARLE_E_DIMS, "carle_ae2d_train: negative batch");
    const bool fused = h == 64 && w == 64 && (packed || ((flags & CARLE_NET_OBS_BINARY) && aligned16(obs)));
    if (!fused && rnd2d_tiled_applies(h, w) && (packed || (flags & CARLE_NET_OBS_BINARY))) {
        // binary grids with H = W a multiple of 64: ONE tiled kernel evaluates the bonus partials and the gradient
        // rows (nets_tiled.hip); finish kernel + row reduction of the staged path.  `partial`: the rows.
        if (n == 0) return 0;
        hipStream_t s = static_cast<hipStream_t>(stream);
        if (int e = ae2d_tiled_train(obs, packed, n, h, params, workspace, partial, grad_scale,
                                     (flags & CARLE_NET_TABLES_VALID) != 0, s))
            return e;
        const int tiles = (h >> 6) * (h >> 6);
        hipLaunchKernelGGL(ae_mse_finish_kernel, dim3(blocks_for(n)), dim3(256), 0, s, workspace + 2048 + 128, bonus,
                           reward_accum, reward_scale, n, tiles, 1.0f / ((float)h * (float)w));
        hipLaunchKernelGGL(staged_grad_reduce_kernel, dim3(6), dim3(1024), 0, s, partial, ae2d_tiled_train_rows(n, h), grad,
                           164);
        return check_launch("carle_ae2d_train(tiled)");
    }
    if (!fused) {
        if (packed)
            return fail_arg(CARLE_E_DIMS, "carle_ae2d_train_b1: bit-packed observations need H = W a multiple of 64");
        if (int e = check_net_dims(n, h, w, "carle_ae2d_train_f32")) return e;
        if (n == 0) return 0;
        return ae2d_train_staged(static_cast<const float *>(obs), static_cast<const float *>(obs), n, h, w, params, workspace,
                                 partial, grad, bonus, reward_accum, reward_scale, grad_scale, static_cast<hipStream_t>(stream));
    }
    if (n == 0) return 0;
    return ae2d_train64(obs, packed, n, params, workspace, bonus, reward_accum, reward_scale, grad_scale,
                        partial, grad, (flags & CARLE_NET_TABLES_VALID) != 0, static_cast<hipStream_t>(stream));
}

extern "C" int carle_ae2d_train_f32(const float *obs, int64_t n, int h, int w, const float *params,
                                    float *workspace, float *partial, float *grad, float *bonus,
                                    float *reward_accum, float reward_scale, float grad_scale, int flags,
                                    void *stream) {
    return ae2d_train_common(obs, false, n, h, w, params, workspace, partial, grad, bonus, reward_accum,
                             reward_scale, grad_scale, flags, stream);
}

extern "C" int carle_ae2d_train_b1(const uint64_t *obs, int64_t n, int h, int w, const float *params,
                                   float *workspace, float *partial, float *grad, float *bonus,
                                   float *reward_accum, float reward_scale, float grad_scale, int flags,
                                   void *stream) {
    return ae2d_train_common(obs, true, n, h, w, params, workspace, partial, grad, bonus, reward_accum,
                             reward_scale, grad_scale, flags, stream);
}

// nets_staged_bwd.hpp -- (included at the end of nets_staged.hip: same translation unit as the staged
// forward kernels it launches)  predictor GRADIENTS of the RND2D / AE2D bonuses as staged HIP kernels, any
// H == W that is a multiple of 8 (the reference's default 256x256 geometry included).
//
// What it replaces: the autograd graph the reference keeps alive over its update window and the
// `loss.backward()` at its end (carle/mcl.py:343-386 RND2D, 504-547 AE2D).  As in the fused 64x64
// kernels (nets_fused.hip) the gradient of  grad_scale/2 * sum_i sum (target - prediction)^2  is
// ACCUMULATED into a flat vector in predictor-blob order in the same call that evaluates the bonus,
// so no observation window is retained and PyTorch autograd is not involved; torch.optim.Adam
// consumes the vector at the end of the window.  Eval-mode semantics (Dropout = identity).
//
// The forward pass is the staged one (nets_staged.hip); its activations stay in the workspace.  The
// backward pass walks the layers in reverse, one kernel per layer, thread = one activation:
//   * max-pool + relu: the gradient of a pooled value goes to the FIRST maximum of its window in
//     scan order (what ATen's max_pool2d backward does; two stacked 2x2 pools = first maximum of the
//     first maxima), and only if the pooled value is positive;
//   * a convolution's input gradient is GATHERED (each input position sums the <= 9 output
//     positions it feeds; the sparse pooled gradients are stored as (value, arg-max index) per
//     window), transposed convolutions likewise: no atomics anywhere;
//   * weight gradients: every thread forms its contributions, the workgroup adds them up by
//     shuffles + a fixed-order LDS combine into ONE partial row, and rnd_grad_reduce_kernel adds
//     the rows in a fixed order: deterministic.
#pragma once
#include "train_bodies.hpp"

namespace carle {

// adds `rows` partial rows of `nparams` floats in a fixed order onto grad[0..nparams): workgroup b
// takes parameters 32 b .. 32 b + 31, thread = (parameter lane, one of 32 row lanes); deterministic
__global__ __launch_bounds__(1024) void staged_grad_reduce_kernel(const float *__restrict__ partial, int rows,
                                                                 float *__restrict__ grad, int nparams) {
    __shared__ float part[32][33];
    staged_grad_reduce_body(part, partial, rows, grad, nparams, (int)blockIdx.x);
}

static inline unsigned blocks256(int64_t total) { return (unsigned)((total + 255) / 256); }
// launch size of the backward kernels: at most 256 workgroups (= partial rows, one per CU); they walk their
// elements with a grid stride and add up in registers first
static inline unsigned bwd_rows(int64_t total) { const int64_t b = (total + 255) / 256; return (unsigned)(b < 256 ? b : 256); }

// Sums of NV per-thread values over the 256 threads of the workgroup -> row[0..NV).  Every thread of
// the workgroup must call it (threads without work contribute zeros).
template <int NV>
__device__ __forceinline__ void block_row_sums(const float (&v)[NV], float *__restrict__ row) {
    __shared__ float s_part[4][NV];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float x = v[i];
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1) x += __shfl_xor(x, sh, 64);
        if (lane == 0) s_part[wv][i] = x;
    }
    __syncthreads();
    if (threadIdx.x < NV)
        row[threadIdx.x] = (s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) +
                           (s_part[2][threadIdx.x] + s_part[3][threadIdx.x]);
}

// first maximum (strict >) of four values in scan order: index 0..3
__device__ __forceinline__ int first_max4(float a, float b, float c, float d) {
    int s = 0;
    float m = a;
    if (b > m) { m = b; s = 1; }
    if (c > m) { m = c; s = 2; }
    if (d > m) { s = 3; }
    return s;
}

// =============================================================================================
// RND2D predictor:  conv1(1->4) relu pool2 pool2 conv2(4->1) relu pool2 flatten linear(D->16) tanh
// blob: conv1.w[36] conv1.b[4] conv2.w[36] conv2.b[1] lin.w[16][D] lin.b[16]
// =============================================================================================
__global__ __launch_bounds__(256) void rnd_bwd_linear_kernel(
    const float *__restrict__ ws2, const float *__restrict__ pred, const float *__restrict__ targ,
    float *__restrict__ grad, float gs, int64_t n, int d) {
    rnd_bwd_linear_body(ws2, pred, targ, grad, gs, n, d, (int64_t)blockIdx.x);
}

// pooled conv2 output f of grid n: gradient through the linear layer, relu gate, arg-max of the 2x2
// window; stores (d, sel) for the gather of the next kernel; conv2 weight / bias gradient rows
__global__ __launch_bounds__(256) void rnd_bwd_conv2_kernel(
    const float *__restrict__ ws1, const float *__restrict__ ws2, const float *__restrict__ pp,
    const float *__restrict__ pred, const float *__restrict__ targ, float *__restrict__ dsel_d,
    int *__restrict__ dsel_s, float *__restrict__ partial, float gs, int64_t n, int h, int w) {
    const int h4 = h >> 2, w4 = w >> 2, w8 = w >> 3, d = (h >> 3) * w8;
    float v[37];
#pragma unroll
    for (int i = 0; i < 37; ++i) v[i] = 0.0f;
    // grid-stride walk (the launch is capped: few partial rows for the fixed-order reduction)
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n * (int64_t)d;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int f = (int)(idx % d);
        const int64_t g = idx / d;
        const int fx = f % w8, fy = f / w8;
        float dfeat = 0.0f;
        const float *lw = pp + 77;
#pragma unroll 4
        for (int k = 0; k < kRndK; ++k) dfeat = fmaf(lw[(int64_t)k * d + f], rnd_dz(pred, targ, g, k, gs), dfeat);
        const float feat = ws2[(g * 2) * (int64_t)d + f];
        const float dd = feat > 0.0f ? dfeat : 0.0f;
        // the four conv2 outputs of the window (as rnd_conv2_pool2_kernel computes them)
        const float *wt = pp + 40;
        float patch[4][4][4];
        float o[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float *pl = ws1 + (g * 6 + c) * (int64_t)h4 * w4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int y = 2 * fy - 1 + r;
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const int x = 2 * fx - 1 + cc;
                    patch[c][r][cc] = ldz(pl, y * w4 + x, y >= 0 && y < h4 && x >= 0 && x < w4);
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
                            o[oy][ox] = fmaf(wt[c * 9 + ky * 3 + kx], patch[c][oy + ky][ox + kx], o[oy][ox]);
        }
        const int sel = first_max4(o[0][0], o[0][1], o[1][0], o[1][1]);
        dsel_d[idx] = dd;
        dsel_s[idx] = sel;
        const int ay = sel >> 1, ax = sel & 1;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float p00 = patch[c][ky][kx], p01 = patch[c][ky][kx + 1];
                    const float p10 = patch[c][ky + 1][kx], p11 = patch[c][ky + 1][kx + 1];
                    v[c * 9 + ky * 3 + kx] += dd * (ay ? (ax ? p11 : p10) : (ax ? p01 : p00));
                }
        v[36] += dd;
    }
    block_row_sums<37>(v, partial + (int64_t)blockIdx.x * 37);
}

// pooled conv1 position (u, v) of grid n, the four predictor channels: gradient gathered from the
// sparse conv2-output gradients, relu gate, arg-max of the 4x4 window (two stacked 2x2 pools), conv1
// weight / bias gradient rows
__global__ __launch_bounds__(256) void rnd_bwd_conv1_kernel(
    const float *__restrict__ obs, const float *__restrict__ ws1, const float *__restrict__ pp,
    const float *__restrict__ dsel_d, const int *__restrict__ dsel_s, float *__restrict__ partial,
    int64_t n, int h, int w) {
    const int h4 = h >> 2, w4 = w >> 2, w8 = w >> 3, d = (h >> 3) * w8;
    float v[40];
#pragma unroll
    for (int i = 0; i < 40; ++i) v[i] = 0.0f;
    // grid-stride walk (the launch is capped: few partial rows for the fixed-order reduction)
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n * (int64_t)h4 * w4;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int px = (int)(idx % w4), py = (int)((idx / w4) % h4);
        const int64_t g = idx / ((int64_t)h4 * w4);
        // d loss / d conv2 output at the nine positions this pooled pixel feeds
        float dm[3][3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int y = py - ky + 1, x = px - kx + 1;
                const bool in = y >= 0 && y < h4 && x >= 0 && x < w4;
                const int64_t fi = in ? g * (int64_t)d + (y >> 1) * w8 + (x >> 1) : 0;      // unconditional loads
                const int sel = dsel_s[fi];
                const float dv = dsel_d[fi];
                dm[ky][kx] = (in && sel == (((y & 1) << 1) | (x & 1))) ? dv : 0.0f;
            }
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
        for (int ch = 0; ch < 4; ++ch) {
            float dp = 0.0f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) dp = fmaf(pp[40 + ch * 9 + ky * 3 + kx], dm[ky][kx], dp);
            const float pooled = ws1[((g * 6 + ch) * h4 + py) * (int64_t)w4 + px];
            const float g1 = pooled > 0.0f ? dp : 0.0f;
            // conv1 outputs of the 4x4 window; first maximum of the first maxima of its 2x2 blocks
            float k9[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) k9[t] = pp[ch * 9 + t];
            float bm[2][2];
            int bs[2][2];
#pragma unroll
            for (int by = 0; by < 2; ++by)
#pragma unroll
                for (int bx = 0; bx < 2; ++bx) {
                    float c4[2][2];
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 2; ++dx) {
                            float acc = 0.0f;
#pragma unroll
                            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                                for (int kx = 0; kx < 3; ++kx)
                                    acc = fmaf(k9[ky * 3 + kx], patch[2 * by + dy + ky][2 * bx + dx + kx], acc);
                            c4[dy][dx] = acc;
                        }
                    const int s4 = first_max4(c4[0][0], c4[0][1], c4[1][0], c4[1][1]);
                    bs[by][bx] = s4;
                    bm[by][bx] = s4 == 0 ? c4[0][0] : s4 == 1 ? c4[0][1] : s4 == 2 ? c4[1][0] : c4[1][1];
                }
            const int sb = first_max4(bm[0][0], bm[0][1], bm[1][0], bm[1][1]);
            const int s4 = sb == 0 ? bs[0][0] : sb == 1 ? bs[0][1] : sb == 2 ? bs[1][0] : bs[1][1];
            const int y0 = 4 * py + 2 * (sb >> 1) + (s4 >> 1) - 1, x0 = 4 * px + 2 * (sb & 1) + (s4 & 1) - 1;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int y = y0 + ky, x = x0 + kx;
                    const float in = ldz(go, y * w + x, y >= 0 && y < h && x >= 0 && x < w);
                    v[ch * 9 + ky * 3 + kx] += g1 * in;
                }
            v[36 + ch] += g1;
        }
    }
    block_row_sums<40>(v, partial + (int64_t)blockIdx.x * 40);
}

int64_t rnd2d_train_staged_scratch_floats(int64_t n, int h, int w) {
    const int64_t d = (int64_t)(h / 8) * (w / 8);
    return 2 * n * kRndK + 2 * n * d + 1024 * (37 + 40) + 64;
}

// bonus (staged forward) + accumulated predictor gradient; validated by the caller
int rnd2d_train_staged(const float *obs, int64_t n, int h, int w, const float *pp, const float *rp, float *ws,
                       float *scratch, float *grad, float *bonus, float *reward, float reward_scale,
                       float grad_scale, hipStream_t s) {
    const int64_t p1 = n * (int64_t)(h / 4) * (w / 4);
    const int d = (h / 8) * (w / 8);
    if (p1 * 6 / 256 > 0x7FFFFFFFll) return fail_arg(CARLE_E_DIMS, "carle_rnd2d_train: batch too large");
    float *ws1 = ws, *ws2 = ws + 6 * p1;
    float *pred = scratch, *targ = pred + n * kRndK, *dsel_d = targ + n * kRndK;
    int *dsel_s = reinterpret_cast<int *>(dsel_d + n * (int64_t)d);
    float *part2 = reinterpret_cast<float *>(dsel_s + n * (int64_t)d);
    const unsigned rows2 = bwd_rows(n * (int64_t)d), rows1 = bwd_rows(p1);
    float *part1 = part2 + (int64_t)rows2 * 37;
    hipLaunchKernelGGL(rnd_conv1_pool4_kernel, dim3(blocks256(p1)), dim3(256), 0, s, obs, pp, rp, ws1, n, h, w);
    hipLaunchKernelGGL(rnd_conv2_pool2_kernel, dim3(blocks256(n * 2 * (int64_t)d)), dim3(256), 0, s, ws1, pp, rp, ws2,
                       n, h, w);
    launch_rnd_linear(ws2, pp, rp, bonus, pred, targ, reward, reward_scale, n, d, s);
    hipLaunchKernelGGL(rnd_bwd_linear_kernel, dim3(blocks256((int64_t)kRndK * d + kRndK)), dim3(256), 0, s, ws2, pred,
                       targ, grad, grad_scale, n, d);
    hipLaunchKernelGGL(rnd_bwd_conv2_kernel, dim3(rows2), dim3(256), 0, s, ws1, ws2, pp, pred, targ, dsel_d, dsel_s,
                       part2, grad_scale, n, h, w);
    hipLaunchKernelGGL(rnd_bwd_conv1_kernel, dim3(rows1), dim3(256), 0, s, obs, ws1, pp, dsel_d, dsel_s, part1, n, h, w);
    hipLaunchKernelGGL(staged_grad_reduce_kernel, dim3(2), dim3(1024), 0, s, part2, (int)rows2, grad + 40, 37);
    hipLaunchKernelGGL(staged_grad_reduce_kernel, dim3(2), dim3(1024), 0, s, part1, (int)rows1, grad, 40);
    return check_launch("carle_rnd2d_train(staged)");
}

// =============================================================================================
// AE2D predictor:  conv1(1->4) relu pool2  conv2(4->2) relu pool2  convT1(2->1,k4,s2,p1) relu
//                  convT2(1->1,k4,s2,p1) sigmoid;  loss = mean_hw (obs - R)^2
// blob: conv1.w[36] b[4] | conv2.w[2][4][9] b[2] | convT1.w[2][16] b[1] | convT2.w[16] b[1]
// =============================================================================================

// transposed conv k4 s2 p1: output row y <- input row iy with tap ky = y + 1 - 2 iy in [0, 3]
__device__ __forceinline__ bool ct_tap(int y, int ky, int lim, int &iy) {
    const int t = y + 1 - ky;
    iy = t >> 1;
    return (t & 1) == 0 && t >= 0 && iy < lim;
}

// output pixel: reconstruction, d loss / d pre-sigmoid; convT2 weight / bias gradient rows
__global__ __launch_bounds__(256) void ae_bwd_out_kernel(
    const float *__restrict__ obs, const float *__restrict__ a3, const float *__restrict__ ap,
    float *__restrict__ dpre4, float *__restrict__ partial, float gs, int64_t n, int h, int w) {
    const int h2 = h >> 1, w2 = w >> 1;
    float v[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) v[i] = 0.0f;
    // grid-stride walk (the launch is capped: few partial rows for the fixed-order reduction)
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n * (int64_t)h * w;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(idx % w), y = (int)((idx / w) % h);
        const int64_t g = idx / ((int64_t)h * w);
        const float *pl = a3 + g * (int64_t)h2 * w2;
        float in[4][4];
        float acc = 0.0f;
#pragma unroll
        for (int ky = 0; ky < 4; ++ky)
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) {
                int iy, ix;
                const bool ok = ((int)ct_tap(y, ky, h2, iy) & (int)ct_tap(x, kx, w2, ix)) != 0;   // both evaluated: out-params
                in[ky][kx] = ldz(pl, iy * w2 + ix, ok);
                acc = fmaf(in[ky][kx], ap[147 + ky * 4 + kx], acc);
            }
        const float r = 1.0f / (1.0f + expf(-(acc + ap[163])));
        const float dd = gs * (r - obs[idx]) * r * (1.0f - r);
        dpre4[idx] = dd;
#pragma unroll
        for (int ky = 0; ky < 4; ++ky)
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) v[ky * 4 + kx] += dd * in[ky][kx];
        v[16] += dd;
    }
    block_row_sums<17>(v, partial + (int64_t)blockIdx.x * 17);
}

// A3 pixel: gradient gathered from the output pixels it feeds, relu gate; convT1 weight / bias rows
__global__ __launch_bounds__(256) void ae_bwd_a3_kernel(
    const float *__restrict__ a2, const float *__restrict__ a3, const float *__restrict__ ap,
    const float *__restrict__ dpre4, float *__restrict__ dpre3, float *__restrict__ partial, int64_t n,
    int h, int w) {
    const int h2 = h >> 1, w2 = w >> 1, h4 = h >> 2, w4 = w >> 2;
    float v[33];
#pragma unroll
    for (int i = 0; i < 33; ++i) v[i] = 0.0f;
    // grid-stride walk (the launch is capped: few partial rows for the fixed-order reduction)
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n * (int64_t)h2 * w2;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(idx % w2), y = (int)((idx / w2) % h2);
        const int64_t g = idx / ((int64_t)h2 * w2);
        const float *dp = dpre4 + g * (int64_t)h * w;
        float da = 0.0f;
#pragma unroll
        for (int ky = 0; ky < 4; ++ky)
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) {
                const int oy = 2 * y + ky - 1, ox = 2 * x + kx - 1;      // output pixel fed through tap (ky, kx)
                da = fmaf(ldz(dp, oy * w + ox, oy >= 0 && oy < h && ox >= 0 && ox < w), ap[147 + ky * 4 + kx], da);
            }
        const float d3 = a3[idx] > 0.0f ? da : 0.0f;
        dpre3[idx] = d3;
#pragma unroll
        for (int ky = 0; ky < 4; ++ky)
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) {
                int iy, ix;
                const bool ok = ((int)ct_tap(y, ky, h4, iy) & (int)ct_tap(x, kx, w4, ix)) != 0;   // both evaluated: out-params
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    v[c * 16 + ky * 4 + kx] += d3 * ldz(a2, ((g * 2 + c) * h4 + iy) * (int64_t)w4 + ix, ok);
            }
        v[32] += d3;
    }
    block_row_sums<33>(v, partial + (int64_t)blockIdx.x * 33);
}

// A2 pixel, both channels: gradient gathered from A3, relu gate, arg-max of the 2x2 window of conv2
// outputs; stores (d, sel); conv2 weight / bias gradient rows
__global__ __launch_bounds__(256) void ae_bwd_a2_kernel(
    const float *__restrict__ a1, const float *__restrict__ a2, const float *__restrict__ ap,
    const float *__restrict__ dpre3, float *__restrict__ dsel_d, int *__restrict__ dsel_s,
    float *__restrict__ partial, int64_t n, int h, int w) {
    const int h2 = h >> 1, w2 = w >> 1, h4 = h >> 2, w4 = w >> 2;
    float v[74];
#pragma unroll
    for (int i = 0; i < 74; ++i) v[i] = 0.0f;
    // grid-stride walk (the launch is capped: few partial rows for the fixed-order reduction)
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n * (int64_t)h4 * w4;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int px = (int)(idx % w4), py = (int)((idx / w4) % h4);
        const int64_t g = idx / ((int64_t)h4 * w4);
        const float *d3 = dpre3 + g * (int64_t)h2 * w2;
        float patch[4][4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float *pl = a1 + (g * 4 + c) * (int64_t)h2 * w2;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int y = 2 * py - 1 + r;
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const int x = 2 * px - 1 + cc;
                    patch[c][r][cc] = ldz(pl, y * w2 + x, y >= 0 && y < h2 && x >= 0 && x < w2);
                }
            }
        }
#pragma unroll
        for (int oc = 0; oc < 2; ++oc) {
            float da = 0.0f;
#pragma unroll
            for (int ky = 0; ky < 4; ++ky)
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) {
                    const int oy = 2 * py + ky - 1, ox = 2 * px + kx - 1;
                    da = fmaf(ldz(d3, oy * w2 + ox, oy >= 0 && oy < h2 && ox >= 0 && ox < w2),
                              ap[114 + oc * 16 + ky * 4 + kx], da);
                }
            const int64_t ai = ((g * 2 + oc) * h4 + py) * (int64_t)w4 + px;
            const float dd = a2[ai] > 0.0f ? da : 0.0f;
            float o[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int oy = 0; oy < 2; ++oy)
#pragma unroll
                    for (int ox = 0; ox < 2; ++ox)
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx)
                                o[oy][ox] = fmaf(ap[40 + (oc * 4 + c) * 9 + ky * 3 + kx], patch[c][oy + ky][ox + kx],
                                                 o[oy][ox]);
            const int sel = first_max4(o[0][0], o[0][1], o[1][0], o[1][1]);
            dsel_d[ai] = dd;
            dsel_s[ai] = sel;
            const int ay = sel >> 1, ax = sel & 1;
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float p00 = patch[c][ky][kx], p01 = patch[c][ky][kx + 1];
                        const float p10 = patch[c][ky + 1][kx], p11 = patch[c][ky + 1][kx + 1];
                        v[(oc * 4 + c) * 9 + ky * 3 + kx] += dd * (ay ? (ax ? p11 : p10) : (ax ? p01 : p00));
                    }
            v[72 + oc] += dd;
        }
    }
    block_row_sums<74>(v, partial + (int64_t)blockIdx.x * 74);
}

// A1 pixel, four channels: gradient gathered from the sparse conv2-output gradients of both output
// channels, relu gate, arg-max of the 2x2 window of conv1 outputs; conv1 weight / bias rows
__global__ __launch_bounds__(256) void ae_bwd_a1_kernel(
    const float *__restrict__ obs, const float *__restrict__ a1, const float *__restrict__ ap,
    const float *__restrict__ dsel_d, const int *__restrict__ dsel_s, float *__restrict__ partial,
    int64_t n, int h, int w) {
    const int h2 = h >> 1, w2 = w >> 1, h4 = h >> 2, w4 = w >> 2;
    float v[40];
#pragma unroll
    for (int i = 0; i < 40; ++i) v[i] = 0.0f;
    // grid-stride walk (the launch is capped: few partial rows for the fixed-order reduction)
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n * (int64_t)h2 * w2;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int px = (int)(idx % w2), py = (int)((idx / w2) % h2);
        const int64_t g = idx / ((int64_t)h2 * w2);
        float dm[2][3][3];
#pragma unroll
        for (int oc = 0; oc < 2; ++oc)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int y = py - ky + 1, x = px - kx + 1;
                    const bool in = y >= 0 && y < h2 && x >= 0 && x < w2;
                    const int64_t ai = in ? ((g * 2 + oc) * h4 + (y >> 1)) * (int64_t)w4 + (x >> 1) : 0;   // unconditional loads
                    const int sel = dsel_s[ai];
                    const float dv = dsel_d[ai];
                    dm[oc][ky][kx] = (in && sel == (((y & 1) << 1) | (x & 1))) ? dv : 0.0f;
                }
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
            float da = 0.0f;
#pragma unroll
            for (int oc = 0; oc < 2; ++oc)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
                        da = fmaf(ap[40 + (oc * 4 + ch) * 9 + ky * 3 + kx], dm[oc][ky][kx], da);
            const float g1 = a1[((g * 4 + ch) * h2 + py) * (int64_t)w2 + px] > 0.0f ? da : 0.0f;
            float c4[2][2];
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
                    c4[dy][dx] = acc;
                }
            const int sel = first_max4(c4[0][0], c4[0][1], c4[1][0], c4[1][1]);
            const int ay = sel >> 1, ax = sel & 1;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float p00 = patch[ky][kx], p01 = patch[ky][kx + 1];
                    const float p10 = patch[ky + 1][kx], p11 = patch[ky + 1][kx + 1];
                    v[ch * 9 + ky * 3 + kx] += g1 * (ay ? (ax ? p11 : p10) : (ax ? p01 : p00));
                }
            v[36 + ch] += g1;
        }
    }
    block_row_sums<40>(v, partial + (int64_t)blockIdx.x * 40);
}


int64_t ae2d_train_staged_scratch_floats(int64_t n, int h, int w) {
    const int64_t q = (int64_t)(h / 2) * (w / 2), px = n * (int64_t)h * w;
    return px + n * q + 2 * (n * 2 * (q / 4)) + 1024 * (17 + 33 + 40 + 74) + 64;
}

// `target`: the frame the reconstruction is compared with -- `obs` itself for AE2D (mcl.py:504-512), a LATER frame for
// PredictionBonus (mcl.py:581-605: decoder(frame t-k) against frame t); conv1's weight gradient reads `obs`, the error
// term and the output layer's gradient read `target`
int ae2d_train_staged(const float *obs, const float *target, int64_t n, int h, int w, const float *ap, float *ws,
                      float *scratch, float *grad, float *bonus, float *reward, float reward_scale, float grad_scale,
                      hipStream_t s) {
    const int64_t q = (int64_t)(h / 2) * (w / 2), px = n * (int64_t)h * w;
    if (px / 256 > 0x7FFFFFFFll) return fail_arg(CARLE_E_DIMS, "carle_ae2d_train: batch too large");
    float *a1 = ws, *a2 = a1 + n * 4 * q, *a3 = a2 + n * 2 * (q / 4);
    const int chunks = (h * w + kMseChunk - 1) / kMseChunk;
    float *mse_part = a3 + n * q;
    float *dpre4 = scratch, *dpre3 = dpre4 + px, *dsel_d = dpre3 + n * q;
    int *dsel_s = reinterpret_cast<int *>(dsel_d + n * 2 * (q / 4));
    float *part4 = reinterpret_cast<float *>(dsel_s + n * 2 * (q / 4));
    const unsigned rows4 = bwd_rows(px), rows3 = bwd_rows(n * q), rows2 = bwd_rows(n * (q / 4)), rows1 = rows3;
    float *part3 = part4 + (int64_t)rows4 * 17, *part2 = part3 + (int64_t)rows3 * 33;
    float *part1 = part2 + (int64_t)rows2 * 74;
    // forward (nets_staged.hip), activations stay in the workspace
    hipLaunchKernelGGL(ae_conv1_pool2_kernel, dim3(blocks256(n * q)), dim3(256), 0, s, obs, ap, a1, n, h, w);
    hipLaunchKernelGGL(ae_conv2_pool2_kernel, dim3(blocks256(n * (q / 4))), dim3(256), 0, s, a1, ap, a2, n, h, w);
    hipLaunchKernelGGL(ae_convt1_kernel, dim3(blocks256(n * q)), dim3(256), 0, s, a2, ap, a3, n, h, w);
    hipLaunchKernelGGL(ae_convt2_mse_kernel, dim3((unsigned)(n * chunks)), dim3(256), 0, s, target, a3, ap, mse_part,
                       static_cast<float *>(nullptr), chunks, h, w);
    hipLaunchKernelGGL(ae_mse_finish_kernel, dim3(blocks256(n)), dim3(256), 0, s, mse_part, bonus, reward,
                       reward_scale, n, chunks, 1.0f / (float)(h * w));
    // backward
    hipLaunchKernelGGL(ae_bwd_out_kernel, dim3(rows4), dim3(256), 0, s, target, a3, ap, dpre4, part4, grad_scale, n, h, w);
    hipLaunchKernelGGL(ae_bwd_a3_kernel, dim3(rows3), dim3(256), 0, s, a2, a3, ap, dpre4, dpre3, part3, n, h, w);
    hipLaunchKernelGGL(ae_bwd_a2_kernel, dim3(rows2), dim3(256), 0, s, a1, a2, ap, dpre3, dsel_d, dsel_s, part2, n, h, w);
    hipLaunchKernelGGL(ae_bwd_a1_kernel, dim3(rows1), dim3(256), 0, s, obs, a1, ap, dsel_d, dsel_s, part1, n, h, w);
    hipLaunchKernelGGL(staged_grad_reduce_kernel, dim3(1), dim3(1024), 0, s, part4, (int)rows4, grad + 147, 17);
    hipLaunchKernelGGL(staged_grad_reduce_kernel, dim3(2), dim3(1024), 0, s, part3, (int)rows3, grad + 114, 33);
    hipLaunchKernelGGL(staged_grad_reduce_kernel, dim3(3), dim3(1024), 0, s, part2, (int)rows2, grad + 40, 74);
    hipLaunchKernelGGL(staged_grad_reduce_kernel, dim3(2), dim3(1024), 0, s, part1, (int)rows1, grad, 40);
    return check_launch("carle_ae2d_train(staged)");
}

}  // namespace carle

namespace carle {
bool rnd2d_tiled_applies(int h, int w);                                   // nets_tiled.hip
int64_t rnd2d_tiled_bits_offset_floats(int64_t n, int size);
}  // namespace carle

using namespace carle;

extern "C" int64_t carle_rnd2d_train_scratch_floats(int64_t n, int h, int w) {
    if (n < 0 || h < 8 || w < 8 || h != w || (h % 8) != 0) return 0;
    const int64_t staged = rnd2d_train_staged_scratch_floats(n, h, w);
    // tiled sizes: + a bit-packed copy of fp32 observations behind it (carle_rnd_ae_train: the tile gradients and the AE2D
    // tiles read the copy the forward tiles leave)
    if (rnd2d_tiled_applies(h, w)) return rnd2d_tiled_bits_offset_floats(n, h) + n * (int64_t)h * (w / 64) * 2;
    return staged;
}

extern "C" int64_t carle_ae2d_train_scratch_floats(int64_t n, int h, int w) {
    if (n < 0 || h < 8 || w < 8 || h != w || (h % 8) != 0) return 0;
    return ae2d_train_staged_scratch_floats(n, h, w);
}

// agent_net.hip -- RandomNetworkAgent forward on the device (carle/agents.py:45-102):
//   Conv2d(1,4,3,p1,no bias) - ReLU - MaxPool(2) - Conv2d(4,1,3,p1,no bias) - ReLU - MaxPool(2) - Flatten
//   - Linear(dense, AW*AH, no bias) - Sigmoid;   action = 1.0 * (output <= toggle_rate)
// Two kernels:
//   rn_features_kernel: one workgroup per 64x64-cell tile of a grid (16x16 features): the tile + a
//     3-cell halo goes to LDS, conv1 + relu + pool -> four 34x34 planes in LDS (zero outside the grid:
//     conv2's padding), conv2 + relu + pool -> features.  Direct fp32 convolutions: any observation
//     values (the reference feeds the observation tensor itself), H and W multiples of 8.
//   rn_linear_kernel: out = feat (N x K) * W^T (K x OUT) on the fp32 matrix cores
//     (v_mfma_f32_32x32x2_f32: this IS a dense contraction), 64x64 output tile per workgroup (2x2
//     waves), K in chunks of 32 through LDS (row stride 33: conflict-free operand reads), sigmoid and
//     the threshold in the epilogue: the (N, OUT) action is written once, the logits never leave the CU.
//     Small batches (the reference's default: a handful of 256x256 instances against a 64 MiB weight
//     matrix) have too few output tiles to pull the weights at HBM rate: K is then split over
//     blockIdx.z, the partial logits go to the workspace and rn_finish_kernel adds them in a fixed
//     order (deterministic) before the sigmoid.
#include "common.hpp"
#include <math.h>

namespace carle {

static constexpr int kTile = 64, kHalo = 3, kCellsDim = kTile + 2 * kHalo;     // 70
static constexpr int kP1Dim = 34;                                              // pooled conv1: 32 + 1 halo each side

__global__ __launch_bounds__(256) void rn_features_kernel(const float *__restrict__ obs, int h, int w,
                                                          const float *__restrict__ w1,
                                                          const float *__restrict__ w2,
                                                          float *__restrict__ feat) {
    __shared__ float s_cells[kCellsDim][kCellsDim + 1];
    __shared__ float s_p1[4][kP1Dim][kP1Dim + 1];
    const int tid = threadIdx.x;
    const int tiles_x = (w + kTile - 1) / kTile;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x % tiles_x;
    const int64_t g = blockIdx.y;
    const float *gin = obs + g * (int64_t)h * w;
    const int y0 = ty * kTile - kHalo, x0 = tx * kTile - kHalo;
    for (int i = tid; i < kCellsDim * kCellsDim; i += 256) {
        const int r = i / kCellsDim, c = i % kCellsDim;
        const int y = y0 + r, x = x0 + c;
        s_cells[r][c] = (y >= 0 && y < h && x >= 0 && x < w) ? gin[(int64_t)y * w + x] : 0.0f;
    }
    float k1[4][9], k2[4][9];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) { k1[c][t] = w1[c * 9 + t]; k2[c][t] = w2[c * 9 + t]; }
    __syncthreads();

    // pooled conv1 positions [32ty-1, 32ty+33) x [32tx-1, 32tx+33); position p covers cells 2p, 2p+1
    const int h2 = h >> 1, w2d = w >> 1;
    for (int i = tid; i < kP1Dim * kP1Dim; i += 256) {
        const int pr = i / kP1Dim, pc = i % kP1Dim;
        const int py = 32 * ty - 1 + pr, px = 32 * tx - 1 + pc;
        float m[4] = {0.f, 0.f, 0.f, 0.f};
        if (py >= 0 && py < h2 && px >= 0 && px < w2d) {
            // cells 2py-1 .. 2py+2 -> local rows 2py-1-y0 = 2pr .. 2pr+3 (y0 = 64ty-3, py = 32ty-1+pr)
            float patch[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) patch[r][c] = s_cells[2 * pr + r][2 * pc + c];
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                float best = 0.0f;                    // relu then max == max(0, max conv)
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        float acc = 0.0f;
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx) acc = fmaf(k1[ch][ky * 3 + kx], patch[dy + ky][dx + kx], acc);
                        best = fmaxf(best, acc);
                    }
                m[ch] = best;
            }
        }
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) s_p1[ch][pr][pc] = m[ch];
    }
    __syncthreads();

    // features: thread (fy, fx) of the tile's 16x16; conv2 positions 2f, 2f+1 of the pooled plane
    const int fyl = tid >> 4, fxl = tid & 15;
    const int fy = 16 * ty + fyl, fx = 16 * tx + fxl;
    if (fy < (h >> 2) && fx < (w >> 2)) {
        float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            float patch[4][4];                        // pooled rows 2fy-1 .. 2fy+2 -> local 2fyl .. 2fyl+3
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) patch[r][c] = s_p1[ch][2 * fyl + r][2 * fxl + c];
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx)
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx)
                            acc[dy][dx] = fmaf(k2[ch][ky * 3 + kx], patch[dy + ky][dx + kx], acc[dy][dx]);
        }
        const float mx = fmaxf(fmaxf(acc[0][0], acc[0][1]), fmaxf(acc[1][0], acc[1][1]));
        feat[g * (int64_t)(h >> 2) * (w >> 2) + (int64_t)fy * (w >> 2) + fx] = fmaxf(mx, 0.0f);
    }
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
static constexpr int kBM = 64, kBN = 64, kBK = 32, kLd = kBK + 1;

// C (M x OUT) = A (M x K, row-major) * B^T (B is OUT x K, row-major).  K % 4 == 0.  SPLIT: blockIdx.z
// takes the K chunks [z * chunks_per_split, ...) and writes raw partial sums to partial[z][M][OUT].
template <bool SPLIT>
__global__ __launch_bounds__(256) void rn_linear_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                        int64_t m, int k, int out_nodes, float toggle_rate,
                                                        float logit_rate, int chunks_per_split,
                                                        float *__restrict__ partial, float *__restrict__ output,
                                                        float *__restrict__ action) {
    __shared__ float s_a[kBM][kLd];
    __shared__ float s_b[kBN][kLd];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;                  // the wave's 32x32 quadrant of the tile
    const int64_t row0 = (int64_t)blockIdx.y * kBM;
    const int col0 = blockIdx.x * kBN;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;

    // loader mapping: 8 threads x float4 cover a 32-float row segment; 32 rows per pass, two passes
    const int lr = tid >> 3, lc = (tid & 7) * 4;
    const int kbeg = SPLIT ? (int)blockIdx.z * chunks_per_split * kBK : 0;
    const int kend = SPLIT ? min(k, kbeg + chunks_per_split * kBK) : k;
    for (int k0 = kbeg; k0 < kend; k0 += kBK) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r = lr + 32 * p;
            float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
            if (row0 + r < m && k0 + lc < k) va = *reinterpret_cast<const float4 *>(a + (row0 + r) * k + k0 + lc);
            if (col0 + r < out_nodes && k0 + lc < k)
                vb = *reinterpret_cast<const float4 *>(b + (int64_t)(col0 + r) * k + k0 + lc);
            s_a[r][lc] = va.x; s_a[r][lc + 1] = va.y; s_a[r][lc + 2] = va.z; s_a[r][lc + 3] = va.w;
            s_b[r][lc] = vb.x; s_b[r][lc + 1] = vb.y; s_b[r][lc + 2] = vb.z; s_b[r][lc + 3] = vb.w;
        }
        __syncthreads();
        // v_mfma_f32_32x32x2_f32: lane l supplies A[i = l % 32][kk = l / 32] and B[kk = l / 32][j = l % 32]
        const int i32 = lane & 31, kh = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < kBK; kk += 2) {
            const float av = s_a[32 * wm + i32][kk + kh];
            const float bv = s_b[32 * wn + i32][kk + kh];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    // D layout: register v of lane l holds row 8 * (v / 4) + 4 * (l / 32) + v % 4, column l % 32
    const int j = col0 + 32 * wn + (lane & 31);
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int64_t i = row0 + 32 * wm + 8 * (v >> 2) + 4 * (lane >> 5) + (v & 3);
        if (i < m && j < out_nodes) {
            if (SPLIT) {
                partial[((int64_t)blockIdx.z * m + i) * out_nodes + j] = acc[v];
            } else {
                // sigmoid is monotonic: sigmoid(x) <= rate  <=>  x <= logit(rate); saves ~30 instructions per output.
                // ONE threshold form in every path (with or without the sigmoid values, split-K or not): the same
                // observation gives the same action whatever the caller asked for (the reference's own form,
                // fp32 sigmoid <= fp32 rate, can differ from it for logits within an ulp of the threshold)
                if (output) output[i * out_nodes + j] = 1.0f / (1.0f + expf(-acc[v]));
                action[i * out_nodes + j] = acc[v] <= logit_rate ? 1.0f : 0.0f;
            }
        }
    }
}

__global__ __launch_bounds__(256) void rn_finish_kernel(const float *__restrict__ partial, int splits, int64_t count,
                                                        float toggle_rate, float logit_rate, float *__restrict__ output,
                                                        float *__restrict__ action) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    float x = 0.0f;
    for (int z = 0; z < splits; ++z) x += partial[(int64_t)z * count + i];
    if (output) output[i] = 1.0f / (1.0f + expf(-x));
    action[i] = x <= logit_rate ? 1.0f : 0.0f;
}

// The same product for large batches: 128x128 output tile per workgroup, a 64x64 quadrant (four 32x32 MFMA
// accumulators) per wave -- half the LDS and global traffic per FLOP of the 64x64 kernel -- and the next K chunk's
// global loads issued into registers BEFORE the MFMA loop of the current one, so their latency hides behind 64
// matrix instructions per wave instead of behind other workgroups.  M >= 128, no K split.
static constexpr int kBM2 = 128, kBN2 = 128;

__global__ __launch_bounds__(256, 2) void rn_linear128_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                              int64_t m, int k, int out_nodes, float toggle_rate,
                                                              float logit_rate, float *__restrict__ output,
                                                              float *__restrict__ action) {
    __shared__ float s_a[kBM2][kLd];
    __shared__ float s_b[kBN2][kLd];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int64_t row0 = (int64_t)blockIdx.y * kBM2;
    const int col0 = blockIdx.x * kBN2;
    f32x16 acc[2][2];
#pragma unroll
    for (int im = 0; im < 2; ++im)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[im][jn][i] = 0.0f;
    // loader: 8 threads x float4 = one 32-float row segment, 32 rows per pass, four passes per operand
    const int lr = tid >> 3, lc = (tid & 7) * 4;
    float4 pa[4], pb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int r = lr + 32 * p;
            pa[p] = make_float4(0.f, 0.f, 0.f, 0.f);
            pb[p] = pa[p];
            if (row0 + r < m && k0 + lc < k) pa[p] = *reinterpret_cast<const float4 *>(a + (row0 + r) * k + k0 + lc);
            if (col0 + r < out_nodes && k0 + lc < k) pb[p] = *reinterpret_cast<const float4 *>(b + (int64_t)(col0 + r) * k + k0 + lc);
        }
    };
    fetch(0);
    const int i32 = lane & 31, kh = lane >> 5;
    for (int k0 = 0; k0 < k; k0 += kBK) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int r = lr + 32 * p;
            s_a[r][lc] = pa[p].x; s_a[r][lc + 1] = pa[p].y; s_a[r][lc + 2] = pa[p].z; s_a[r][lc + 3] = pa[p].w;
            s_b[r][lc] = pb[p].x; s_b[r][lc + 1] = pb[p].y; s_b[r][lc + 2] = pb[p].z; s_b[r][lc + 3] = pb[p].w;
        }
        __syncthreads();
        if (k0 + kBK < k) fetch(k0 + kBK);              // travels during the 64 MFMAs below
#pragma unroll
        for (int kk = 0; kk < kBK; kk += 2) {
            const float a0 = s_a[64 * wm + i32][kk + kh], a1 = s_a[64 * wm + 32 + i32][kk + kh];
            const float b0 = s_b[64 * wn + i32][kk + kh], b1 = s_b[64 * wn + 32 + i32][kk + kh];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int im = 0; im < 2; ++im)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            const int j = col0 + 64 * wn + 32 * jn + (lane & 31);
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int64_t i = row0 + 64 * wm + 32 * im + 8 * (v >> 2) + 4 * (lane >> 5) + (v & 3);
                if (i < m && j < out_nodes) {
                    if (output) output[i * out_nodes + j] = 1.0f / (1.0f + expf(-acc[im][jn][v]));
                    action[i * out_nodes + j] = acc[im][jn][v] <= logit_rate ? 1.0f : 0.0f;
                }
            }
        }
}

// K splits of the linear layer: 1 when the output tiles alone fill the chip
static int rn_splits(int64_t n, int dense, int out_nodes) {
    const int64_t tiles = ((n + kBM - 1) / kBM) * ((out_nodes + kBN - 1) / kBN);
    const int chunks = (dense + kBK - 1) / kBK;
    if (tiles >= 512) return 1;
    int64_t s = (1024 + tiles - 1) / tiles;
    if (s > chunks) s = chunks;
    if (s > 32) s = 32;
    return (int)(s < 1 ? 1 : s);
}

}  // namespace carle

using namespace carle;

extern "C" int64_t carle_agent_rn_workspace_floats(int64_t n, int h, int w, int out_nodes) {
    if (n < 0 || h < 8 || w < 8 || (h & 7) || (w & 7) || out_nodes < 1) return 0;
    const int dense = (h >> 2) * (w >> 2);
    const int splits = rn_splits(n, dense, out_nodes);
    return n * (int64_t)dense + (splits > 1 ? (int64_t)splits * n * out_nodes : 0);
}

extern "C" int carle_agent_rn_forward_f32(const float *obs, int64_t n, int h, int w, const float *conv1_w,
                                          const float *conv2_w, const float *lin_w, int out_nodes,
                                          float toggle_rate, float *workspace, float *output, float *action,
                                          void *stream) {
    if (!obs || !conv1_w || !conv2_w || !lin_w || !workspace || !action)
        return fail_arg(CARLE_E_NULL, "carle_agent_rn_forward_f32: NULL pointer");
    if (n < 0 || h < 8 || w < 8 || (h & 7) || (w & 7) || out_nodes < 1)
        return fail_arg(CARLE_E_DIMS, "carle_agent_rn_forward_f32: H and W must be multiples of 8 (>= 8), OUT >= 1, N >= 0");
    if (!aligned16(workspace) || !aligned16(lin_w))
        return fail_arg(CARLE_E_ALIGN, "carle_agent_rn_forward_f32: workspace and lin_w must be 16-byte aligned");
    if (n == 0) return 0;
    const int tiles = ((h + kTile - 1) / kTile) * ((w + kTile - 1) / kTile);
    const int64_t mtiles = (n + kBM - 1) / kBM;
    if (n > 65535 || mtiles > 65535) return fail_arg(CARLE_E_DIMS, "carle_agent_rn_forward_f32: batch too large");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(rn_features_kernel, dim3((unsigned)tiles, (unsigned)n), dim3(256), 0, s, obs, h, w, conv1_w,
                       conv2_w, workspace);
    const int dense = (h >> 2) * (w >> 2);
    const int splits = rn_splits(n, dense, out_nodes);
    const unsigned ntiles = (unsigned)((out_nodes + kBN - 1) / kBN);
    // threshold on the logit (used when the sigmoid values are not requested)
    const float logit_rate = toggle_rate <= 0.0f ? -INFINITY : toggle_rate >= 1.0f ? INFINITY
                                                 : (float)log((double)toggle_rate / (1.0 - (double)toggle_rate));
    const int64_t big_tiles = ((n + kBM2 - 1) / kBM2) * ((out_nodes + kBN2 - 1) / kBN2);
    if (n >= kBM2 && big_tiles >= 256) {
        hipLaunchKernelGGL(rn_linear128_kernel, dim3((unsigned)((out_nodes + kBN2 - 1) / kBN2), (unsigned)((n + kBM2 - 1) / kBM2)),
                           dim3(256), 0, s, workspace, lin_w, n, dense, out_nodes, toggle_rate, logit_rate, output, action);
    } else if (splits == 1) {
        hipLaunchKernelGGL(rn_linear_kernel<false>, dim3(ntiles, (unsigned)mtiles), dim3(256), 0, s, workspace, lin_w,
                           n, dense, out_nodes, toggle_rate, logit_rate, 0, nullptr, output, action);
    } else {
        const int chunks = (dense + kBK - 1) / kBK, per = (chunks + splits - 1) / splits;
        float *partial = workspace + n * (int64_t)dense;
        hipLaunchKernelGGL(rn_linear_kernel<true>, dim3(ntiles, (unsigned)mtiles, (unsigned)splits), dim3(256), 0, s,
                           workspace, lin_w, n, dense, out_nodes, toggle_rate, logit_rate, per, partial, nullptr, nullptr);
        const int64_t count = n * (int64_t)out_nodes;
        hipLaunchKernelGGL(rn_finish_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, partial, splits,
                           count, toggle_rate, logit_rate, output, action);
    }
    return check_launch("carle_agent_rn_forward_f32");
}

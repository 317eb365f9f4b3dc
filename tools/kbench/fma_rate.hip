// Issue rate of v_fma_f32 against v_pk_fma_f32 on gfx950: 1024 single-wave-per-SIMD-slot workgroups x 8 waves per SIMD,
// 16 independent accumulator chains per lane, 4096 iterations.  Prints FLOP/s of each form.
// Build + run on the GPU box:  hipcc -w -O3 -fno-slp-vectorize --offload-arch=gfx950 tools/kbench/fma_rate.hip -o /tmp/fma_rate && /tmp/fma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <bool PK>
__global__ __launch_bounds__(256) void fma_kernel(float *out, float a, float b, int iters) {
    if (PK) {
        f2 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = (f2){(float)threadIdx.x + i, (float)i};
        const f2 av = {a, a + 1.0f}, bv = {b, b + 1.0f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_elementwise_fma(acc[i], av, bv);
        }
        f2 s = acc[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) s += acc[i];
        out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
    } else {
        float acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = (float)threadIdx.x + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = fmaf(acc[i], a, b);
        }
        float s = acc[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) s += acc[i];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}

// the conv2 form: accumulator pair += broadcast(lane value) * (scalar pair)   [op_sel broadcast + SGPR pair operand]
__global__ __launch_bounds__(256) void fma_bcast_kernel(float *out, const float *__restrict__ w, int iters) {
    f2 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (f2){(float)threadIdx.x + i, (float)i};
    float x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = 1.0f + 1e-6f * (float)(threadIdx.x + i);
    f2 wv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wv[i] = (f2){w[2 * i], w[2 * i + 1]};      // uniform: SGPR pairs
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_elementwise_fma((f2){x[(i + r) & 3], x[(i + r) & 3]}, wv[r], acc[i]);
    }
    f2 s = acc[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}

template <bool PK>
static void run(const char *name, float *out) {
    const int blocks = 256 * 8, iters = 4096;          // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(fma_kernel<PK>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(fma_kernel<PK>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * 64.0 * iters * (double)blocks * 256.0;     // 64 fma per lane and iteration in both forms
    const double instr = (PK ? 32.0 : 64.0) * iters * (double)blocks * 4.0;   // wave instructions
    printf("%-14s %8.3f ms  %7.1f TFLOP/s  %6.2f cycles per wave instruction and SIMD at 2.4 GHz\n", name, ms,
           flop / ms * 1e-9, ms * 1e-3 * 2.4e9 * 1024.0 / instr);
}

static void run_bcast(float *out, const float *w) {
    const int blocks = 256 * 8, iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(fma_bcast_kernel, dim3(blocks), dim3(256), 0, 0, out, w, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(fma_bcast_kernel, dim3(blocks), dim3(256), 0, 0, out, w, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * 64.0 * iters * (double)blocks * 256.0;
    const double instr = 32.0 * iters * (double)blocks * 4.0;
    printf("%-14s %8.3f ms  %7.1f TFLOP/s  %6.2f cycles per wave instruction and SIMD at 2.4 GHz\n", "pk bcast*sgpr", ms,
           flop / ms * 1e-9, ms * 1e-3 * 2.4e9 * 1024.0 / instr);
}

int main() {
    float *out, *w;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    hipMalloc(&w, 64);
    hipMemset(w, 0, 64);
    run_bcast(out, w);
    run<false>("v_fma_f32", out);
    run<true>("v_pk_fma_f32", out);
    run<false>("v_fma_f32", out);
    run<true>("v_pk_fma_f32", out);
    hipFree(out);
    return 0;
}

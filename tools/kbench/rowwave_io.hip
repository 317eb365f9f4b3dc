// I/O skeleton of a wave-per-grid 64x64 fp32 step with lane = row: does the "row per load instruction" form
// (dword loads / stores, 256 contiguous bytes per wave instruction, cells -> bits by v_cmp into an SGPR pair, bits ->
// cells by v_cndmask from an SGPR pair) stream as fast as the 16-byte-per-lane form of ca_step_64_kernel?
// N grids, one wavefront per grid, WG = 16 wavefronts (one workgroup per CU at N = 4096), state + action in, state out
// (192 MiB at N = 4096); the "generation" is the toggle alone.  Build + run on the GPU box:
//   hipcc -w -O3 --offload-arch=gfx950 tools/kbench/rowwave_io.hip -o /tmp/rowwave_io && /tmp/rowwave_io 4096
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// v_writelane_b32 through the LLVM intrinsic (this clang has no __builtin_amdgcn_writelane): the compiler sees the
// instruction and pads its hazards, which it would not do for inline asm
extern "C" __device__ int llvm_writelane(int val, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

template <int CHUNK, int WAVES, int LDS_KB = 0>
__global__ __launch_bounds__(64 * WAVES) void rows_kernel(const float *__restrict__ u_in, const float *__restrict__ act,
                                                          float *__restrict__ u_out, int64_t n) {
    // LDS_KB: a dummy allocation that fixes the workgroups per CU (the fused kernel's tables + planes)
    __shared__ float dummy[LDS_KB * 256 + 1];
    if (LDS_KB) dummy[threadIdx.x] = (float)threadIdx.x;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: no waterfall loops
    const int64_t g = (int64_t)blockIdx.x * WAVES + wv;
    if (g >= n) return;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(u_in) + g * 16384), 0, 16384, 0x00020000);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(act) + g * 16384), 0, 16384, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char *>(u_out) + g * 16384, 0, 16384, 0x00020000);
    int lo = 0, hi = 0;
    unsigned long long any = 0, notone = 0;
#pragma unroll
    for (int c = 0; c < 64 / CHUNK; ++c) {
        float s[CHUNK], a[CHUNK];
#pragma unroll
        for (int r = 0; r < CHUNK; ++r)
            s[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4 + r * 256, c * CHUNK * 256, 2));
#pragma unroll
        for (int r = 0; r < CHUNK; ++r)
            a[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, lane * 4 + r * 256, c * CHUNK * 256, 2));
#pragma unroll
        for (int r = 0; r < CHUNK; ++r) {
            // (an opaque float: with the bit cast visible, hipcc rewrites x != 0.0f into v_and + v_cmp_ne_u32)
            asm volatile("" : "+v"(s[r]));
            asm volatile("" : "+v"(a[r]));
            const unsigned long long ms = __ballot(s[r] != 0.0f), ma = __ballot(a[r] != 0.0f);
            any |= ma;
            notone |= __ballot(a[r] != 1.0f);
            asm volatile("" : "+s"(any), "+s"(notone));      // fold now (the scheduler otherwise parks the masks in VGPR lanes)
            const unsigned long long x = ms ^ ma;
            lo = llvm_writelane((int)(uint32_t)x, c * CHUNK + r, lo);
            hi = llvm_writelane((int)(uint32_t)(x >> 32), c * CHUNK + r, hi);
        }
    }
    // stand-in for the generation: rotate each row by one (keeps lo / hi live as VGPRs)
    const uint64_t row = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
    const uint64_t nx = (row << 1) | (row >> 63);
    const int nlo = (int)(uint32_t)nx, nhi = (int)(uint32_t)(nx >> 32);
#pragma unroll
    for (int r = 0; r < 64; ++r) {
        const unsigned long long m = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(nhi, r) << 32) |
                                     (uint32_t)__builtin_amdgcn_readlane(nlo, r);
        const float o = __builtin_amdgcn_inverse_ballot_w64(m) ? 1.0f : 0.0f;      // v_cndmask_b32 from the SGPR pair
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, o), ro, lane * 4 + (r & 15) * 256, (r >> 4) * 4096, 0);
    }
    if (lane == 0 && (any == 123 || notone == 77)) u_out[0] = LDS_KB ? dummy[(threadIdx.x + 1) & 1023] : 2.0f;     // keeps the predicates alive
}

// Variant: chunks of 16 rows taken in a per-grid rotated order (wavefronts that run in lockstep then touch all four
// 4-KiB slices of their grids at any time, not the same one), the NEXT chunk's loads issued before the current chunk is
// converted (two register sets).
template <int WAVES, int LDS_KB, bool ROT, bool PIPE>
__global__ __launch_bounds__(64 * WAVES) void rows2_kernel(const float *__restrict__ u_in, const float *__restrict__ act,
                                                           float *__restrict__ u_out, int64_t n) {
    __shared__ float dummy[LDS_KB * 256 + 1];
    if (LDS_KB) dummy[threadIdx.x] = (float)threadIdx.x;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t g = (int64_t)blockIdx.x * WAVES + wv;
    if (g >= n) return;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(u_in) + g * 16384), 0, 16384, 0x00020000);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(act) + g * 16384), 0, 16384, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char *>(u_out) + g * 16384, 0, 16384, 0x00020000);
    const int rot = ROT ? (int)(g & 3) : 0;
    int lo = 0, hi = 0;
    unsigned long long any = 0, notone = 0;
    float s[2][16], a[2][16];
    auto issue = [&](int c, float (&sv)[16], float (&av)[16]) __attribute__((always_inline)) {
        const int cc = (c + rot) & 3;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            sv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4 + r * 256, cc * 4096, 2));
#pragma unroll
        for (int r = 0; r < 16; ++r)
            av[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, lane * 4 + r * 256, cc * 4096, 2));
    };
    // rows of chunk cc land in lanes 16 * c + r of a ROTATED row order; the generation stand-in does not care, the real
    // kernel would un-rotate with one DPP wave rotate (or write lane 16 * cc + r: the lane index is then an SGPR)
    auto convert = [&](int c, float (&sv)[16], float (&av)[16]) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            asm volatile("" : "+v"(sv[r]));
            asm volatile("" : "+v"(av[r]));
            const unsigned long long ms = __ballot(sv[r] != 0.0f), ma = __ballot(av[r] != 0.0f);
            any |= ma;
            notone |= __ballot(av[r] != 1.0f);
            asm volatile("" : "+s"(any), "+s"(notone));
            const unsigned long long x = ms ^ ma;
            lo = llvm_writelane((int)(uint32_t)x, c * 16 + r, lo);
            hi = llvm_writelane((int)(uint32_t)(x >> 32), c * 16 + r, hi);
        }
    };
    if (PIPE) {
        issue(0, s[0], a[0]);
        issue(1, s[1], a[1]);
        convert(0, s[0], a[0]);
        issue(2, s[0], a[0]);
        convert(1, s[1], a[1]);
        issue(3, s[1], a[1]);
        convert(2, s[0], a[0]);
        convert(3, s[1], a[1]);
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) { issue(c, s[0], a[0]); convert(c, s[0], a[0]); }
    }
    const uint64_t row = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
    const uint64_t nx = (row << 1) | (row >> 63);
    const int nlo = (int)(uint32_t)nx, nhi = (int)(uint32_t)(nx >> 32);
#pragma unroll
    for (int r = 0; r < 64; ++r) {
        const unsigned long long m = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(nhi, r) << 32) |
                                     (uint32_t)__builtin_amdgcn_readlane(nlo, r);
        const float o = __builtin_amdgcn_inverse_ballot_w64(m) ? 1.0f : 0.0f;
        const int cc = ((r >> 4) + rot) & 3;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, o), ro, lane * 4 + (r & 15) * 256, cc * 4096, 0);
    }
    if (lane == 0 && (any == 123 || notone == 77)) u_out[0] = LDS_KB ? dummy[(threadIdx.x + 1) & 1023] : 2.0f;
}

// reference form: 16-byte accesses, nibble conversion (what ca_step_64_kernel does), same toggle + rotate
__device__ __forceinline__ uint32_t nz(uint32_t b) { return min(b & 0x7FFFFFFFu, 1u); }
__global__ __launch_bounds__(64) void quads_kernel(const float *__restrict__ u_in, const float *__restrict__ act,
                                                   float *__restrict__ u_out, int64_t n) {
    const int lane = threadIdx.x & 63;
    const int64_t g = blockIdx.x;
    const u32x4 *ps = reinterpret_cast<const u32x4 *>(u_in + g * 4096) + lane;
    const u32x4 *pa = reinterpret_cast<const u32x4 *>(act + g * 4096) + lane;
    u32x4 *po = reinterpret_cast<u32x4 *>(u_out + g * 4096) + lane;
    uint64_t patch = 0;
    u32x4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = __builtin_nontemporal_load(ps + 64 * i);
#pragma unroll
    for (int i = 0; i < 16; ++i)
        patch |= (uint64_t)(nz(v[i].x) | nz(v[i].y) << 1 | nz(v[i].z) << 2 | nz(v[i].w) << 3) << (4 * i);
    asm volatile("" : "+v"(patch));
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = __builtin_nontemporal_load(pa + 64 * i);
#pragma unroll
    for (int i = 0; i < 16; ++i)
        patch ^= (uint64_t)(nz(v[i].x) | nz(v[i].y) << 1 | nz(v[i].z) << 2 | nz(v[i].w) << 3) << (4 * i);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t nib = (uint32_t)(patch >> (4 * i)) & 15u;
        u32x4 o;
        o.x = (nib & 1u) ? 0x3F800000u : 0u; o.y = (nib & 2u) ? 0x3F800000u : 0u;
        o.z = (nib & 4u) ? 0x3F800000u : 0u; o.w = (nib & 8u) ? 0x3F800000u : 0u;
        po[64 * i] = o;
    }
}

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 4096;
    const size_t cells = (size_t)n * 4096;
    float *u[2], *act[6];
    for (auto &p : u) hipMalloc(&p, cells * 4);
    std::vector<float> h(cells);
    for (size_t i = 0; i < cells; ++i) h[i] = (float)((i * 2654435761u >> 13) & 1u);
    hipMemcpy(u[0], h.data(), cells * 4, hipMemcpyHostToDevice);
    for (int k = 0; k < 6; ++k) {
        hipMalloc(&act[k], cells * 4);
        for (size_t i = 0; i < cells; ++i) h[i] = (float)(((i + k) * 40503u >> 9) % 10u == 0u);
        hipMemcpy(act[k], h.data(), cells * 4, hipMemcpyHostToDevice);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto launch) {
        for (int rep = 0; rep < 3; ++rep) {
            for (int i = 0; i < 6; ++i) launch(i);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int i = 0; i < 300; ++i) launch(i);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double us = ms / 300 * 1e3;
            printf("N=%lld %-28s %7.2f us  %5.2f TB/s  frac %.3f\n", (long long)n, name, us, cells * 12.0 / us / 1e6, cells * 12.0 / us / 8e6);
        }
    };
    run("quads (16 B/lane, nibbles)", [&](int i) { hipLaunchKernelGGL(quads_kernel, dim3(n), dim3(64), 0, 0, u[i & 1], act[i % 6], u[(i & 1) ^ 1], n); });
    run("rows chunk 16, WG 16 waves", [&](int i) { hipLaunchKernelGGL((rows_kernel<16, 16>), dim3((n + 15) / 16), dim3(1024), 0, 0, u[i & 1], act[i % 6], u[(i & 1) ^ 1], n); });
    run("rows chunk 8, WG 16 waves", [&](int i) { hipLaunchKernelGGL((rows_kernel<8, 16>), dim3((n + 15) / 16), dim3(1024), 0, 0, u[i & 1], act[i % 6], u[(i & 1) ^ 1], n); });
    run("rows chunk 32, WG 16 waves", [&](int i) { hipLaunchKernelGGL((rows_kernel<32, 16>), dim3((n + 15) / 16), dim3(1024), 0, 0, u[i & 1], act[i % 6], u[(i & 1) ^ 1], n); });
    run("rows chunk 16, WG 1 wave", [&](int i) { hipLaunchKernelGGL((rows_kernel<16, 1>), dim3(n), dim3(64), 0, 0, u[i & 1], act[i % 6], u[(i & 1) ^ 1], n); });
    run("rows chunk 16, WG 4 waves", [&](int i) { hipLaunchKernelGGL((rows_kernel<16, 4>), dim3((n + 3) / 4), dim3(256), 0, 0, u[i & 1], act[i % 6], u[(i & 1) ^ 1], n); });
    run("rows 16, WG 16 waves, 1 WG/CU", [&](int i) { hipLaunchKernelGGL((rows_kernel<16, 16, 100>), dim3((n + 15) / 16), dim3(1024), 0, 0, u[i & 1], act[i % 6], u[(i & 1) ^ 1], n); });
    run("rows 16, WG 8 waves, 2 WG/CU", [&](int i) { hipLaunchKernelGGL((rows_kernel<16, 8, 60>), dim3((n + 7) / 8), dim3(512), 0, 0, u[i & 1], act[i % 6], u[(i & 1) ^ 1], n); });
    run("rows 16, WG 4 waves, 4 WG/CU", [&](int i) { hipLaunchKernelGGL((rows_kernel<16, 4, 36>), dim3((n + 3) / 4), dim3(256), 0, 0, u[i & 1], act[i % 6], u[(i & 1) ^ 1], n); });
    run("rows 16, WG 2 waves, 8 WG/CU", [&](int i) { hipLaunchKernelGGL((rows_kernel<16, 2, 18>), dim3((n + 1) / 2), dim3(128), 0, 0, u[i & 1], act[i % 6], u[(i & 1) ^ 1], n); });
    run("rows 16, WG 1 wave, 16 WG/CU", [&](int i) { hipLaunchKernelGGL((rows_kernel<16, 1, 9>), dim3(n), dim3(64), 0, 0, u[i & 1], act[i % 6], u[(i & 1) ^ 1], n); });
#define R2(W, KB, ROT, PIPE, name) run(name, [&](int i) { hipLaunchKernelGGL((rows2_kernel<W, KB, ROT, PIPE>), dim3((n + W - 1) / W), dim3(64 * W), 0, 0, u[i & 1], act[i % 6], u[(i & 1) ^ 1], n); })
    R2(16, 100, true, false, "rows2 W16 rot");
    R2(16, 100, false, true, "rows2 W16 pipe");
    R2(16, 100, true, true, "rows2 W16 rot+pipe");
    R2(4, 36, true, false, "rows2 W4 rot");
    R2(4, 36, true, true, "rows2 W4 rot+pipe");
    R2(1, 9, true, true, "rows2 W1 rot+pipe");
    R2(1, 9, false, true, "rows2 W1 pipe");
    // correctness of the skeleton on grid 0, row 0 (out = rotl(state ^ action, 1))
    hipLaunchKernelGGL((rows_kernel<16, 16>), dim3((n + 15) / 16), dim3(1024), 0, 0, u[0], act[0], u[1], n);
    std::vector<float> a(64), b(64), c(64);
    hipMemcpy(a.data(), u[0] + 4096 * 3 + 64 * 5, 256, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), act[0] + 4096 * 3 + 64 * 5, 256, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), u[1] + 4096 * 3 + 64 * 5, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int j = 0; j < 64; ++j) bad += (c[(j + 1) & 63] != (float)((a[j] != 0) ^ (b[j] != 0)));
    printf("skeleton check: %s\n", bad ? "MISMATCH" : "ok");
    return bad != 0;
}

// A/B microbenchmark for the bit-packed 512x512 step kernel (development tool).
// build (after make -C carle_amd/csrc; the rest of the library is linked, this kernel file is compiled in):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/kbench/ca_b1_variants tools/kbench/ca_b1_variants.hip \
//         -Lcarle_amd/lib -lcarle_hip -Wl,-rpath,'$ORIGIN/../../carle_amd/lib'
// the library exports the same symbols: this build's copies get their own names
#define ca_step_b1_kernel ca_step_b1_kernel_dbg
#define carle_ca_step_b1 carle_ca_step_b1_dbg
#define carle_ca_multistep_b1 carle_ca_multistep_b1_dbg
#define ca_multistep_b1_kernel ca_multistep_b1_kernel_dbg
#define carle_pack_f32_b1 carle_pack_f32_b1_dbg
#define carle_unpack_b1_f32 carle_unpack_b1_f32_dbg
#define pack_f32_b1_kernel pack_f32_b1_kernel_dbg
#define unpack_b1_f32_kernel unpack_b1_f32_kernel_dbg
#include "../../carle_amd/csrc/ca_step_b1.hip"
#include <algorithm>
#include <vector>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

using namespace carle;

using v4u = __attribute__((ext_vector_type(4))) unsigned int;
template <int WW, bool VIA_LDS>
__global__ __launch_bounds__(64) void copy_strip_kernel(const uint64_t *in, uint64_t *out, int64_t n) {
    const int lane = threadIdx.x;
    const int64_t wid = blockIdx.x;
    const v4u *s4 = reinterpret_cast<const v4u *>(in + wid * 64 * WW);
    v4u *d4 = reinterpret_cast<v4u *>(out + wid * 64 * WW);
    if (VIA_LDS) {   // lane = row access pattern (4 x 16 B at stride 64 B)
#pragma unroll
        for (int i = 0; i < WW / 2; ++i) d4[lane * (WW / 2) + i] = s4[lane * (WW / 2) + i];
    } else {
#pragma unroll
        for (int i = 0; i < WW / 2; ++i) __builtin_nontemporal_store(__builtin_nontemporal_load(s4 + 64 * i + lane), d4 + 64 * i + lane);
    }
}
// big-block flat copy (reference ceiling)
__global__ __launch_bounds__(256) void copy_flat_kernel(const v4u *in, v4u *out, int64_t n16) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256)
        __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 4096;
    const int rounds = 12, reps = 20;
    constexpr int WW = 8;
    const size_t words = (size_t)n * 512 * WW;
    uint64_t *a, *b, *act; int32_t *st;
    CK(hipMalloc(&a, words * 8)); CK(hipMalloc(&b, words * 8)); CK(hipMalloc(&st, 64)); CK(hipMemset(st, 0, 64));
    CK(hipMalloc(&act, (size_t)n * 64 * 8));
    std::vector<uint64_t> h(words);
    srand(1);
    for (auto &v : h) v = ((uint64_t)rand() << 33) ^ ((uint64_t)rand() << 11) ^ rand();
    CK(hipMemcpy(a, h.data(), words * 8, hipMemcpyHostToDevice));
    CK(hipMemset(act, 0x11, (size_t)n * 64 * 8));
    const uint32_t high = (1u << 12) | (1u << 15) | 0xC, life = (1u << 12) | 0xC;
    struct V { const char *name; int id; } vs[] = {{"flat_copy", 0}, {"strip_copy_direct", 1}, {"strip_copy_rowlane", 2},
        {"step_noact_highlife", 3}, {"step_act_highlife", 4}, {"step_act_nostatus", 7}, {"step_noact_life", 5}, {"step_noact_dense", 6},
        {"act1_kernel_window_h0", 12}, {"act1_kernel_h64_nostatus", 13}, {"act1_h64_win_rows_0_63", 14}};
    const int nv = sizeof(vs) / sizeof(vs[0]);
    std::vector<std::vector<float>> t(nv);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < rounds; ++r)
        for (int v = 0; v < nv; ++v) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; ++i) {
                switch (vs[v].id) {
                    case 0: hipLaunchKernelGGL(copy_flat_kernel, dim3(2048), dim3(256), 0, 0, (const v4u *)a, (v4u *)b, (int64_t)(words / 2)); break;
                    case 1: hipLaunchKernelGGL((copy_strip_kernel<WW, false>), dim3((unsigned)(n * WW)), dim3(64), 0, 0, a, b, n); break;
                    case 2: hipLaunchKernelGGL((copy_strip_kernel<WW, true>), dim3((unsigned)(n * WW)), dim3(64), 0, 0, a, b, n); break;
                    case 3: carle_ca_step_b1(a, nullptr, b, n, 512, 512, 0, 0, high, st, 0, nullptr); break;
                    case 4: carle_ca_step_b1(a, act, b, n, 512, 512, 64, 64, high, st, 0, nullptr); break;
                    case 7: carle_ca_step_b1(a, act, b, n, 512, 512, 64, 64, high, nullptr, 0, nullptr); break;
                    case 5: carle_ca_step_b1(a, nullptr, b, n, 512, 512, 0, 0, life, st, 0, nullptr); break;
                    case 6: carle_ca_step_b1(a, nullptr, b, n, 512, 512, 0, 0, 0x3FFFF, st, 0, nullptr); break;
                    case 12: hipLaunchKernelGGL((ca_step_b1_kernel<8, 1, kRuleHighLife>), dim3(8168), dim3(64), 0, 0,
                                                a, act, b, n, 0, 64, high, st, 0, 0, 8); break;      // ACT=1 code, no strip meets the window
                    case 13: hipLaunchKernelGGL((ca_step_b1_kernel<8, 1, kRuleHighLife>), dim3(8168), dim3(64), 0, 0,
                                                a, act, b, n, 64, 64, high, (int32_t *)nullptr, 0, 0, 8); break;
                    case 14: hipLaunchKernelGGL((ca_step_b1_kernel<8, 1, kRuleHighLife>), dim3(8168), dim3(64), 0, 0,
                                                a, act, b, n, 64, 64, high, (int32_t *)nullptr, 0, 5, 8); break;   // strips rotated by 5
                    case 8: hipLaunchKernelGGL((ca_step_b1_kernel<8, 0, kRuleHighLife>), dim3((unsigned)(n * 5)), dim3(64), 0, 0,
                                               a, (const uint64_t *)nullptr, b, n, 0, 0, high, st, 0, 5, 5); break;
                    case 10: hipLaunchKernelGGL((ca_step_b1_kernel<8, 0, kRuleHighLife>), dim3((unsigned)(n * 5)), dim3(64), 0, 0,
                                                a, (const uint64_t *)nullptr, b, n, 0, 0, high, st, 0, 0, 5); break;
                    case 11: hipLaunchKernelGGL((ca_step_b1_kernel<8, 0, kRuleHighLife>), dim3((unsigned)(n * 4)), dim3(64), 0, 0,
                                                a, (const uint64_t *)nullptr, b, n, 0, 0, high, st, 0, 0, 4); break;
                    case 9: hipLaunchKernelGGL((ca_step_b1_kernel<8, 2, kRuleHighLife>), dim3((unsigned)(n * 3)), dim3(64), 0, 0,
                                               a, act, b, n, 64, 64, high, st, 0, 2, 3); break;
                }
                std::swap(a, b);
            }
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) t[v].push_back(ms * 1000.f / reps);
        }
    const double bytes = (double)words * 16;
    for (int v = 0; v < nv; ++v) {
        std::sort(t[v].begin(), t[v].end());
        float med = t[v][t[v].size() / 2];
        printf("%-22s median %7.2f us  %6.3f TB/s\n", vs[v].name, med, bytes / med * 1e-6);
    }
#ifdef CARLE_B1_STAMPS
    {   // per-strip wave timing of the action launch: median cycles start->loads, loads->computed, ->stored
        const int64_t blocks = (n + 7) / 8 * 8 * 8;
        unsigned long long *st4; CK(hipMalloc(&st4, blocks * 4 * 8)); CK(hipMemset(st4, 0, blocks * 4 * 8));
        for (int mode = 0; mode < 2; ++mode) {
            CK(hipMemcpyToSymbol(HIP_SYMBOL(g_b1_stamps), &st4, sizeof(st4)));
            if (mode == 0) carle_ca_step_b1(a, nullptr, b, n, 512, 512, 0, 0, high, st, 0, nullptr);
            else carle_ca_step_b1(a, act, b, n, 512, 512, 64, 64, high, st, 0, nullptr);
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> hs(blocks * 4);
            CK(hipMemcpy(hs.data(), st4, blocks * 4 * 8, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull, t1 = 0;
            for (auto v : hs) if (v) { t0 = std::min(t0, v); t1 = std::max(t1, v); }
            printf("%s launch: span %llu cycles\n", mode ? "action" : "no-action", t1 - t0);
            for (int sidx = 0; sidx < 8; ++sidx) {
                std::vector<double> d[4];
                for (int64_t blk = 0; blk < blocks; ++blk) {
                    if ((int)((blk >> 3) % 8) != sidx) continue;
                    const unsigned long long *p = &hs[blk * 4];
                    if (!p[0] || !p[3]) continue;
                    d[0].push_back((double)(p[1] - p[0])); d[1].push_back((double)(p[2] - p[1]));
                    d[2].push_back((double)(p[3] - p[2])); d[3].push_back((double)(p[3] - p[0]));
                }
                auto med = [](std::vector<double> &v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
                printf("   strip %d: loads %7.0f  compute %7.0f  store-issue %6.0f  total %7.0f  (%zu waves)\n", sidx,
                       med(d[0]), med(d[1]), med(d[2]), med(d[3]), d[0].size());
            }
        }
    }
#endif
    return 0;
}

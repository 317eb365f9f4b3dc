// Phase-time diagnostic of the fused CA + RND2D kernel (development tool, not shipped): the kernel
// file is compiled in with CARLE_RND_STAMPS, wave 0 of every workgroup stamps s_memtime at the phase
// boundaries of each grid; prints the median cycles per phase over workgroups, per iteration.
// build (after make -C carle_amd/csrc):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCARLE_RND_STAMPS -o tools/kbench/ca_rnd_variants \
//         tools/kbench/ca_rnd_variants.hip -Lcarle_amd/lib -lcarle_hip -Wl,-rpath,'$ORIGIN/../../carle_amd/lib'
// the library exports the same kernel: give this build's copy its own names so that the stamping
// code object is the one that is launched
#define ca_step_rnd64_kernel ca_step_rnd64_kernel_dbg
#define ca_step_rnd64 ca_step_rnd64_dbg
#include "../../carle_amd/csrc/ca_rnd_fused.hip"
#include <algorithm>
#include <vector>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 4096;
    const int ring = 6;
    const size_t cells = (size_t)n * 4096;
    float *a, *b, *act[8], *pp, *rp, *ws, *bonus;
    int32_t *st;
    CK(hipMalloc(&a, cells * 4)); CK(hipMalloc(&b, cells * 4)); CK(hipMalloc(&st, 64)); CK(hipMemset(st, 0, 64));
    std::vector<float> h(cells);
    srand(1);
    for (auto &v : h) v = (rand() & 1) ? 1.0f : 0.0f;
    CK(hipMemcpy(a, h.data(), cells * 4, hipMemcpyHostToDevice));
    for (int r = 0; r < ring; ++r) {
        for (auto &v : h) v = (rand() % 10 == 0) ? 1.0f : 0.0f;
        CK(hipMalloc(&act[r], cells * 4));
        CK(hipMemcpy(act[r], h.data(), cells * 4, hipMemcpyHostToDevice));
    }
    std::vector<float> par(2048);
    for (auto &v : par) v = (rand() % 2001 - 1000) * 2e-4f;
    CK(hipMalloc(&pp, 8192)); CK(hipMalloc(&rp, 8192));
    CK(hipMemcpy(pp, par.data(), 1117 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(rp, par.data() + 900, 1079 * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&ws, carle::kFusedWsTotal * 4 + 1024)); CK(hipMalloc(&bonus, (n + 1) * 4));
    const uint32_t life = (1u << 12) | 0xC;
    const size_t nst = (size_t)(n + 1) * 8 * 16;
    unsigned long long *stamps;
    CK(hipMalloc(&stamps, nst * 8)); CK(hipMemset(stamps, 0, nst * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int NS = 10;
    const char *names[NS] = {"start", "A nibbles (loads waited)", "A rows stored", "barrier1", "B | linear(prev) done",
                             "barrier2", "C issued", "D done", "barrier3", "conv2 done"};
    const char *cfgs[] = {"0"};
    for (const char *cfg : cfgs) {
        if (atoll(cfg) > 0) setenv("CARLE_RND_BLOCKS", cfg, 1); else unsetenv("CARLE_RND_BLOCKS");
#ifdef CARLE_RND_STAMPS
        unsigned long long *null_p = nullptr;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_rnd_stamps), &null_p, sizeof(null_p)));
#endif
        for (int i = 0; i < 20; ++i) {
            carle::ca_step_rnd64(a, act[i % ring], b, n, 64, life, st, 0, pp, rp, ws, bonus, i > 0, 0);
            std::swap(a, b);
        }
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 100; ++i) {
            carle::ca_step_rnd64(a, act[i % ring], b, n, 64, life, st, 0, pp, rp, ws, bonus, true, 0);
            std::swap(a, b);
        }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("== WG/CU=%d conv1 batch=%d blocks=%s: %.2f us per launch (%s)\n", CARLE_RND_WG_PER_CU, CARLE_RND_CONV1_BATCH, cfg, ms * 10.f,
#ifdef CARLE_RND_STAMPS
               "stamping build, stamps off"
#else
               "plain build"
#endif
        );
#ifdef CARLE_RND_STAMPS
        CK(hipMemset(stamps, 0, nst * 8));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_rnd_stamps), &stamps, sizeof(stamps)));
        carle::ca_step_rnd64(a, act[0], b, n, 64, life, st, 0, pp, rp, ws, bonus, true, 0);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> hs(nst);
        CK(hipMemcpy(hs.data(), stamps, nst * 8, hipMemcpyDeviceToHost));
        const int64_t blocks = atoll(cfg) > 0 ? std::min<int64_t>(atoll(cfg), n + 1) : std::min<int64_t>((n + 3) / 4, 256 * CARLE_RND_WG_PER_CU);
        unsigned long long tmin = ~0ull, tmax = 0;
        for (int64_t blk = 0; blk < blocks; ++blk)
            for (int it = 0; it < 8; ++it)
                for (int k = 0; k < NS; ++k) {
                    unsigned long long v = hs[((size_t)blk * 8 + it) * 16 + k];
                    if (v) { tmin = std::min(tmin, v); tmax = std::max(tmax, v); }
                }
        printf("   kernel span by stamps: %llu cycles\n", tmax - tmin);
        for (int it = 0; it < 5; ++it) {
            std::vector<std::vector<double>> d(NS);
            std::vector<double> starts;
            for (int64_t blk = 0; blk < blocks; ++blk) {
                const unsigned long long *p = &hs[((size_t)blk * 8 + it) * 16];
                if (!p[0] || !p[NS - 1]) continue;
                starts.push_back((double)(p[0] - tmin));
                for (int k = 1; k < NS; ++k) d[k].push_back((double)(p[k] - p[k - 1]));
                d[0].push_back((double)(p[NS - 1] - p[0]));
            }
            if (starts.empty()) continue;
            auto med = [](std::vector<double> &v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
            printf("   iteration %d (%zu workgroups): start @%.0f (median)  total %.0f cycles\n", it, starts.size(),
                   med(starts), med(d[0]));
            for (int k = 1; k < NS; ++k) printf("      -> %-26s %8.0f\n", names[k], med(d[k]));
        }
#endif
    }
    return 0;
}

// Kernel A/B microbenchmark (development tool, not shipped): variants of the 64x64 fp32 CA
// step kernel timed interleaved in ONE process (cdna_hip_programming.md 5.4 rule 24).
// build (after make -C carle_amd/csrc; the rest of the library is linked, this kernel file is compiled in):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/kbench/ca_f32_variants tools/kbench/ca_f32_variants.hip \
//         -Lcarle_amd/lib -lcarle_hip -Wl,-rpath,'$ORIGIN/../../carle_amd/lib'
#include "../../carle_amd/csrc/ca_step_cells.hip"
#include <algorithm>
#include <vector>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

typedef void (*launch_fn)(const float *, const float *, float *, int64_t, uint32_t, int32_t *, hipStream_t);

template <bool NL, bool NS, int BLOCK, int LDS = 0>
void launch(const float *in, const float *act, float *out, int64_t n, uint32_t rule, int32_t *st, hipStream_t s) {
    const int wpb = BLOCK / 64;
    hipLaunchKernelGGL((carle::ca_step_64_kernel<carle::F32Cells, carle::F32Cells, true, NL, NS, BLOCK>), dim3((unsigned)((n + wpb - 1) / wpb)), dim3(BLOCK), LDS, s,
                       in, act, out, n, 64, rule, st, 0);
}

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 4096;
    const int rounds = argc > 2 ? atoi(argv[2]) : 20, reps = 50, ring = 6;
    const size_t cells = (size_t)n * 4096;
    float *a, *b, *act[8];
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
    struct V { const char *name; launch_fn f; } vs[] = {
        {"plain", launch<false, false, 256>}, {"nt_load", launch<true, false, 256>},
        {"nt_store", launch<false, true, 256>}, {"nt_both", launch<true, true, 256>},
        {"nt_b64", launch<true, true, 64>}, {"nt_b128", launch<true, true, 128>}, {"nt_b512", launch<true, true, 512>},
        {"nt_occ1", launch<true, true, 256, 100 * 1024>}, {"nt_occ2", launch<true, true, 256, 60 * 1024>},
        {"nt_occ3", launch<true, true, 256, 45 * 1024>},
    };
    const int nv = sizeof(vs) / sizeof(vs[0]);
    std::vector<std::vector<float>> t(nv);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t life = (1u << 12) | 0xC;
    for (int r = 0; r < rounds; ++r)
        for (int v = 0; v < nv; ++v) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; ++i) { vs[v].f(a, act[i % ring], b, n, life, st, 0); std::swap(a, b); }
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) t[v].push_back(ms * 1000.f / reps);
        }
    const double bytes = (double)n * 4096 * 12;
    for (int v = 0; v < nv; ++v) {
        std::sort(t[v].begin(), t[v].end());
        float med = t[v][t[v].size() / 2], mn = t[v][0];
        printf("%-10s median %7.2f us  min %7.2f us   %6.3f TB/s (median)  %6.3f TB/s (best)\n", vs[v].name, med, mn,
               bytes / med * 1e-6, bytes / mn * 1e-6);
    }
    return 0;
}

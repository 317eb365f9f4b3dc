// Can the bandwidth-bound CA step of step t+1 overlap the compute-bound RND2D forward of step t
// when they are enqueued on two HIP streams?  (development experiment)
#include "../../carle_amd/csrc/abi.hip"
#include "../../carle_amd/csrc/ca_step_cells.hip"
#include "../../carle_amd/csrc/ca_step_b1.hip"
#include "../../carle_amd/csrc/nets_staged.hip"
#include "../../carle_amd/csrc/nets_fused.hip"
#include <vector>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
int main() {
    const int64_t n = 4096; const size_t cells = (size_t)n * 4096;
    float *u[3], *act, *ws, *bonus, *pp, *rp; int32_t *st;
    for (int i = 0; i < 3; ++i) CK(hipMalloc(&u[i], cells * 4));
    CK(hipMalloc(&act, cells * 4)); CK(hipMalloc(&st, 64)); CK(hipMemset(st, 0, 64));
    CK(hipMalloc(&ws, 1 << 20)); CK(hipMalloc(&bonus, n * 4)); CK(hipMalloc(&pp, 4096 * 4)); CK(hipMalloc(&rp, 4096 * 4));
    std::vector<float> h(cells);
    srand(1); for (auto &v : h) v = (rand() % 10 == 0) ? 1.0f : 0.0f;
    CK(hipMemcpy(u[0], h.data(), cells * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(act, h.data(), cells * 4, hipMemcpyHostToDevice));
    std::vector<float> w(4096); for (auto &v : w) v = (rand() % 2000 - 1000) / 3000.0f;
    CK(hipMemcpy(pp, w.data(), 4096 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(rp, w.data(), 4096 * 4, hipMemcpyHostToDevice));
    hipStream_t sa, sb; CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
    hipEvent_t e0, e1, ev[64]; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const uint32_t life = (1u << 12) | 0xC; const int reps = 200;
    for (int mode = 0; mode < 3; ++mode) {
        for (int pass = 0; pass < 2; ++pass) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, sa));
            for (int t = 0; t < reps; ++t) {
                float *in = u[t % 3], *out = u[(t + 1) % 3];
                carle_ca_step_f32(in, act, out, n, 64, 64, 64, 64, life, st, 0, sa);
                if (mode == 1) carle_rnd2d_bonus_f32(out, n, 64, 64, pp, rp, ws, bonus, nullptr, nullptr, 3, sa);
                if (mode == 2) {
                    CK(hipEventRecord(ev[t % 64], sa)); CK(hipStreamWaitEvent(sb, ev[t % 64], 0));
                    carle_rnd2d_bonus_f32(out, n, 64, 64, pp, rp, ws, bonus, nullptr, nullptr, t ? 3 : 1, sb);
                }
            }
            if (mode == 2) { CK(hipEventRecord(ev[0], sb)); CK(hipStreamWaitEvent(sa, ev[0], 0)); }
            CK(hipEventRecord(e1, sa)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (pass) printf("%s: %.2f us per step\n", mode == 0 ? "CA only" : mode == 1 ? "CA + RND, one stream" : "CA + RND, two streams", ms * 1000.f / reps);
        }
    }
    return 0;
}

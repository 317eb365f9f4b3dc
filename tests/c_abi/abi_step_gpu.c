/* Pure C consumer of the drop-in boundary ON THE GPU: no Python, no torch.  Allocates with the HIP runtime's C API,
 * steps a batch of 64x64 Life grids through carle_env_step_f32 (toggle + generation + epilogue) and compares every
 * cell with oracle/ca_oracle.c (linked in: the checker).  Also one bit-packed step (pack -> carle_ca_step_b1 ->
 * unpack).  Built with gcc and run by tests/test_gpu_ca.py::test_c_consumer_steps_on_the_gpu. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "carle_hip.h"

int oracle_env_step(const float *u_in, const float *action, float *u_out, float *scratch, int64_t n, int h, int w,
                    int ah, int aw, uint32_t rule, int32_t *flags_out);

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at line %d\n", (int)e_, __LINE__); return 2; } } while (0)
#define CHECK(cond) do { if (!(cond)) { printf("FAILED: %s (line %d): %s\n", #cond, __LINE__, carle_last_error()); return 1; } } while (0)

int main(void) {
    const int64_t n = 37;
    const int h = 64, w = 64, steps = 5;
    const size_t cells = (size_t)n * h * w;
    const uint32_t life = (1u << (9 + 3)) | (1u << 2) | (1u << 3);       /* B3/S23: bit 9 + k = birth at k, bit k = survival */
    float *hu = malloc(cells * 4), *ha = malloc(cells * 4), *href = malloc(cells * 4), *hscr = malloc(cells * 4),
          *hout = malloc(cells * 4);
    unsigned s = 12345u;
    for (size_t i = 0; i < cells; ++i) { s = s * 1664525u + 1013904223u; hu[i] = (s >> 24) < 96 ? 1.0f : 0.0f; }
    float *du, *dv, *da, *dr;
    int32_t *dst;
    uint64_t *dp, *dq;
    HIPCHK(hipMalloc((void **)&du, cells * 4));
    HIPCHK(hipMalloc((void **)&dv, cells * 4));
    HIPCHK(hipMalloc((void **)&da, cells * 4));
    HIPCHK(hipMalloc((void **)&dr, n * 4));
    HIPCHK(hipMalloc((void **)&dst, CARLE_ST_WORDS * 4));
    HIPCHK(hipMalloc((void **)&dp, cells / 8));
    HIPCHK(hipMalloc((void **)&dq, cells / 8));
    HIPCHK(hipMemset(dst, 0, CARLE_ST_WORDS * 4));
    HIPCHK(hipMemcpy(du, hu, cells * 4, hipMemcpyHostToDevice));
    for (int t = 0; t < steps; ++t) {
        for (size_t i = 0; i < cells; ++i) { s = s * 1664525u + 1013904223u; ha[i] = (s >> 24) < 26 ? 1.0f : 0.0f; }
        HIPCHK(hipMemcpy(da, ha, cells * 4, hipMemcpyHostToDevice));
        CHECK(carle_env_step_f32(du, da, dv, dr, n, h, w, 64, 64, life, dst, t & 1, CARLE_STEP_PUBLISH_FLAGS, NULL) == 0);
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(hout, dv, cells * 4, hipMemcpyDeviceToHost));
        CHECK(oracle_env_step(hu, ha, href, hscr, n, h, w, 64, 64, life, NULL) == 0);
        CHECK(memcmp(hout, href, cells * 4) == 0);
        memcpy(hu, href, cells * 4);
        float *tmp = du; du = dv; dv = tmp;
    }
    /* counters of the status block: step_number == steps */
    int32_t hst[CARLE_ST_WORDS];
    HIPCHK(hipMemcpy(hst, dst, sizeof hst, hipMemcpyDeviceToHost));
    CHECK(hst[CARLE_ST_STEP_NUMBER] == steps && hst[CARLE_ST_STEPS_SINCE_ACTION] == 0);
    /* the master reset (env.py:208-216): an all-ones action zeroes the batch and the step counter */
    for (size_t i = 0; i < cells; ++i) ha[i] = 1.0f;
    HIPCHK(hipMemcpy(da, ha, cells * 4, hipMemcpyHostToDevice));
    CHECK(carle_env_step_f32(du, da, dv, dr, n, h, w, 64, 64, life, dst, steps & 1, CARLE_STEP_PUBLISH_FLAGS, NULL) == 0);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(hout, dv, cells * 4, hipMemcpyDeviceToHost));
    CHECK(oracle_env_step(hu, ha, href, hscr, n, h, w, 64, 64, life, NULL) == 1);
    CHECK(memcmp(hout, href, cells * 4) == 0);
    HIPCHK(hipMemcpy(hst, dst, sizeof hst, hipMemcpyDeviceToHost));
    CHECK(hst[CARLE_ST_STEP_NUMBER] == 0 && hst[CARLE_ST_LAST_RESET] == 1 && hst[CARLE_ST_RESETS] == 1);
    /* the same five steps + the reset as ONE launch each (CARLE_STEP_DEFERRED): grids final at once, counters after
     * carle_env_settle (or the next call on the block) */
    HIPCHK(hipMemset(dst, 0, CARLE_ST_WORDS * 4));
    {
        unsigned s2 = 777u;
        for (size_t i = 0; i < cells; ++i) { s2 = s2 * 1664525u + 1013904223u; hu[i] = (s2 >> 24) < 96 ? 1.0f : 0.0f; }
        HIPCHK(hipMemcpy(du, hu, cells * 4, hipMemcpyHostToDevice));
        for (int t = 0; t < steps + 1; ++t) {
            for (size_t i = 0; i < cells; ++i) { s2 = s2 * 1664525u + 1013904223u; ha[i] = (t == steps || (s2 >> 24) < 26) ? 1.0f : 0.0f; }
            if (t == 2) memset(ha, 0, cells * 4);                       /* an all-zero action: steps_since_action */
            HIPCHK(hipMemcpy(da, ha, cells * 4, hipMemcpyHostToDevice));
            CHECK(carle_env_step_f32(du, da, dv, dr, n, h, w, 64, 64, life, dst, t & 1,
                                     CARLE_STEP_PUBLISH_FLAGS | CARLE_STEP_DEFERRED, NULL) == 0);
            HIPCHK(hipDeviceSynchronize());
            HIPCHK(hipMemcpy(hout, dv, cells * 4, hipMemcpyDeviceToHost));
            CHECK(oracle_env_step(hu, ha, href, hscr, n, h, w, 64, 64, life, NULL) == (t == steps));
            CHECK(memcmp(hout, href, cells * 4) == 0);
            memcpy(hu, href, cells * 4);
            float *tmp = du; du = dv; dv = tmp;
            if (t == steps - 1) {
                CHECK(carle_env_settle(dst, NULL) == 0);
                HIPCHK(hipMemcpy(hst, dst, sizeof hst, hipMemcpyDeviceToHost));
                CHECK(hst[CARLE_ST_STEP_NUMBER] == steps && hst[CARLE_ST_STEPS_SINCE_ACTION] == 1 && hst[CARLE_ST_PENDING] == 0);
            }
        }
        HIPCHK(hipMemcpy(hst, dst, sizeof hst, hipMemcpyDeviceToHost));
        CHECK(hst[CARLE_ST_PENDING] != 0);                              /* the reset step is still pending ... */
        CHECK(carle_env_settle(dst, NULL) == 0);
        HIPCHK(hipMemcpy(hst, dst, sizeof hst, hipMemcpyDeviceToHost));  /* ... and applied here */
        CHECK(hst[CARLE_ST_STEP_NUMBER] == 0 && hst[CARLE_ST_LAST_RESET] == 1 && hst[CARLE_ST_RESETS] == 1 && hst[CARLE_ST_PENDING] == 0);
        CHECK(hst[CARLE_ST_SCAN_TICKET(0)] == 0 && hst[CARLE_ST_SCAN_TICKET(1)] == 0 && hst[CARLE_ST_SCAN_DONE(0)] == 0 && hst[CARLE_ST_SCAN_DONE(1)] == 0);
    }
    /* one generation on bit-packed state, no action */
    CHECK(carle_pack_f32_b1(du, dp, n, h, w, NULL, 0, NULL) == 0);
    CHECK(carle_ca_step_b1(dp, NULL, dq, n, h, w, 0, 0, life, NULL, 0, NULL) == 0);
    CHECK(carle_unpack_b1_f32(dq, dv, n, h, w, NULL) == 0);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(hout, dv, cells * 4, hipMemcpyDeviceToHost));
    CHECK(oracle_env_step(hu, NULL, href, hscr, n, h, w, 0, 0, life, NULL) == 0);
    CHECK(memcmp(hout, href, cells * 4) == 0);
    printf("abi gpu ok: %d steps of %lld grids bit-identical to the C oracle, fp32 and bit-packed\n", steps, (long long)n);
    return 0;
}

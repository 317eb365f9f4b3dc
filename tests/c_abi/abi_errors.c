/* C consumer of include/carle_hip.h: proves the header is plain C and that the argument-error
 * paths of libcarle_hip.so answer without touching a GPU.  Built and run by
 * tests/test_abi_cpu.py::test_header_is_plain_c_and_links. */
#include <stdio.h>
#include <string.h>
#include "carle_hip.h"

#define CHECK(cond) do { if (!(cond)) { printf("FAILED: %s (line %d): %s\n", #cond, __LINE__, carle_last_error()); return 1; } } while (0)

int main(void) {
    static float grid[64] __attribute__((aligned(16)));
    static int32_t status[CARLE_ST_WORDS] __attribute__((aligned(16)));
    CHECK(carle_abi_version() == CARLE_ABI_VERSION);
    CHECK(strcmp(carle_build_arch(), "gfx950") == 0);
    CHECK(carle_ca_step_f32(NULL, NULL, NULL, 1, 64, 64, 0, 0, 0, NULL, 0, NULL) == CARLE_E_NULL);
    CHECK(strstr(carle_last_error(), "NULL") != NULL);
    CHECK(carle_ca_step_f32(grid, NULL, grid + 32, 1, 64, 32, 0, 0, 0, NULL, 0, NULL) == CARLE_E_DIMS);
    CHECK(carle_ca_step_f32(grid, NULL, grid + 32, 1, 64, 64, 0, 0, 1u << 18, NULL, 0, NULL) == CARLE_E_RULE);
    CHECK(carle_ca_step_u8((const uint8_t *)grid, grid, 9, (uint8_t *)(grid + 32), 1, 64, 64, 64, 64, 0, NULL, 0, NULL) == CARLE_E_FORMAT);
    CHECK(carle_env_step_f32(grid, NULL, grid + 32, NULL, 1, 64, 64, 0, 0, 0, NULL, 0,
                             CARLE_STEP_PUBLISH_FLAGS | CARLE_STEP_MERGE_MOMENTS, NULL) == CARLE_E_NULL);
    CHECK(carle_ca_multistep_b1((const uint64_t *)grid, (uint64_t *)grid, 0, 64, 64, 0, 4, status, NULL) == 0);
    CHECK(carle_rnd2d_workspace_floats(16, 256, 256) > 0 && carle_ae2d_workspace_floats(16, 256, 256) > 0);
    CHECK(CARLE_ST_SCAN_DONE(1) < CARLE_ST_WORDS && CARLE_ST_WORDS * 4 == 128);
    CHECK(carle_env_settle(NULL, NULL) == CARLE_E_NULL);
    CHECK(carle_env_settle(status + 1, NULL) == CARLE_E_ALIGN);
    carle_reload_knobs();
    printf("abi ok\n");
    return 0;
}

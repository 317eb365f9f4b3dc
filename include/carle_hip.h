/*
 * carle_hip.h -- C ABI of libcarle_hip.so, the MI355X (gfx950) hot path of CARLE.
 *
 * The reference (riveSunder/carle) is pure Python/PyTorch and has no FFI of its own; the
 * interface each entry point replaces is therefore a Python method of the reference, cited as
 * file:line next to every declaration.  INTEGRATION.md shows the ctypes binding a maintainer
 * of the reference would add.
 *
 * Conventions (all entry points)
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer on the current HIP
 *     device, owned by the caller, contiguous, 16-byte aligned.  The library allocates no
 *     persistent device memory and keeps no global mutable state (thread-local error text only).
 *   - asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream); no host
 *     synchronisation inside; safe from several host threads on different streams.
 *   - return 0 on success; a POSITIVE value is an argument error (CARLE_E_*); a NEGATIVE value
 *     is -(hipError_t).  carle_last_error() returns the message of the calling thread.
 *   - grids are (N,1,H,W) row-major; the action window (N,1,AH,AW) is centred at row offset
 *     (H-AH)/2 and column offset (W-AW)/2 (carle/env.py:119-132).
 *   - `rule` = 18-bit mask: bit k (k=0..8) = a live cell with k live neighbours survives,
 *     bit 9+k = a dead cell with k live neighbours is born (carle/env.py:221-227).
 */
#ifndef CARLE_HIP_H
#define CARLE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CARLE_ABI_VERSION 2

/* argument error codes (positive return values) */
#define CARLE_E_NULL      1   /* required pointer is NULL */
#define CARLE_E_DIMS      2   /* unsupported / inconsistent N,H,W,AH,AW */
#define CARLE_E_RULE      3   /* rule mask has bits above bit 17 */
#define CARLE_E_ALIGN     4   /* pointer not 16-byte aligned */
#define CARLE_E_ALIAS     5   /* u_in and u_out overlap */
#define CARLE_E_FORMAT    6   /* unknown grid format */

/* grid storage formats */
#define CARLE_FMT_F32 0   /* float32 per cell, any non-zero = alive (the reference's format)  */
#define CARLE_FMT_U8  1   /* uint8 per cell, any non-zero = alive                              */
#define CARLE_FMT_B1  2   /* 1 bit per cell: W/64 little-endian uint64 words per row, bit j of
                             word i = column 64*i+j; requires W % 64 == 0                     */

/* Layout of the per-environment device status block (int32[CARLE_ST_WORDS] = 128 bytes, 16-byte aligned,
 * zero-initialised by the caller once).  Two flag slots alternate between consecutive steps (slot = step
 * parity): consecutive steps of one environment MUST use alternating slots. */
#define CARLE_ST_ANY_NONZERO(slot)   (0 + (slot))   /* some action element != 0   (env.py:191) */
#define CARLE_ST_NOT_ALL_ONES(slot)  (2 + (slot))   /* some action element != 1.0 (env.py:208) */
#define CARLE_ST_LAST_RESET          4              /* 1 if the last finished step was a master reset */
#define CARLE_ST_PENDING             5              /* bookkeeping of a CARLE_STEP_DEFERRED step not yet applied to the
                                                       counters: bit 0 = pending, bit 1 = the step had an action,
                                                       bit 2 = its flag slot.  Applied by the next step / finish /
                                                       run / multistep call on this block, or by carle_env_settle */
#define CARLE_ST_STEP_NUMBER         6              /* int64 at int32 index 6..7  (env.py:142,230) */
#define CARLE_ST_STEPS_SINCE_ACTION  8              /* int64 at int32 index 8..9  (env.py:145,200) */
#define CARLE_ST_RESETS              10             /* int64 at int32 index 10..11: number of master resets */
#define CARLE_ST_MOMENTS_OUT         12             /* device pointer (64 bit) at index 12..13: double[3], see below */
#define CARLE_ST_MOMENTS_ACC         14             /* device pointer (64 bit) at index 14..15: double[3], see below */
#define CARLE_ST_SCAN_TICKET(slot)   (16 + (slot))  /* transient, in-kernel: next action grid to verify ...         */
#define CARLE_ST_SCAN_DONE(slot)     (18 + (slot))  /* ... and number verified all-ones, bit 31 = one was not (see
                                                       CARLE_STEP_DEFERRED); both are zero between steps              */
#define CARLE_ST_MOM_PARTIALS        20             /* device pointer (64 bit) at 20..21: per-workgroup (sum, sum of squares) fp64
                                                       pairs of the rewards of a CARLE_STEP_DEFERRED step whose reward
                                                       moments are still to be merged (carle_env_step_f32_rnd) ...     */
#define CARLE_ST_MOM_COUNT           22             /* ... number of pairs (0 = nothing pending) ...                   */
#define CARLE_ST_MOM_N               23             /* ... and number of rewards they cover                            */
#define CARLE_ST_TRACE_CAP            24             /* > 1: MOMENTS_OUT points at TRACE_CAP consecutive double[3] rows, and ...  */
#define CARLE_ST_TRACE_IDX            25             /* ... the step that writes its moments takes row TRACE_IDX % TRACE_CAP and
                                                       increments TRACE_IDX: the per-step reward totals of the harness loops
                                                       (train_mcl.py:68-69, eval.py:69-70) without a launch per step.  0 / 1: the one
                                                       triple at MOMENTS_OUT, as before.  A run of k steps in one launch writes k rows */
#define CARLE_ST_WORDS               32             /* words 26..31 reserved, zero */

/* `publish_flags` argument of the carle_env_step_* entry points (bit set) */
#define CARLE_STEP_PUBLISH_FLAGS     1   /* the step kernel writes the action predicates into the status block */
#define CARLE_STEP_MERGE_MOMENTS     2   /* the reward of this step is final: the epilogue also reduces it to the
                                            moment triple (count, mean, M2; fp64) at MOMENTS_OUT and Chan-merges it
                                            into the running triple at MOMENTS_ACC -- the per-step reward statistics
                                            of carle_batch_moments_f32 without a third launch.  Ignored while the
                                            two pointer slots of the status block are zero. */

#define CARLE_STEP_DEFERRED          4   /* ONE kernel launch per step where the geometry allows it (64x64 fp32 /
                                            uint8 grids, AW % 8 == 0, more than 16 instances, action tensor or
                                            none): the step kernel applies the master reset itself, writes the zero
                                            reward and (MERGE_MOMENTS) the moments of the all-zero reward, and leaves
                                            the counter update of env.py:191-230 PENDING in the status block
                                            (CARLE_ST_PENDING).  Grids, reward and moments are final when the kernel
                                            ends; step_number / steps_since_action / resets / LAST_RESET are valid
                                            after the next call on the block or after carle_env_settle().  Other
                                            geometries ignore the bit (step kernel + carle_ca_finish).
                                            How the kernel decides the WHOLE-BATCH reset without a second launch: a
                                            wavefront that finds its own grid's action all ones (never the case for
                                            an agent's toggles) verifies the other grids' actions together with the
                                            other such wavefronts -- tickets from SCAN_TICKET, results in SCAN_DONE,
                                            agent-scope atomics, no wavefront ever waits for one that has not
                                            started -- and zeroes its grid iff all N were verified. */

int         carle_abi_version(void);
const char *carle_last_error(void);
/* "gfx950" etc.: the offload architectures compiled into the library */
const char *carle_build_arch(void);
/* Experiment knobs (CARLE_* environment variables, DESIGN.md "Knobs") are read once, at first use; this re-reads
 * them.  For tests and A/B tools; not safe while another thread is inside a carle_* call. */
void        carle_reload_knobs(void);

/*
 * One CA generation, fp32 drop-in format.  Replaces CARLE.apply_action + the update branch of
 * CARLE.step (carle/env.py:150-182, 219-229): u_out = rule(count3x3_torus(u_in XOR pad(action))).
 *   u_in, u_out  N*H*W floats; must not alias.  Any non-zero u_in value counts as alive; u_out
 *                holds exactly 0.0f / 1.0f.
 *   action       N*AH*AW floats or NULL (= no toggles; AH/AW ignored).  Any non-zero toggles.
 *   status       NULL, or the status block: the kernel sets ANY_NONZERO(slot) if some action
 *                element is non-zero and NOT_ALL_ONES(slot) if some element differs from 1.0f.
 *                The generation is ALWAYS computed; the master reset (env.py:208-216) is applied
 *                afterwards by carle_ca_finish().
 * Supported: square even grids H == W >= 2, AH == AW <= H with (H-AH) even (the reference's
 * consistent domain, env.py:119-132).  64x64 grids with AW % 8 == 0 (and 256/512 grids, see
 * DESIGN.md) take the wavefront-per-strip fast path; everything else a generic kernel.
 */
int carle_ca_step_f32(const float *u_in, const float *action, float *u_out,
                      int64_t n, int h, int w, int ah, int aw, uint32_t rule,
                      int32_t *status, int slot, void *stream);

/*
 * One CA generation on bit-packed grids (CARLE_FMT_B1): 0.25 B of traffic per cell update.
 * Same semantics as carle_ca_step_f32 (carle/env.py:150-182, 219-229) on 1-bit cells.
 *   u_in, u_out  N*H*(W/64) uint64 words; must not alias.  H == W in {64, 128, 256, 512}.
 *   action       N*AH*(AW/64) words or NULL; AW % 64 == 0; a set bit toggles.  The status flags
 *                are derived from the bits (any bit set / every bit set): exact for {0,1}-valued
 *                actions.  A caller holding an fp32 action packs it with carle_pack_f32_b1, which
 *                publishes the exact fp32 predicates; then pass status = NULL here.
 */
int carle_ca_step_b1(const uint64_t *u_in, const uint64_t *action, uint64_t *u_out,
                     int64_t n, int h, int w, int ah, int aw, uint32_t rule,
                     int32_t *status, int slot, void *stream);

/*
 * One CA generation on uint8 grids (CARLE_FMT_U8): the fp32 layout at 1 B per cell, 2 B of traffic
 * per cell update.  Same semantics as carle_ca_step_f32 (carle/env.py:150-182, 219-229).
 *   u_in, u_out  N*H*W bytes; must not alias.  Any non-zero byte is alive; u_out holds 0 / 1.
 *   action       NULL, or N*AH*AW elements of `action_fmt`: CARLE_FMT_F32 (the tensor a reference
 *                agent emits; predicates as in carle_ca_step_f32) or CARLE_FMT_U8 (non-zero byte
 *                toggles; NOT_ALL_ONES is set by any byte != 1).
 * Same geometry domain and fast paths as carle_ca_step_f32 (64x64 with AW % 8 == 0; 256/512).
 */
int carle_ca_step_u8(const uint8_t *u_in, const void *action, int action_fmt, uint8_t *u_out,
                     int64_t n, int h, int w, int ah, int aw, uint32_t rule,
                     int32_t *status, int slot, void *stream);
/* stand-alone CARLE.apply_action on uint8 grids (carle/env.py:150-182); u_out may alias u_in */
int carle_ca_toggle_u8(const uint8_t *u_in, const void *action, int action_fmt, uint8_t *u_out,
                       int64_t n, int h, int w, int ah, int aw, void *stream);
/* uint8 <-> bit-packed (W % 64 == 0); status (nullable) receives the predicates of the uint8 source */
int carle_pack_u8_b1(const uint8_t *src, uint64_t *dst, int64_t n, int h, int w,
                     int32_t *status, int slot, void *stream);
int carle_unpack_b1_u8(const uint64_t *src, uint8_t *dst, int64_t n, int h, int w, void *stream);
/*
 * The two whole-batch predicates CARLE.step evaluates on the action AS PASSED
 * (`torch.sum(action)` truthiness, carle/env.py:191; `torch.mean(action) == 1.0`, env.py:208) for
 * `count` fp32 elements: sets ANY_NONZERO(slot) / NOT_ALL_ONES(slot), never clears them.  The
 * step kernels publish the same flags themselves; this entry point serves callers whose action
 * does not reach a step kernel unchanged (oversize actions that are centre-cropped, env.py:164-169).
 */
int carle_action_flags_f32(const float *action, int64_t count, int32_t *status, int slot,
                           void *stream);

/*
 * k generations without actions on bit-packed grids in ONE launch: the result of k consecutive
 * CARLE.step calls with an all-zero action (carle/env.py:188-242; the toggle is the identity and
 * the master reset cannot fire).  One workgroup holds a grid in registers for all k generations:
 * 0.25 B of HBM traffic per cell per k generations.  u_out may be u_in (in place).
 *   status (nullable): step_number += k, steps_since_action += k (env.py:199-200, 230).
 * H == W in {64, 128, 256, 512}; k >= 0 (k == 0 copies).
 */
int carle_ca_multistep_b1(const uint64_t *u_in, uint64_t *u_out, int64_t n, int h, int w,
                          uint32_t rule, int k, int32_t *status, void *stream);

/* fp32 (N,1,H,W) -> bit-packed, bit = (value != 0) (the truthiness of env.py:182); W % 64 == 0.
 * status (nullable): publishes ANY_NONZERO / NOT_ALL_ONES of the fp32 source into `slot`. */
int carle_pack_f32_b1(const float *src, uint64_t *dst, int64_t n, int h, int w,
                      int32_t *status, int slot, void *stream);
/* bit-packed -> fp32 0.0f / 1.0f (materialises the reference-format observation). */
int carle_unpack_b1_f32(const uint64_t *src, float *dst, int64_t n, int h, int w, void *stream);

/*
 * Toggle only: u_out = (u_in != 0) XOR pad(action != 0), as 0.0f/1.0f.  Replaces a stand-alone
 * call of CARLE.apply_action (carle/env.py:150-182); CARLE.step uses the fused kernel above.
 * u_out may alias u_in (elementwise).
 */
int carle_ca_toggle_f32(const float *u_in, const float *action, float *u_out,
                        int64_t n, int h, int w, int ah, int aw, void *stream);

/*
 * Completes a step started by carle_ca_step_*: applies the whole-batch master reset of
 * CARLE.step (carle/env.py:208-216) and maintains the counters (env.py:142-145,200,230).
 *   if has_action and NOT_ALL_ONES(slot) == 0  ->  u_out := 0 (all instances), step_number := 0,
 *                                                   steps_since_action := 0, resets += 1
 *   else                                        ->  step_number += 1
 *   if ANY_NONZERO(slot) == 0 (or no action)    ->  steps_since_action += 1   (before the reset)
 *   reward[0..n) := 0.0f  (the fresh zero reward of env.py:238; may be NULL)
 *   flag slot (slot ^ 1) is cleared for the next step.
 * `u_out_bytes` is the size of the output grid buffer in bytes (any format).
 */
int carle_ca_finish(void *u_out, int64_t u_out_bytes, float *reward, int64_t n,
                    int32_t *status, int slot, int has_action, void *stream);

/*
 * Whole CARLE.step (carle/env.py:188-242) in one call: carle_ca_step_f32 followed by
 * carle_ca_finish on the same stream.  publish_flags = 0 keeps the step kernel from writing the
 * action flags (the caller has already stored them in the status block, e.g. for an oversize
 * action whose predicates are defined on the uncropped tensor, env.py:164-169,191,208).
 */
int carle_env_step_f32(const float *u_in, const float *action, float *u_out, float *reward,
                       int64_t n, int h, int w, int ah, int aw, uint32_t rule,
                       int32_t *status, int slot, int publish_flags, void *stream);
/* carle_env_step_f32 that ALSO leaves a bit-packed copy of the new state (H rows of W/64 words per grid: the layout of
 * carle_pack_f32_b1) in bits_out where the launch produces it for free -- the one-launch step (CARLE_STEP_DEFERRED) of
 * 64- / 256- / 512-column grids.  The reward-net kernels (carle_rnd2d_bonus_b1, carle_ae2d_train_b1, carle_rnd_ae_bonus with
 * CARLE_FMT_B1, ...) read 512 bytes per grid / tile from it instead of 16 / 22 KB of fp32 cells (carle/mcl.py:398-407 on the
 * observation of carle/env.py:229).  *bits_written (host int, nullable) = 1 if the copy was written, 0 if bits_out was not touched. */
int carle_env_step_f32_bits(const float *u_in, const float *action, float *u_out, float *reward, int64_t n, int h, int w,
                            int ah, int aw, uint32_t rule, int32_t *status, int slot, int step_flags, uint64_t *bits_out,
                            int *bits_written, void *stream);
/* CARLE.step (CARLE_STEP_DEFERRED semantics where the geometry has a one-launch step, step kernel + epilogue otherwise) of
 * n >= 1 grids of 64x64, 256x256 or 512x512 fp32 cells whose INPUT state is read from its bit-packed copy bits_in (H rows of
 * W/64 words per grid, as a previous carle_env_step_f32_bits / carle_env_step_f32_packed / carle_pack_f32_b1 left it) instead
 * of the fp32 cells -- the same bits in 1/32 of the bytes (64x64: 512 B instead of 16 KiB per grid).  The fp32
 * observation u_out (carle/env.py:229) and its packed copy bits_out are written as by carle_env_step_f32_bits; grids, counters,
 * reward and flags are those of carle_env_step_f32 on the fp32 state.  The caller vouches that bits_in IS the current state
 * (carle_amd.env.CARLE.step: the very tensor the last step returned, not modified since); bits_in and bits_out must not
 * overlap; 16-byte aligned for 256 / 512 columns.  action: fp32 (n,1,ah,aw) window (64x64 grids: aw a multiple of 8; wider
 * grids: aw a multiple of 4 with (w - aw) a multiple of 8), or NULL. */
int carle_env_step_f32_packed(const uint64_t *bits_in, const float *action, float *u_out, float *reward, int64_t n, int h,
                              int w, int ah, int aw, uint32_t rule, int32_t *status, int slot, int step_flags,
                              uint64_t *bits_out, void *stream);

/* Applies a pending CARLE_STEP_DEFERRED step to the counters of the status block (one tiny kernel; a no-op on the
 * device when nothing is pending).  Call before copying the counters to the host (env.py:142-145,199-200,230). */
int carle_env_settle(int32_t *status, void *stream);

/*
 * k consecutive CARLE.step calls (carle/env.py:188-242) in one library call -- the inner loop of
 * carle/train_mcl.py:64-69 / evaluation/eval.py:63-70 when the actions do not depend on the
 * observations that the loop produces (a device-side RandomAgent, carle/agents.py:35-42, or a
 * recorded action sequence).
 *   u_in                   N*H*W floats, the state on entry; READ ONLY (it may be a tensor the caller has handed
 *                          out as an observation: the reference returns a fresh tensor per step, env.py:229)
 *   buf_a, buf_b           N*H*W floats each, caller-owned scratch, no overlap with each other or u_in: step t
 *                          writes buf_a for even t and buf_b for odd t, so the final state is in buf_a for odd k
 *                          and in buf_b for even k (k == 0: nothing is written; k == 1: buf_b may be NULL)
 *   actions != NULL        k contiguous (N,1,AH,AW) fp32 tensors; step t uses the t-th
 *   actions == NULL, p>=0  Bernoulli(p) toggles drawn in the step kernel, step t with Philox offset
 *                          `offset + t` (= k calls of carle_env_step_f32_rand)
 *   actions == NULL, p<0   no action (AH/AW ignored)
 *   reward                 N floats: the reward of the LAST step (zero for the bare environment)
 *   slot0                  flag slot of the first step; step t uses (slot0 + t) & 1
 *   step_flags             CARLE_STEP_MERGE_MOMENTS, CARLE_STEP_DEFERRED or 0
 * Every step is complete (master reset, flags; counters: see CARLE_STEP_DEFERRED) before the next one starts,
 * exactly as k separate calls of carle_env_step_f32* on the same stream (grids, counters, reward moments and the
 * reward of the last step are identical; the contents of the buffer that does not hold the final state
 * and the transient action-flag words of the status block are unspecified).  64x64 grids: batches of up
 * to 16 run all k steps in ONE launch with the grids in registers, and without an action source any
 * batch does (no predicates, no master reset without an action); these launches write the final buffer only.
 */
int carle_env_run_f32(const float *u_in, float *buf_a, float *buf_b, const float *actions, float *reward,
                      int64_t n, int h, int w, int ah, int aw, uint32_t rule, float p, uint64_t seed,
                      uint64_t offset, int k, int32_t *status, int slot0, int step_flags, void *stream);

/* The same on bit-packed state: carle_ca_step_b1 + carle_ca_finish. */
int carle_env_step_b1(const uint64_t *u_in, const uint64_t *action, uint64_t *u_out, float *reward,
                      int64_t n, int h, int w, int ah, int aw, uint32_t rule,
                      int32_t *status, int slot, int publish_flags, void *stream);

/* The same on uint8 state: carle_ca_step_u8 + carle_ca_finish. */
int carle_env_step_u8(const uint8_t *u_in, const void *action, int action_fmt, uint8_t *u_out,
                      float *reward, int64_t n, int h, int w, int ah, int aw, uint32_t rule,
                      int32_t *status, int slot, int publish_flags, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Intrinsic-reward forward passes (eval mode: Dropout = identity), fp32, tolerance 1e-5 abs.
 * Parameter blobs are flat fp32 arrays in the state_dict order of the reference's
 * nn.Sequential modules (carle/mcl.py:262-276, 285-296, 452-467):
 *   RND predictor: conv1.w[4,1,3,3] conv1.b[4] conv2.w[1,4,3,3] conv2.b[1] lin.w[16,D] lin.b[16]
 *   RND random   : conv1.w[2,1,3,3] conv1.b[2] conv2.w[1,2,3,3] conv2.b[1] lin.w[16,D] lin.b[16]
 *   AE predictor : conv1.w[4,1,3,3] b[4] conv2.w[2,4,3,3] b[2] convT1.w[2,1,4,4] b[1]
 *                  convT2.w[1,1,4,4] b[1]                                    (164 floats)
 * with D = (H/8)*(W/8).  `workspace` is caller-owned scratch of at least
 * carle_*_workspace_floats(n,h,w) floats.  H == W, multiple of 8.
 * ------------------------------------------------------------------------------------------- */

/* reward_accum (nullable, N floats): additionally reward_accum[i] += reward_scale * bonus[i], the
 * wrapper's `reward += reward_scale * bonus` (carle/mcl.py:403-405) fused into the last kernel. */

/* RND2D.get_bonus (carle/mcl.py:319-330,343-352): bonus[i] = mean_k (random(obs_i)_k -
 * predictor(obs_i)_k)^2.  pred_out / target_out: optional (N,16) network outputs, may be NULL. */
int64_t carle_rnd2d_workspace_floats(int64_t n, int h, int w);
/* flags: CARLE_NET_OBS_BINARY = the caller guarantees every obs value is exactly 0.0f or 1.0f
 * (true for anything CARLE.step returned); enables the fused 64x64 kernel whose first conv is a
 * table lookup on the binary image.  Without the flag obs values are used as fp32 numbers. */
#define CARLE_NET_OBS_BINARY 1
/* CARLE_NET_TABLES_VALID: `workspace` still holds the lookup tables a previous fused call built
 * from the SAME parameter values (the caller tracks parameter updates); skips the table build.
 * Binary observations with H == W a multiple of 64 above 64 (tiled kernels): ONE launch -- the workgroup that delivers a
 * grid's last tile finishes the grid -- and CARLE_NET_TABLES_VALID additionally promises that the previous call on the
 * workspace had the same (n, h, w) (the workspace keeps per-grid arrival counters behind the tables).
 */
#define CARLE_NET_TABLES_VALID 2
int carle_rnd2d_bonus_f32(const float *obs, int64_t n, int h, int w,
                          const float *pred_params, const float *rand_params,
                          float *workspace, float *bonus, float *pred_out, float *target_out,
                          float *reward_accum, float reward_scale, int flags, void *stream);
/* the same on bit-packed observations (CARLE_FMT_B1), 64x64 grids */
int carle_rnd2d_bonus_b1(const uint64_t *obs, int64_t n, int h, int w,
                         const float *pred_params, const float *rand_params,
                         float *workspace, float *bonus, float *pred_out, float *target_out,
                         float *reward_accum, float reward_scale, int flags, void *stream);

/* RND2D bonus AND the predictor's gradient in one pass: replaces the autograd
 * graph the reference keeps alive across the update window (carle/mcl.py:343-386).  Eval-mode
 * semantics.  grad[0..77 + 16 D + 16) (predictor blob order, D = (H/8)(W/8); 1117 floats at 64x64)
 * is ACCUMULATED:
 *   grad += d/dtheta [ grad_scale/2 * sum_i sum_k (target_ik - prediction_ik)^2 ],
 * so grad_scale = 2 / (16 * N * batch_size) reproduces sum_t mean_n mean_k(.)^2 / batch_size.
 *   64x64 binary observations (CARLE_NET_OBS_BINARY, or bit-packed): one fused forward+backward
 *     kernel; partial = scratch of carle_rnd2d_train_partial_floats(n) floats;
 *   any other H == W multiple of 8 (the reference's default 256x256) or non-binary fp32 observations:
 *     staged forward + staged backward kernels (pool / relu arg-max routing as ATen's max_pool2d
 *     backward, gathered convolution gradients, deterministic row reductions); partial = scratch of
 *     carle_rnd2d_train_scratch_floats(n, h, w) floats.
 * workspace: carle_rnd2d_workspace_floats(n, h, w) floats, as for the bonus. */
int64_t carle_rnd2d_train_partial_floats(int64_t n);
int64_t carle_rnd2d_train_scratch_floats(int64_t n, int h, int w);
int carle_rnd2d_train_f32(const float *obs, int64_t n, int h, int w, const float *pred_params,
                          const float *rand_params, float *workspace, float *partial, float *grad,
                          float *bonus, float *reward_accum, float reward_scale, float grad_scale,
                          int flags, void *stream);
int carle_rnd2d_train_b1(const uint64_t *obs, int64_t n, int h, int w, const float *pred_params,
                         const float *rand_params, float *workspace, float *partial, float *grad,
                         float *bonus, float *reward_accum, float reward_scale, float grad_scale,
                         int flags, void *stream);

/* The same for AE2D (carle/mcl.py:504-547): grad[0..164) accumulates
 * d/dtheta [ grad_scale/2 * sum_i sum_hw (obs_i - decoder(obs_i))^2 ]; grad_scale = 2/(H*W*N*batch_size).
 * Same two paths; the staged one needs partial = carle_ae2d_train_scratch_floats(n, h, w) floats. */
int64_t carle_ae2d_train_partial_floats(int64_t n);
int64_t carle_ae2d_train_scratch_floats(int64_t n, int h, int w);
int carle_ae2d_train_f32(const float *obs, int64_t n, int h, int w, const float *params,
                         float *workspace, float *partial, float *grad, float *bonus,
                         float *reward_accum, float reward_scale, float grad_scale, int flags,
                         void *stream);
int carle_ae2d_train_b1(const uint64_t *obs, int64_t n, int h, int w, const float *params,
                        float *workspace, float *partial, float *grad, float *bonus,
                        float *reward_accum, float reward_scale, float grad_scale, int flags,
                        void *stream);

/* optimizer.step() of the wrappers' Adam (RND2D / AE2D.update_predictor, carle/mcl.py:333-341, 498-502; Adam(lr=6e-2)
 * with torch's defaults, mcl.py:243,278-279) on the flat gradient the carle_*_train_* kernels accumulated -- one launch,
 * torch.optim.Adam's arithmetic operation by operation (no weight decay, no amsgrad).
 *   param_ptrs / param_counts  HOST arrays of n_params (<= CARLE_ADAM_MAX_PARAMS) device pointers / element counts:
 *                              the predictor's parameter tensors in state_dict order, updated in place
 *   blob (nullable)            flat copy of the parameters in the same order (the *_params argument of the bonus
 *                              kernels): receives the new values too
 *   grad, exp_avg, exp_avg_sq  flat fp32 vectors of sum(param_counts) elements (Adam's state lives with the caller);
 *                              zero_grad != 0 clears grad afterwards (optimizer.zero_grad(), mcl.py:339)
 *   step                       1 for the first update (torch increments its step counter before using it) */
#define CARLE_ADAM_MAX_PARAMS 8
int carle_adam_update_f32(float *const *param_ptrs, const int64_t *param_counts, int n_params, float *blob,
                          float *grad, float *exp_avg, float *exp_avg_sq, double lr, double beta1, double beta2,
                          double eps, int64_t step, int zero_grad, void *stream);

/* AE2D.get_bonus (carle/mcl.py:436-443,504-512): bonus[i] = mean_{h,w} (obs_i - decoder(obs_i))^2.
 * recon_out: optional (N,1,H,W) reconstruction, may be NULL. */
int64_t carle_ae2d_workspace_floats(int64_t n, int h, int w);
int carle_ae2d_bonus_f32(const float *obs, int64_t n, int h, int w, const float *params,
                         float *workspace, float *bonus, float *recon_out,
                         float *reward_accum, float reward_scale, int flags, void *stream);
/* the same on bit-packed observations (CARLE_FMT_B1), 64x64 grids */
int carle_ae2d_bonus_b1(const uint64_t *obs, int64_t n, int h, int w, const float *params,
                        float *workspace, float *bonus, float *recon_out,
                        float *reward_accum, float reward_scale, int flags, void *stream);
/* carle_ae2d_bonus_b1 for an AE2D that is the OUTERMOST wrapper of a 64x64 stack -- AE2D(RND2D(env)), carle/train_mcl.py:36-40 --
 * whose rewards feed the running reward statistics (CARLE_ST_MOMENTS_*): reward_accum is required, and the kernel that
 * finishes the step's rewards also leaves their (sum, sum of squares) per workgroup in moment_partials
 * (carle_ae2d_stats_partials_doubles(n) doubles, caller-owned, 16-byte aligned, alive until the next call on the status
 * block) and registers them in `status`: the next call on the block (or carle_env_settle) merges them, as for
 * carle_env_step_f32_rnd with CARLE_STEP_DEFERRED | MERGE_MOMENTS.  No stand-alone moments launch per step. */
int64_t carle_ae2d_stats_partials_doubles(int64_t n);
int carle_ae2d_bonus_b1_stats(const uint64_t *obs, int64_t n, int h, int w, const float *params, float *workspace,
                              float *bonus, float *reward_accum, float reward_scale, int flags, double *moment_partials,
                              int32_t *status, void *stream);

/* PredictionBonus.get_bonus (carle/mcl.py:581-605): the AE2D predictor applied to frame t-k (`obs`), squared error against
 * frame t (`target`, same shape): bonus[i] = mean_{h,w} (target_i - decoder(obs_i))^2.  Per-layer kernels: any H = W
 * multiple of 8, any fp32 values; workspace = carle_ae2d_workspace_floats(n, h, w) floats; recon_out optional.
 * carle_ae2d_predict_train_f32 additionally accumulates d/dtheta [ grad_scale/2 * sum_i sum_hw (target_i - decoder(obs_i))^2 ]
 * into grad[0..164) (PredictionBonus.get_bonus_accumulate / get_bonus_update, mcl.py:644-677; grad_scale as
 * carle_ae2d_train_f32); scratch = carle_ae2d_train_scratch_floats(n, h, w) floats. */
int carle_ae2d_predict_bonus_f32(const float *obs, const float *target, int64_t n, int h, int w, const float *params,
                                 float *workspace, float *bonus, float *recon_out, void *stream);
int carle_ae2d_predict_train_f32(const float *obs, const float *target, int64_t n, int h, int w, const float *params,
                                 float *workspace, float *scratch, float *grad, float *bonus, float grad_scale,
                                 void *stream);
/* the same on bit-packed frames (CARLE_FMT_B1; binary grids, H = W a multiple of 64 -- what CARLE.step returns): AE2D's fused
 * 64x64 / tiled kernels with the error term and the output layer's gradient taken against `target`; workspace, partial,
 * flags (CARLE_NET_TABLES_VALID) as carle_ae2d_bonus_b1 / carle_ae2d_train_b1 */
int carle_ae2d_predict_bonus_b1(const uint64_t *obs, const uint64_t *target, int64_t n, int h, int w, const float *params,
                                float *workspace, float *bonus, float *recon_out, int flags, void *stream);
int carle_ae2d_predict_train_b1(const uint64_t *obs, const uint64_t *target, int64_t n, int h, int w, const float *params,
                                float *workspace, float *partial, float *grad, float *bonus, float grad_scale, int flags,
                                void *stream);

/* RND2D.get_bonus and AE2D.get_bonus of the SAME observations (carle/mcl.py:343-352, 504-512) and both wrappers'
 * `reward += reward_scale * bonus` (mcl.py:403-405, RND2D's first: the stack AE2D(RND2D(env)) of train_mcl.py:34-38 and
 * eval.py:83-85) in ONE launch instead of four -- the tile workgroups of both nets run side by side, and the workgroup that
 * delivers the last tile of a grid forms both bonuses and the reward.  CARLE_NET_TABLES_VALID in rnd_flags additionally
 * promises that the previous call on rnd_workspace had the same (n, h, w): the workspace keeps per-grid arrival counters.  Binary observations with H == W a multiple of 64 above 64 (the reference's default
 * 256x256); obs_fmt CARLE_FMT_F32 (every value exactly 0.0f / 1.0f) or CARLE_FMT_B1.  Everything else -- parameter
 * blobs, workspaces (carle_*_workspace_floats), bonus vectors, CARLE_NET_TABLES_VALID per net -- as for
 * carle_rnd2d_bonus_* / carle_ae2d_bonus_*; results are bit-identical to calling those two. */
int carle_rnd_ae_bonus(const void *obs, int obs_fmt, int64_t n, int h, int w, const float *rnd_pred_params,
                       const float *rnd_rand_params, float *rnd_workspace, float *rnd_bonus, float rnd_scale,
                       int rnd_flags, const float *ae_params, float *ae_workspace, float *ae_bonus, float ae_scale,
                       int ae_flags, float *reward_accum, double *moment_partials, int32_t *status, void *stream);
/* moment_partials + status (both or neither; reward_accum required): the rewards are final in the finish kernel of these
 * calls, so that kernel leaves per-workgroup (sum, sum of squares) pairs in moment_partials
 * (carle_rnd_ae_partials_doubles(n, h, w) fp64 values) and registers them in the status block of the environment
 * (CARLE_ST_MOM_*): the next launch on that block -- the next environment step -- or carle_env_settle merges them into the
 * running reward moments (CARLE_ST_MOMENTS_*), as for the one-launch RND2D step.  The block's previous pairs must have
 * been merged (any environment step or carle_env_settle since). */
int64_t carle_rnd_ae_partials_doubles(int64_t n, int h, int w);

/* The same stack in TRAINING mode (carle/train_mcl.py:34-38, 64-69; mcl.py:368-386, 521-547): both wrappers' bonuses, both
 * `reward += reward_scale * bonus` and the accumulation of both predictors' gradients for one step -- what
 * carle_rnd2d_train_* followed by carle_ae2d_train_* do -- in FOUR launches instead of nine (at the reference's default
 * geometry a step is a chain of dependent dispatches of 4-7 us each, whatever their size; the workgroup that delivers the
 * last tile of a grid finishes the grid).  CARLE_NET_TABLES_VALID in rnd_flags additionally promises that the previous
 * call on rnd_workspace had the same (n, h, w): the workspace keeps per-grid arrival counters behind the tables.  Binary observations, H == W a
 * multiple of 64 above 64; rnd_scratch / ae_scratch: carle_rnd2d_train_scratch_floats / carle_ae2d_train_scratch_floats
 * floats; gradients are ADDED to rnd_grad / ae_grad (blob order); everything else as for the two per-net entry points,
 * results bit-identical to calling them one after the other. */
int carle_rnd_ae_train(const void *obs, int obs_fmt, int64_t n, int h, int w, const float *rnd_pred_params,
                       const float *rnd_rand_params, float *rnd_workspace, float *rnd_scratch, float *rnd_grad,
                       float *rnd_bonus, float rnd_scale, float rnd_grad_scale, int rnd_flags, const float *ae_params,
                       float *ae_workspace, float *ae_scratch, float *ae_grad, float *ae_bonus, float ae_scale,
                       float ae_grad_scale, int ae_flags, float *reward_accum, double *moment_partials, int32_t *status,
                       void *stream);

/* CARLE.step + the RND2D wrapper's bonus on the fresh observation (carle/env.py:188-242 +
 * carle/mcl.py:398-407) in one call.
 *   64x64 grids with an aligned action window (AW % 8 == 0 or no action): ONE kernel steps the
 *   automaton and evaluates the reward net on the state it still holds in registers; the epilogue
 *   writes reward = reward_scale * bonus (all-dead-grid bonus for every instance after a master
 *   reset).  Everything runs on stream_env; stream_reward (nullable) is ordered behind it through
 *   `event`, so reward consumers enqueued there see the result.
 *   other geometries: step and epilogue on stream_env, the bonus kernels (reward += reward_scale *
 *   bonus fused in) on stream_reward behind `event` (or on stream_env when stream_reward is NULL).
 * bonus: N + 1 floats (element N receives the bonus of the all-dead grid).
 * net_flags: CARLE_NET_TABLES_VALID as for carle_rnd2d_bonus_f32.
 * CARLE_STEP_DEFERRED in publish_flags (64x64, more than 16 instances): no epilogue kernel -- the fused kernel applies
 * the master reset per grid, writes the reward, and with CARLE_STEP_MERGE_MOMENTS leaves per-workgroup partial sums of
 * the rewards in `moment_partials` (caller-owned scratch of carle_rnd_step_partials_doubles(n) doubles, 16-byte
 * aligned, alive until the NEXT call on the status block or carle_env_settle has merged them; NULL without
 * MERGE_MOMENTS).  Counters as described at CARLE_STEP_DEFERRED. */
int64_t carle_rnd_step_partials_doubles(int64_t n);
/* carle_env_step_f32_rnd, and where that step is ONE launch (64x64, CARLE_STEP_DEFERRED, more than 16 instances) the kernel
 * also stores the new state once more as n x 64 row words (bit j of word r = cell (r, j); 512 bytes per grid instead of
 * 16 KiB) in bits_out (16-byte aligned, nullable) -- the observation as the 64x64 reward-net kernels of a wrapper stacked
 * on the RND2D read it (carle_ae2d_bonus_b1, carle_ae2d_train_b1: AE2D(RND2D(env)), carle/train_mcl.py:36-40).
 * *bits_written (nullable) := 1 iff the copy was written. */
int carle_env_step_f32_rnd_bits(const float *u_in, const float *action, float *u_out, float *reward, int64_t n, int h, int w,
                                int ah, int aw, uint32_t rule, int32_t *status, int slot, int step_flags,
                                const float *pred_params, const float *rand_params, float *workspace, float *bonus,
                                float reward_scale, int net_flags, double *moment_partials, void *event, void *stream_env,
                                void *stream_reward, uint64_t *bits_out, int *bits_written);
/* carle_env_step_f32_rnd_bits whose INPUT state is read from its bit-packed copy bits_in (n x 64 row words, as a previous
 * carle_env_step_f32_rnd_bits / _rnd_packed / carle_env_step_f32_bits / carle_pack_f32_b1 left it) instead of the fp32 cells: see
 * carle_env_step_f32_packed.  Always ONE launch (more than 16 grids of 64x64, action width a multiple of 8; CARLE_STEP_DEFERRED
 * semantics) or an error; bits_in / bits_out 16-byte aligned, not overlapping.  The caller vouches that bits_in IS the
 * current state. */
int carle_env_step_f32_rnd_packed(const uint64_t *bits_in, const float *action, float *u_out, float *reward, int64_t n, int h,
                                  int w, int ah, int aw, uint32_t rule, int32_t *status, int slot, int step_flags,
                                  const float *pred_params, const float *rand_params, float *workspace, float *bonus,
                                  float reward_scale, int net_flags, double *moment_partials, void *event, void *stream_env,
                                  void *stream_reward, uint64_t *bits_out);
int carle_env_step_f32_rnd(const float *u_in, const float *action, float *u_out, float *reward,
                           int64_t n, int h, int w, int ah, int aw, uint32_t rule,
                           int32_t *status, int slot, int publish_flags,
                           const float *pred_params, const float *rand_params, float *workspace,
                           float *bonus, float reward_scale, int net_flags, double *moment_partials,
                           void *event, void *stream_env, void *stream_reward);

/* The fused kernel of the 64x64 case alone, without the epilogue (as carle_ca_step_f32 is to
 * carle_env_step_f32): u_out = one generation of (u_in XOR action) (carle/env.py:150-182, 219-229),
 * bonus[i] = RND2D.get_bonus(u_out_i) (carle/mcl.py:343-352), bonus[N] = bonus of the all-dead grid.
 * 64x64 grids, AW % 8 == 0 (or action == NULL); status nullable. */
int carle_ca_step_f32_rnd(const float *u_in, const float *action, float *u_out, int64_t n, int h, int w,
                          int ah, int aw, uint32_t rule, int32_t *status, int slot,
                          const float *pred_params, const float *rand_params, float *workspace,
                          float *bonus, int net_flags, void *stream);

/* Orders `to_stream` after everything enqueued so far on `from_stream` (hipEventRecord on
 * from_stream + hipStreamWaitEvent on to_stream; `event` is a caller-owned hipEvent_t).  Used to
 * run the reward networks of step t on a second stream beside the CA kernel of step t+1. */
int carle_stream_fork(void *event, void *from_stream, void *to_stream);

/* ---------------------------------------------------------------------------------------------
 * Device-side agents.  The reference's RandomAgent draws its toggles with
 * torch.rand on the CPU and copies them to the device every step (carle/agents.py:35-42,
 * carle/mcl.py:400).  Philox4x32-10, counter = (quad index of the flattened action, offset),
 * key = seed: element 4q+j toggles iff (word_j >> 8) * 2^-24 <= p  (csrc/philox.hpp).
 * ------------------------------------------------------------------------------------------- */
/* out[0..count) := Bernoulli(p) as 0.0f / 1.0f; count % 4 == 0 */
int carle_bernoulli_f32(float *out, int64_t count, float p, uint64_t seed, uint64_t offset,
                        void *stream);
/* RandomNetworkAgent.forward (carle/agents.py:45-102): obs (N,1,H,W) fp32 ->
 *   Conv2d(1,4,3,p1) - ReLU - MaxPool(2) - Conv2d(4,1,3,p1) - ReLU - MaxPool(2) - Flatten - Linear(dense, OUT) -
 *   Sigmoid, all without bias (agents.py:78-87);  action[n][o] = 1.0f if sigmoid <= toggle_rate else 0.0f
 *   (agents.py:98), laid out (N, OUT) = the reference's reshape to (N,1,AW,AH).
 * conv1_w [4][1][3][3], conv2_w [1][4][3][3], lin_w [OUT][dense] with dense = (H/4)*(W/4): the
 * state_dict tensors `network.0.weight`, `network.3.weight`, `network.7.weight`.  H, W multiples of 8.
 * workspace: carle_agent_rn_workspace_floats(n, h, w, out_nodes) floats (the pooled features and, for small
 * batches, the K-split partial logits).  output (nullable): the (N, OUT) sigmoid values.  The linear layer runs
 * on the fp32 matrix cores (MFMA). */
int64_t carle_agent_rn_workspace_floats(int64_t n, int h, int w, int out_nodes);
int carle_agent_rn_forward_f32(const float *obs, int64_t n, int h, int w, const float *conv1_w,
                               const float *conv2_w, const float *lin_w, int out_nodes, float toggle_rate,
                               float *workspace, float *output, float *action, void *stream);
/* carle_ca_step_f32 with the action drawn IN the kernel (identical to generating the action with
 * carle_bernoulli_f32(seed, offset) and passing it in): no action tensor is read.  64x64 grids
 * with AW % 8 == 0, or 256 / 512 grids with AW % 4 == 0 and (W-AW) % 8 == 0.  action_out (nullable):
 * also write the drawn (N,1,AW,AW) action. */
int carle_ca_step_f32_rand(const float *u_in, float *u_out, int64_t n, int h, int w, int ah, int aw,
                           uint32_t rule, float p, uint64_t seed, uint64_t offset, float *action_out,
                           int32_t *status, int slot, void *stream);
/* ... followed by carle_ca_finish: the whole CARLE.step for a RandomAgent-driven environment.
 * step_flags: CARLE_STEP_MERGE_MOMENTS or 0 (the action predicates are always published). */
int carle_env_step_f32_rand(const float *u_in, float *u_out, float *reward, int64_t n, int h, int w,
                            int ah, int aw, uint32_t rule, float p, uint64_t seed, uint64_t offset,
                            float *action_out, int32_t *status, int slot, int step_flags, void *stream);

/* The fused random agent on uint8 state (same geometries as carle_ca_step_f32_rand) and on bit-packed
 * state (H == W in {64, 128, 256, 512}, the 64x64 window): bit-identical to generating the action with
 * carle_bernoulli_f32(seed, offset) and passing it (packed) to carle_ca_step_u8 / carle_ca_step_b1.
 * (carle/agents.py:35-42 + carle/env.py:150-182, 219-229.)  carle_env_step_*_rand = + carle_ca_finish. */
int carle_ca_step_u8_rand(const uint8_t *u_in, uint8_t *u_out, int64_t n, int h, int w, int ah, int aw,
                          uint32_t rule, float p, uint64_t seed, uint64_t offset, float *action_out,
                          int32_t *status, int slot, void *stream);
int carle_ca_step_b1_rand(const uint64_t *u_in, uint64_t *u_out, int64_t n, int h, int w, int ah, int aw,
                          uint32_t rule, float p, uint64_t seed, uint64_t offset, int32_t *status, int slot,
                          void *stream);
int carle_env_step_u8_rand(const uint8_t *u_in, uint8_t *u_out, float *reward, int64_t n, int h, int w,
                           int ah, int aw, uint32_t rule, float p, uint64_t seed, uint64_t offset,
                           int32_t *status, int slot, int step_flags, void *stream);
int carle_env_step_b1_rand(const uint64_t *u_in, uint64_t *u_out, float *reward, int64_t n, int h, int w,
                           int ah, int aw, uint32_t rule, float p, uint64_t seed, uint64_t offset,
                           int32_t *status, int slot, int step_flags, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Per-instance sums for the hand-coded reward wrappers: for every instance i
 * of x (n x elems floats): out4[4i] = sum x, out4[4i+k+1] = sum x * w_k for the (nullable) weight
 * planes w0..w2 of `elems` floats each.  Replaces torch.sum(universe, dim=[1,2,3]) (mcl.py:769),
 * the masked sums of CornerBonus (mcl.py:222-223), the centre-of-mass numerators of SpeedDetector
 * (mcl.py:772-775) and action.sum(axis=[1,2,3]) of ParsimonyBonus (mcl.py:102-103).
 * ------------------------------------------------------------------------------------------- */
int carle_instance_sums_f32(const float *x, int64_t n, int64_t elems, const float *w0,
                            const float *w1, const float *w2, float *out4, void *stream);

/* SpeedDetector.step behind the reduction (carle/mcl.py:769-797), no host round trip: from sums4 =
 * carle_instance_sums_f32(obs, mass_weight_h, mass_weight_w) (and live4 = the sums of the inner universe where that is
 * not the observation; NULL otherwise) computes the centre of mass per instance, velocity (2,N) = previous - new,
 * speed = sqrt(sum of all 2N squares) (one scalar for the whole batch, mcl.py:789) and adds it to every reward.
 * com (2,N) holds the previous centre of mass (in/out), state[0] says whether it is valid yet (0 on the first step:
 * then only com / live_out are written, as mcl.py:781-782); velocity (2,N), speed (1), live_out (N) are outputs. */
int carle_speed_update_f32(const float *sums4, const float *live4, int64_t n, float *com, int32_t *state,
                           float *velocity, float *speed, float *live_out, float *reward, void *stream);
/* PufferDetector.step behind the reduction (carle/mcl.py:832-851): live_total = sum_i sums4[4i]; if the step had no
 * non-zero action element (any_flag: device int32, e.g. status + CARLE_ST_ANY_NONZERO(slot) of the step just taken;
 * NULL = no action) the total is appended to the history and, once it holds more than cap - 1 = growth_threshold
 * entries, the oldest is dropped and every reward grows by 1 if newest - oldest > 0.01; else the history is cleared.
 * ring: cap floats, ring_state: int32[2] = {head, count}, both caller-owned and zero-initialised. */
int carle_puffer_update_f32(const float *sums4, int64_t n, const int32_t *any_flag, float *ring,
                            int32_t *ring_state, int cap, float *live_total, float *reward, void *stream);
/* MorphoBonus.step (carle/mcl.py:170-183): my_grid = |universe - action| (the grid BEFORE the environment step; action with
 * the universe's shape), conv_obs = conv2d(my_grid, patterns) with K <= 32 patterns of 8 x 8 and no padding,
 * out2[2 i] / out2[2 i + 1] = max / min of instance i's conv_obs over all patterns and positions, and (reward nullable)
 * reward[i] += reward_scale * (max + min).  8 <= H, 8 <= W <= 512; partial: carle_morpho_partial_floats(n, h) floats. */
int64_t carle_morpho_partial_floats(int64_t n, int h);
int carle_morpho_match_f32(const float *universe, const float *action, const float *patterns, int k_patterns, int64_t n,
                           int h, int w, float *partial, float *out2, float *reward, float reward_scale, void *stream);
/* PredictionBonus / SurpriseBonus.step behind the bonus (carle/mcl.py:686-694, 715-723): b_i = offset + sign * err[i]
 * (offset 0.1, sign -1: PredictionBonus; 0, +1: SurpriseBonus), b_i *= 0 where sums4[4 i] == 0 (an instance without a live
 * cell; sums4 = carle_instance_sums_f32 of the observation), reward[i] += scale * b_i (two roundings, as torch);
 * live_total (nullable): one float, the observation's total live cells (mcl.py:697, 726).  One launch, no host sync. */
int carle_bonus_nonempty_f32(float *reward, const float *err, const float *sums4, float offset, float sign, float scale,
                             int64_t n, float *live_total, void *stream);
/* PufferDetector(SpeedDetector(env)) -- the top of the evaluation stack, evaluation/eval.py:86-87 -- in ONE launch instead
 * of three: carle_instance_sums_f32(obs, w_h, w_w) into sums4 (n x 4 floats, an output), then -- by the workgroup that
 * finishes last -- carle_speed_update_f32 (live4 = NULL: the observation is the inner universe) and
 * carle_puffer_update_f32 on those sums, in that order (mcl.py:795, then mcl.py:843).  counter: one device int32, zero
 * before the first call (the kernel leaves it zero).  All other arguments as for the three entry points. */
int carle_speed_puffer_f32(const float *obs, int64_t n, int64_t elems, const float *w_h, const float *w_w, float *sums4,
                           int32_t *counter, float *com, int32_t *state, float *velocity, float *speed, float *live_out,
                           const int32_t *any_flag, float *ring, int32_t *ring_state, int cap, float *live_total,
                           float *reward, void *stream);

/* The same two updates under BATCH SHARDING (one process per GPU, carle_amd/dist.py), where the reference's scalars couple
 * the whole batch: SpeedDetector's speed = sqrt(sum over ALL instances of velocity^2) (carle/mcl.py:789), PufferDetector's
 * live-cell total and "the action has a non-zero element" (mcl.py:832, 835).  Split around ONE all-reduce(SUM) of
 * exch = double[4] = { sum of this shard's velocity^2, 1.0 if a previous centre of mass existed, this shard's live cells,
 * 1.0 if this shard's action had a non-zero element }:
 *   carle_wrapper_scalars_local_f32   what & 1: SpeedDetector's per-instance part (com, velocity, live_out, state as in
 *                                     carle_speed_update_f32; exch[0], exch[1]); what & 2: PufferDetector's (exch[2] from
 *                                     column 0 of live4, or of sums4 when live4 is NULL; exch[3] from any_flag, NULL = no
 *                                     action).  Entries of a part not asked for are written as 0.
 *   -- caller: all-reduce(SUM) of exch over the ranks (RCCL / gloo) --
 *   carle_wrapper_scalars_apply_f32   what & 1: speed = sqrt(exch[0]) -> speed_out, reward += speed (if exch[1] != 0);
 *                                     what & 2: the history update of carle_puffer_update_f32 with total = exch[2] and
 *                                     "action" = exch[3] != 0, reward += 1 when it fires.  Same order as the stacked wrappers
 *                                     (speed first, mcl.py:795, then the growth bonus, mcl.py:843).
 * With one rank and no all-reduce the pair is carle_speed_update_f32 / carle_puffer_update_f32 (fp64 sums, same rounding). */
int carle_wrapper_scalars_local_f32(const float *sums4, const float *live4, int64_t n, float *com, int32_t *state,
                                    float *velocity, float *live_out, const int32_t *any_flag, double *exch, int what,
                                    void *stream);
int carle_wrapper_scalars_apply_f32(const double *exch, int what, float *speed, float *ring, int32_t *ring_state, int cap,
                                    float *live_total, float *reward, int64_t n, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Reward statistics for batch-sharded runs (build-side feature required by the north star; the
 * reference has none).  out3 / moments3 = fp64 (count, mean, M2) on the device.  The triples of
 * all ranks are all-gathered over RCCL and Chan-merged by carle_amd/dist.py; the merged triple
 * also gives the global reward sum of train_mcl.py:68-69 / eval.py:69-70.
 * ------------------------------------------------------------------------------------------- */
/* out3 := moments of x[0..n); if acc3 != NULL additionally acc3 := merge(acc3, out3) (Chan). */
int carle_batch_moments_f32(const float *x, int64_t n, double *out3, double *acc3, void *stream);
/* last3 := Chan-merge of gathered[world][3] in rank order; state3 := merge(state3, last3);
 * pending3 (nullable, may alias gathered when world == 1) := 0. */
int carle_merge_moments(const double *gathered, int world, double *state3, double *last3,
                        double *pending3, void *stream);
int carle_normalize_f32(const float *x, float *y, int64_t n, const double *moments3, double eps,
                        void *stream);

/* ---------------------------------------------------------------------------------------------
 * Run-length codec.  HOST pointers, synchronous, no HIP calls.
 * encode: body of CARLE.get_rle (carle/env.py:431-462) for h x w byte cells (non-zero = alive);
 *         returns the number of bytes of the body, writes at most `cap` of them (call with
 *         out = NULL to size the buffer).  Byte-compatible with the reference, quirks included.
 * decode: CARLE.rle_to_grid (carle/env.py:260-328) into h x w byte cells.
 * ------------------------------------------------------------------------------------------- */
int64_t carle_rle_encode_body(const uint8_t *cells, int h, int w, char *out, int64_t cap);
int carle_rle_decode_body(const char *rle, int64_t len, uint8_t *cells, int h, int w);

#ifdef __cplusplus
}
#endif
#endif /* CARLE_HIP_H */

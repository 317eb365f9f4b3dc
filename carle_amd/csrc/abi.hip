// abi.hip -- library-level C-ABI entry points: version, error text, and the step epilogue
// (master reset + counters, carle/env.py:208-216,142-145,200,230).
#include <stdlib.h>
#include "ca_cells.hpp"
#include "common.hpp"

namespace carle {

std::string &last_error() {
    thread_local std::string msg;
    return msg;
}

static int64_t env_i64(const char *name) {
    const char *e = getenv(name);
    return (e && *e) ? atoll(e) : -1;
}
static Knobs read_knobs() {
    Knobs k;
    k.stream_store_min_bytes = env_i64("CARLE_STREAM_STORE_MIN_BYTES");
    k.b1_lds_min_bytes = env_i64("CARLE_B1_LDS_MIN_BYTES");
    k.b1_nt_min_bytes = env_i64("CARLE_B1_NT_MIN_BYTES");
    k.b1_resident_bytes = env_i64("CARLE_B1_RESIDENT_BYTES");
    k.b1_hybrid_all = env_i64("CARLE_B1_HYBRID_ALL");
    k.b1_blocks = env_i64("CARLE_B1_BLOCKS");
    k.b1_units = env_i64("CARLE_B1_UNITS");
    k.wide_blocks = env_i64("CARLE_WIDE_BLOCKS");
    k.rnd_blocks = env_i64("CARLE_RND_BLOCKS");
    k.packed_grids = env_i64("CARLE_PACKED_GRIDS");
    k.rnd_cached_load_max_bytes = env_i64("CARLE_RND_CACHED_LOAD_MAX_BYTES");
    k.net_blocks = env_i64("CARLE_NET_BLOCKS");
    k.b1_dbg = env_i64("CARLE_B1_DBG");
    k.two_launch_step = env_i64("CARLE_TWO_LAUNCH_STEP");
    return k;
}
static Knobs &knobs_storage() {
    static Knobs k = read_knobs();       // C++11: initialised once, thread-safe
    return k;
}
const Knobs &knobs() { return knobs_storage(); }

// Applies the whole-batch master reset decided by the flags the step kernel published, zeroes
// the reward vector, maintains the counters and recycles the other flag slot.
__global__ __launch_bounds__(256) void ca_finish_kernel(uint4 *__restrict__ out, int64_t n16,
                                                        unsigned char *__restrict__ tail,
                                                        int tail_bytes, float *__restrict__ reward,
                                                        int64_t n, int32_t *__restrict__ st,
                                                        int slot, int has_action,
                                                        uint4 *__restrict__ extra, int64_t extra16,
                                                        float *__restrict__ bonus, float reward_scale,
                                                        int merge_moments) {
    // the whole status block in one round trip (it is one 64-byte line)
    const uint4 *st4 = reinterpret_cast<const uint4 *>(st);
    const uint4 w0 = st4[0], w1 = st4[1], w2 = st4[2], w3 = st4[3];
    double *mom_out = reinterpret_cast<double *>(((uint64_t)w3.y << 32) | w3.x);
    double *mom_acc = reinterpret_cast<double *>(((uint64_t)w3.w << 32) | w3.z);
    const bool moments = merge_moments && reward && mom_out && mom_acc;
    // reward moments a one-launch CA + RND2D step left pending are merged first (wavefront 0 of block 0: the same
    // wavefront that merges this step's below)
    if (blockIdx.x == 0 && threadIdx.x < 64) settle_moments_wave(st, threadIdx.x);
    const int32_t any_flag = slot ? (int32_t)w0.y : (int32_t)w0.x;
    const int32_t notone_flag = slot ? (int32_t)w0.w : (int32_t)w0.z;
    const bool reset = has_action && notone_flag == 0;
    const bool any = has_action && any_flag != 0;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    if (reset) {
        const uint4 z = {0u, 0u, 0u, 0u};
        for (int64_t i = tid; i < n16; i += nthreads) out[i] = z;
        if (tid < tail_bytes) tail[tid] = 0;
        for (int64_t i = tid; i < extra16; i += nthreads) extra[i] = z;   // side copy of the state
    }
    if (moments) {
        // block 0 writes every reward itself and reduces what it writes: (count, mean, M2) of the
        // step's final rewards, merged into the running triple (the arithmetic of
        // batch_moments_kernel, stats.hip).  The running triple is requested first: its round trip
        // hides under the reward pass.
        if (blockIdx.x == 0) {
            __shared__ double part[2][4];
            double na = 0.0, ma = 0.0, qa = 0.0;
            if (threadIdx.x == 0) { na = mom_acc[0]; ma = mom_acc[1]; qa = mom_acc[2]; }
            double sum = 0.0, sq = 0.0;
            if (bonus == nullptr) {
                // bare CARLE.step: the reward is all zero (env.py:238), nothing to reduce
                for (int64_t i = threadIdx.x; i < n; i += blockDim.x) reward[i] = 0.0f;
            } else {
                const float dead = bonus[n];
                double s = 0.0, q = 0.0;
                // eight independent loads in flight per thread and pass (latency-bound: one block)
                for (int64_t base = 0; base < n; base += 8 * (int64_t)blockDim.x) {
                    float b[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int64_t i = base + u * (int64_t)blockDim.x + threadIdx.x;
                        b[u] = (!reset && i < n) ? bonus[i] : dead;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int64_t i = base + u * (int64_t)blockDim.x + threadIdx.x;
                        if (i < n) {
                            const float r = reward_scale * b[u];
                            if (reset) bonus[i] = dead;
                            reward[i] = r;
                            s += (double)r;
                            q = fma((double)r, (double)r, q);
                        }
                    }
                }
#pragma unroll
                for (int sh = 32; sh > 0; sh >>= 1) { s += __shfl_xor(s, sh, 64); q += __shfl_xor(q, sh, 64); }
                if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = s; part[1][threadIdx.x >> 6] = q; }
                __syncthreads();
                sum = (part[0][0] + part[0][1]) + (part[0][2] + part[0][3]);
                sq = (part[1][0] + part[1][1]) + (part[1][2] + part[1][3]);
            }
            if (threadIdx.x == 0) {
                const double nb = (double)n, mean = n > 0 ? sum / nb : 0.0;
                double m2 = sq - sum * mean;
                if (m2 < 0.0) m2 = 0.0;
                double *row = moments_out_slot(st, mom_out);
                row[0] = nb; row[1] = mean; row[2] = m2;
                const double tot = na + nb;
                if (tot > 0.0) {
                    const double delta = mean - ma;
                    mom_acc[0] = tot;
                    mom_acc[1] = ma + delta * nb / tot;
                    mom_acc[2] = qa + m2 + delta * delta * na * nb / tot;
                }
            }
        }
    } else if (reward && bonus) {
        // fused CA + RND2D step: reward = 0 + scale * bonus (mcl.py:403-405); after a master reset
        // every observation is the all-dead grid, whose bonus the step kernel left in bonus[n]
        const float dead = bonus[n];
        for (int64_t i = tid; i < n; i += nthreads) {
            const float b = reset ? dead : bonus[i];
            if (reset) bonus[i] = dead;
            reward[i] = reward_scale * b;
        }
    } else if (reward) {
        for (int64_t i = tid; i < n; i += nthreads) reward[i] = 0.0f;
    }
    if (tid == 0) finish_status(st, w0, w1, w2, slot, reset, any);
}

}  // namespace carle

using namespace carle;

extern "C" int carle_abi_version(void) { return CARLE_ABI_VERSION; }

extern "C" const char *carle_last_error(void) { return last_error().c_str(); }

extern "C" void carle_reload_knobs(void) { knobs_storage() = read_knobs(); }

extern "C" const char *carle_build_arch(void) {
#ifdef CARLE_BUILD_ARCH
    return CARLE_BUILD_ARCH;
#else
    return "unknown";
#endif
}

namespace carle {
int ca_finish_impl(void *u_out, int64_t u_out_bytes, float *reward, int64_t n, int32_t *status, int slot,
                   int has_action, void *extra, int64_t extra_bytes, float *bonus, float reward_scale,
                   int merge_moments, void *stream);
}

extern "C" int carle_ca_finish(void *u_out, int64_t u_out_bytes, float *reward, int64_t n,
                               int32_t *status, int slot, int has_action, void *stream) {
    return ca_finish_impl(u_out, u_out_bytes, reward, n, status, slot, has_action, nullptr, 0, nullptr, 0.0f,
                          0, stream);
}

// `extra` (16-byte multiple): a second buffer zeroed together with u_out on a master reset.
// `bonus` (n + 1 floats, nullable): reward := reward_scale * bonus instead of zero (fused RND2D step).
int carle::ca_finish_impl(void *u_out, int64_t u_out_bytes, float *reward, int64_t n, int32_t *status,
                          int slot, int has_action, void *extra, int64_t extra_bytes, float *bonus,
                          float reward_scale, int merge_moments, void *stream) {
    if (!status) return fail_arg(CARLE_E_NULL, "carle_ca_finish: status is NULL");
    if (!u_out && u_out_bytes > 0) return fail_arg(CARLE_E_NULL, "carle_ca_finish: u_out is NULL");
    if (slot != 0 && slot != 1) return fail_arg(CARLE_E_DIMS, "carle_ca_finish: slot must be 0 or 1");
    if (u_out_bytes < 0 || n < 0) return fail_arg(CARLE_E_DIMS, "carle_ca_finish: negative size");
    if (!aligned16(u_out) || !aligned16(status))
        return fail_arg(CARLE_E_ALIGN, "carle_ca_finish: pointers must be 16-byte aligned");
    const int64_t n16 = u_out_bytes / 16;
    const int tail = (int)(u_out_bytes % 16);
    // fixed grid: in the common (no reset) case every block but the first exits after one flag
    // load; on a master reset the blocks zero the output together (8 blocks: 870 us for the 1 GiB of
    // N=4096 256x256 grids; 128 blocks: see profiles/r01_bench_matrix.txt)
    constexpr int kFinishBlocks = 128;
    hipLaunchKernelGGL(ca_finish_kernel, dim3(kFinishBlocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<uint4 *>(u_out), n16,
                       reinterpret_cast<unsigned char *>(u_out) + n16 * 16, tail, reward, n, status,
                       slot, (n > 0 && has_action) ? 1 : 0, reinterpret_cast<uint4 *>(extra),
                       extra_bytes / 16, bonus, reward_scale, merge_moments);
    return check_launch("carle_ca_finish");
}

namespace carle {
struct RandAct;
int env_step_small64(const float *u_in, const float *action, float *u_out, float *reward, int64_t n, int h,
                     int w, int ah, int aw, uint32_t rule, int32_t *status, int slot, int step_flags,
                     const RandAct *rand, void *stream);
int env_step_small64_rand(const float *u_in, float *u_out, float *reward, int64_t n, int h, int w, int ah, int aw,
                          uint32_t rule, float p, uint64_t seed, uint64_t offset, float *action_out,
                          int32_t *status, int slot, int step_flags, void *stream);
int env_run_small64(const float *u_in, const float *actions, float *u_out, float *reward, int64_t n, int h, int w, int ah,
                    int aw, uint32_t rule, float p, uint64_t seed, uint64_t offset, int k, int32_t *status, int slot0,
                    int step_flags, void *stream);
int env_run_noaction64(const float *u_in, float *u_out, float *reward, int64_t n, int h, int w, uint32_t rule, int k,
                       int32_t *status, int slot0, int step_flags, void *stream);
int env_step_one_launch_f32(const float *u_in, const float *action, float *u_out, float *reward, int64_t n, int h,
                            int w, int ah, int aw, uint32_t rule, int32_t *status, int slot, int step_flags,
                            void *stream, uint64_t *bits_out = nullptr, int *bits_written = nullptr,
                            const uint64_t *bits_in = nullptr);
int env_step_one_launch_u8(const uint8_t *u_in, const void *action, int action_fmt, uint8_t *u_out, float *reward,
                           int64_t n, int h, int w, int ah, int aw, uint32_t rule, int32_t *status, int slot,
                           int step_flags, void *stream);
int ca_step_wide_packed_f32(const float *u_in, const uint64_t *bits_in, const float *action, float *u_out, uint64_t *bits_out,
                            int64_t n, int h, int w, int ah, int aw, uint32_t rule, int32_t *status, int slot, void *stream);
int env_step_one_launch_b1(const uint64_t *u_in, const uint64_t *action, uint64_t *u_out, float *reward, int64_t n, int h,
                           int w, int ah, int aw, uint32_t rule, int32_t *status, int slot, int step_flags, void *stream);

__global__ __launch_bounds__(64) void env_settle_kernel(int32_t *st) {
    settle_moments_wave(st, threadIdx.x);
    if (threadIdx.x == 0) settle_status_in_place(st);
}
}

extern "C" int carle_env_settle(int32_t *status, void *stream) {
    if (!status) return fail_arg(CARLE_E_NULL, "carle_env_settle: status is NULL");
    if (!aligned16(status)) return fail_arg(CARLE_E_ALIGN, "carle_env_settle: status must be 16-byte aligned");
    hipLaunchKernelGGL(env_settle_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), status);
    return check_launch("carle_env_settle");
}

// step + epilogue in one call: the per-step host cost of the drop-in surface is dominated by
// Python/ctypes overhead, so CARLE.step makes ONE library call per step.  Batches of up to 16
// 64x64 grids take ONE launch (step and epilogue in the same single-workgroup kernel).
extern "C" int carle_env_step_f32(const float *u_in, const float *action, float *u_out,
                                  float *reward, int64_t n, int h, int w, int ah, int aw,
                                  uint32_t rule, int32_t *status, int slot, int publish_flags,
                                  void *stream) {
    if (!status) return fail_arg(CARLE_E_NULL, "carle_env_step_f32: status is NULL");
    if (const int e = env_step_small64(u_in, action, u_out, reward, n, h, w, ah, aw, rule, status, slot,
                                       publish_flags, nullptr, stream); e != -1000)
        return e;
    if (publish_flags & CARLE_STEP_DEFERRED)
        if (const int e = env_step_one_launch_f32(u_in, action, u_out, reward, n, h, w, ah, aw, rule, status, slot,
                                                  publish_flags, stream); e != -1000)
            return e;
    if (int e = carle_ca_step_f32(u_in, action, u_out, n, h, w, ah, aw, rule,
                                  (publish_flags & CARLE_STEP_PUBLISH_FLAGS) ? status : nullptr, slot, stream))
        return e;
    return ca_finish_impl(u_out, (int64_t)n * h * w * (int64_t)sizeof(float), reward, n, status, slot,
                          action != nullptr, nullptr, 0, nullptr, 0.0f,
                          publish_flags & CARLE_STEP_MERGE_MOMENTS, stream);
}

// carle_env_step_f32 that ALSO leaves a bit-packed copy of the new state where the launch can produce it for free (the
// one-launch step of 64 / 256 / 512-column grids: the reward-net kernels read 512 bytes per grid / tile instead of 16 / 22 KB);
// *bits_written (host) says whether it did -- 0: bits_out was not touched.
extern "C" int carle_env_step_f32_bits(const float *u_in, const float *action, float *u_out, float *reward, int64_t n,
                                       int h, int w, int ah, int aw, uint32_t rule, int32_t *status, int slot,
                                       int publish_flags, uint64_t *bits_out, int *bits_written, void *stream) {
    if (bits_written) *bits_written = 0;
    if (!status) return fail_arg(CARLE_E_NULL, "carle_env_step_f32_bits: status is NULL");
    if (bits_out && (publish_flags & CARLE_STEP_DEFERRED))
        if (const int e = env_step_one_launch_f32(u_in, action, u_out, reward, n, h, w, ah, aw, rule, status, slot,
                                                  publish_flags, stream, bits_out, bits_written); e != -1000)
            return e;
    // 256 / 512 columns beyond the one-launch step (or with it switched off): step kernel with the copy + epilogue, which
    // clears the copy together with the state on a master reset
    if (bits_out && aligned16(bits_out))
        if (const int e = ca_step_wide_packed_f32(u_in, nullptr, action, u_out, bits_out, n, h, w, ah, aw, rule,
                                                  (publish_flags & CARLE_STEP_PUBLISH_FLAGS) ? status : nullptr, slot, stream);
            e != -1000) {
            if (e) return e;
            if (bits_written) *bits_written = 1;
            return ca_finish_impl(u_out, (int64_t)n * h * w * (int64_t)sizeof(float), reward, n, status, slot, action != nullptr,
                                  bits_out, (int64_t)n * h * (w / 8), nullptr, 0.0f, publish_flags & CARLE_STEP_MERGE_MOMENTS, stream);
        }
    return carle_env_step_f32(u_in, action, u_out, reward, n, h, w, ah, aw, rule, status, slot, publish_flags, stream);
}

// The one-launch step of N x 64 x 64 fp32 grids whose INPUT state is taken from the bit-packed copy a previous
// carle_env_step_f32_bits / carle_env_step_f32_packed / carle_pack_f32_b1 left of it (bits_in, n x 64 row words) instead of the
// fp32 cells: the launch reads 512 B instead of 16 KiB per grid, writes the fp32 observation u_out (carle/env.py:229) and its
// packed copy bits_out as before.  The caller vouches that bits_in IS the current state (CARLE.step: the very tensor the last
// step returned, unmodified since).  CARLE_STEP_DEFERRED semantics (one launch, counters settled lazily).
extern "C" int carle_env_step_f32_packed(const uint64_t *bits_in, const float *action, float *u_out, float *reward, int64_t n,
                                         int h, int w, int ah, int aw, uint32_t rule, int32_t *status, int slot,
                                         int step_flags, uint64_t *bits_out, void *stream) {
    if (!status || !bits_in || !bits_out) return fail_arg(CARLE_E_NULL, "carle_env_step_f32_packed: NULL pointer");
    if (n < 1 || h != w || (w != 64 && w != 256 && w != 512))
        return fail_arg(CARLE_E_DIMS, "carle_env_step_f32_packed: n >= 1 grids of 64x64, 256x256 or 512x512");
    if (w == 64) {
        if (action && (aw % 8) != 0) return fail_arg(CARLE_E_DIMS, "carle_env_step_f32_packed: action width must be a multiple of 8");
        if (knobs().two_launch_step > 0) return fail_arg(CARLE_E_DIMS, "carle_env_step_f32_packed: CARLE_TWO_LAUNCH_STEP is set");
    }
    const int e = env_step_one_launch_f32(nullptr, action, u_out, reward, n, h, w, ah, aw, rule, status, slot,
                                          step_flags | CARLE_STEP_DEFERRED, stream, bits_out, nullptr, bits_in);
    if (e != -1000) return e;
    // 256 / 512 columns beyond the one-launch step: step kernel + epilogue (which clears state AND copy on a master reset)
    if (!aligned16(bits_in) || !aligned16(bits_out))
        return fail_arg(CARLE_E_ALIGN, "carle_env_step_f32_packed: bits_in / bits_out must be 16-byte aligned");
    const int e2 = ca_step_wide_packed_f32(nullptr, bits_in, action, u_out, bits_out, n, h, w, ah, aw, rule,
                                           (step_flags & CARLE_STEP_PUBLISH_FLAGS) ? status : nullptr, slot, stream);
    if (e2 == -1000) return fail_arg(CARLE_E_DIMS, "carle_env_step_f32_packed: geometry without a packed-input step kernel");
    if (e2) return e2;
    return ca_finish_impl(u_out, (int64_t)n * h * w * (int64_t)sizeof(float), reward, n, status, slot, action != nullptr,
                          bits_out, (int64_t)n * h * (w / 8), nullptr, 0.0f, step_flags & CARLE_STEP_MERGE_MOMENTS, stream);
}

extern "C" int carle_ca_step_b1(const uint64_t *, const uint64_t *, uint64_t *, int64_t, int, int, int,
                                int, uint32_t, int32_t *, int, void *);

// Whole CARLE.step on bit-packed state: packed step kernel + epilogue, one call.
extern "C" int carle_env_step_b1(const uint64_t *u_in, const uint64_t *action, uint64_t *u_out,
                                 float *reward, int64_t n, int h, int w, int ah, int aw,
                                 uint32_t rule, int32_t *status, int slot, int publish_flags,
                                 void *stream) {
    if (!status) return fail_arg(CARLE_E_NULL, "carle_env_step_b1: status is NULL");
    if (publish_flags & CARLE_STEP_DEFERRED)
        if (const int e = env_step_one_launch_b1(u_in, action, u_out, reward, n, h, w, ah, aw, rule, status, slot,
                                                 publish_flags, stream); e != -1000)
            return e;
    if (int e = carle_ca_step_b1(u_in, action, u_out, n, h, w, ah, aw, rule,
                                 (publish_flags & CARLE_STEP_PUBLISH_FLAGS) ? status : nullptr, slot, stream))
        return e;
    return ca_finish_impl(u_out, (int64_t)n * h * (w / 8), reward, n, status, slot, action != nullptr,
                          nullptr, 0, nullptr, 0.0f, publish_flags & CARLE_STEP_MERGE_MOMENTS, stream);
}

extern "C" int carle_ca_step_u8(const uint8_t *, const void *, int, uint8_t *, int64_t, int, int, int, int,
                                uint32_t, int32_t *, int, void *);

// Whole CARLE.step on uint8 state (fp32 or uint8 action): step kernel + epilogue, one call.
extern "C" int carle_env_step_u8(const uint8_t *u_in, const void *action, int action_fmt, uint8_t *u_out,
                                 float *reward, int64_t n, int h, int w, int ah, int aw, uint32_t rule,
                                 int32_t *status, int slot, int publish_flags, void *stream) {
    if (!status) return fail_arg(CARLE_E_NULL, "carle_env_step_u8: status is NULL");
    if (publish_flags & CARLE_STEP_DEFERRED)
        if (const int e = env_step_one_launch_u8(u_in, action, action_fmt, u_out, reward, n, h, w, ah, aw, rule, status,
                                                 slot, publish_flags, stream); e != -1000)
            return e;
    if (int e = carle_ca_step_u8(u_in, action, action_fmt, u_out, n, h, w, ah, aw, rule,
                                 (publish_flags & CARLE_STEP_PUBLISH_FLAGS) ? status : nullptr, slot, stream))
        return e;
    return ca_finish_impl(u_out, (int64_t)n * h * w, reward, n, status, slot, action != nullptr, nullptr, 0,
                          nullptr, 0.0f, publish_flags & CARLE_STEP_MERGE_MOMENTS, stream);
}

extern "C" int carle_ca_step_f32_rand(const float *, float *, int64_t, int, int, int, int, uint32_t, float,
                                      uint64_t, uint64_t, float *, int32_t *, int, void *);

// CARLE.step with the fused device-side random agent + epilogue, one call.
extern "C" int carle_env_step_f32_rand(const float *u_in, float *u_out, float *reward, int64_t n, int h,
                                       int w, int ah, int aw, uint32_t rule, float p, uint64_t seed,
                                       uint64_t offset, float *action_out, int32_t *status, int slot,
                                       int step_flags, void *stream) {
    if (!status) return fail_arg(CARLE_E_NULL, "carle_env_step_f32_rand: status is NULL");
    if (const int e = env_step_small64_rand(u_in, u_out, reward, n, h, w, ah, aw, rule, p, seed, offset, action_out,
                                            status, slot, step_flags, stream); e != -1000)
        return e;
    if (int e = carle_ca_step_f32_rand(u_in, u_out, n, h, w, ah, aw, rule, p, seed, offset, action_out,
                                       status, slot, stream))
        return e;
    return ca_finish_impl(u_out, (int64_t)n * h * w * (int64_t)sizeof(float), reward, n, status, slot, 1,
                          nullptr, 0, nullptr, 0.0f, step_flags & CARLE_STEP_MERGE_MOMENTS, stream);
}

// k consecutive CARLE.step calls in ONE library call: the loop of carle/train_mcl.py:64-69 with the
// actions already on the device (a stack of k action tensors) or drawn in the step kernel (the
// device-side RandomAgent).  The host pays one ctypes round trip for k steps instead of k.
extern "C" int carle_env_run_f32(const float *u_in, float *buf_a, float *buf_b, const float *actions, float *reward,
                                 int64_t n, int h, int w, int ah, int aw, uint32_t rule, float p, uint64_t seed,
                                 uint64_t offset, int k, int32_t *status, int slot0, int step_flags,
                                 void *stream) {
    if (!status) return fail_arg(CARLE_E_NULL, "carle_env_run_f32: status is NULL");
    if (k < 0) return fail_arg(CARLE_E_DIMS, "carle_env_run_f32: k must be >= 0");
    if (slot0 != 0 && slot0 != 1) return fail_arg(CARLE_E_DIMS, "carle_env_run_f32: slot0 must be 0 or 1");
    if (k > 0 && (!u_in || !buf_a || (k > 1 && !buf_b)))
        return fail_arg(CARLE_E_NULL, "carle_env_run_f32: u_in / buf_a / buf_b is NULL");
    if (k > 1 && n > 0 && h > 0 && w > 0) {
        const size_t bytes = (size_t)n * h * w * sizeof(float);
        if (overlap(buf_a, bytes, buf_b, bytes) || overlap(u_in, bytes, buf_b, bytes))
            return fail_arg(CARLE_E_ALIAS, "carle_env_run_f32: u_in / buf_a / buf_b overlap");
    }
    const int64_t act_elems = n * (int64_t)ah * aw;
    float *fin = (k & 1) ? buf_a : buf_b;                  // the buffer of the final state
    if (k > 0 && n >= 1 && n <= 16 && h == 64 && w == 64) {
        // small batches: all k steps in ONE launch, the grids stay in registers (ca_env_run_small64_kernel)
        const int e = env_run_small64(u_in, actions, fin, reward, n, h, w, ah, aw, rule, p, seed, offset, k,
                                      status, slot0, step_flags, stream);
        if (e != -1000) return e;
    }
    if (k > 0 && !actions && p < 0.0f && h == 64 && w == 64 && n > 16) {
        // no action source: no predicates, no master reset -- all k generations in one launch, grids in registers
        const int e = env_run_noaction64(u_in, fin, reward, n, h, w, rule, k, status, slot0, step_flags, stream);
        if (e != -1000) return e;
    }
    for (int t = 0; t < k; ++t) {
        const float *in = t == 0 ? u_in : ((t & 1) ? buf_a : buf_b);
        float *out = (t & 1) ? buf_b : buf_a;
        const int slot = (slot0 + t) & 1;
        int e;
        if (actions)
            e = carle_env_step_f32(in, actions + t * act_elems, out, reward, n, h, w, ah, aw, rule, status, slot,
                                   step_flags | CARLE_STEP_PUBLISH_FLAGS, stream);
        else if (p >= 0.0f)
            e = carle_env_step_f32_rand(in, out, reward, n, h, w, ah, aw, rule, p, seed, offset + (uint64_t)t, nullptr,
                                        status, slot, step_flags, stream);
        else
            e = carle_env_step_f32(in, nullptr, out, reward, n, h, w, 0, 0, rule, status, slot, step_flags, stream);
        if (e) return e;
    }
    return 0;
}

extern "C" int carle_ca_step_u8_rand(const uint8_t *, uint8_t *, int64_t, int, int, int, int, uint32_t, float, uint64_t,
                                     uint64_t, float *, int32_t *, int, void *);
extern "C" int carle_ca_step_b1_rand(const uint64_t *, uint64_t *, int64_t, int, int, int, int, uint32_t, float,
                                     uint64_t, uint64_t, int32_t *, int, void *);

// Whole CARLE.step with the fused device-side random agent on uint8 / bit-packed state.
extern "C" int carle_env_step_u8_rand(const uint8_t *u_in, uint8_t *u_out, float *reward, int64_t n, int h, int w,
                                      int ah, int aw, uint32_t rule, float p, uint64_t seed, uint64_t offset,
                                      int32_t *status, int slot, int step_flags, void *stream) {
    if (!status) return fail_arg(CARLE_E_NULL, "carle_env_step_u8_rand: status is NULL");
    if (int e = carle_ca_step_u8_rand(u_in, u_out, n, h, w, ah, aw, rule, p, seed, offset, nullptr, status, slot, stream))
        return e;
    return ca_finish_impl(u_out, (int64_t)n * h * w, reward, n, status, slot, 1, nullptr, 0, nullptr, 0.0f,
                          step_flags & CARLE_STEP_MERGE_MOMENTS, stream);
}

extern "C" int carle_env_step_b1_rand(const uint64_t *u_in, uint64_t *u_out, float *reward, int64_t n, int h, int w,
                                      int ah, int aw, uint32_t rule, float p, uint64_t seed, uint64_t offset,
                                      int32_t *status, int slot, int step_flags, void *stream) {
    if (!status) return fail_arg(CARLE_E_NULL, "carle_env_step_b1_rand: status is NULL");
    if (int e = carle_ca_step_b1_rand(u_in, u_out, n, h, w, ah, aw, rule, p, seed, offset, status, slot, stream))
        return e;
    return ca_finish_impl(u_out, (int64_t)n * h * (w / 8), reward, n, status, slot, 1, nullptr, 0, nullptr, 0.0f,
                          step_flags & CARLE_STEP_MERGE_MOMENTS, stream);
}

// Orders `to_stream` after everything enqueued so far on `from_stream` (event record + wait):
// lets the reward networks of step t run on their own stream beside the CA kernel of step t+1.
extern "C" int carle_stream_fork(void *event, void *from_stream, void *to_stream) {
    if (!event) return fail_arg(CARLE_E_NULL, "carle_stream_fork: event is NULL");
    hipError_t e = hipEventRecord(static_cast<hipEvent_t>(event), static_cast<hipStream_t>(from_stream));
    if (e != hipSuccess) return fail_hip(e, "carle_stream_fork(record)");
    e = hipStreamWaitEvent(static_cast<hipStream_t>(to_stream), static_cast<hipEvent_t>(event), 0);
    if (e != hipSuccess) return fail_hip(e, "carle_stream_fork(wait)");
    return 0;
}

extern "C" int carle_rnd2d_bonus_f32(const float *, int64_t, int, int, const float *, const float *,
                                     float *, float *, float *, float *, float *, float, int, void *);

// CARLE.step followed by the RND2D wrapper's reward bonus on the fresh observation
// (carle/env.py:188-242 + carle/mcl.py:398-407) in ONE library call: step + epilogue on
// `stream_env`; the bonus (reward += scale * bonus fused into its last kernel) on `stream_reward`,
// ordered behind the step through `event` -- or on `stream_env` itself when stream_reward is
// NULL / equal.  The observation is the kernel's own 0/1 output, so the fused table-lookup path
// of the reward net applies.  One call instead of four keeps the host below the GPU time.
namespace carle {
int ca_step_rnd64(const float *u_in, const float *action, float *u_out, int64_t n, int aw, uint32_t rule,
                  int32_t *status, int slot, const float *pp, const float *rp, float *ws, float *bonus,
                  bool tables_valid, hipStream_t s, const RndEnvArgs *env, bool packed_in = false);
int64_t rnd64_blocks(int64_t n);
}

// doubles of the `moment_partials` scratch of carle_env_step_f32_rnd: two buffers (one per flag slot) of one
// (sum, sum of squares) pair per workgroup
extern "C" int64_t carle_rnd_step_partials_doubles(int64_t n) { return n > 0 ? 4 * rnd64_blocks(n) : 4; }

// argument checks of the fused 64x64 CA + RND2D kernel (shared by the two entry points below)
static int rnd_step_checks(const char *who, const float *u_in, const float *action, float *u_out, int64_t n,
                           int h, int w, int ah, int aw, uint32_t rule, int slot, const float *workspace) {
    std::string p(who);
    if (!u_in || !u_out) return fail_arg(CARLE_E_NULL, (p + ": u_in/u_out is NULL").c_str());
    if (int e = check_dims(n, h, w, action != nullptr, ah, aw)) return e;
    if (rule >> 18) return fail_arg(CARLE_E_RULE, (p + ": rule mask exceeds 18 bits").c_str());
    if (slot != 0 && slot != 1) return fail_arg(CARLE_E_DIMS, (p + ": slot must be 0 or 1").c_str());
    if (!aligned16(u_in) || !aligned16(u_out) || (action && !aligned16(action)) || !aligned16(workspace))
        return fail_arg(CARLE_E_ALIGN, (p + ": pointers must be 16-byte aligned").c_str());
    const size_t bytes = (size_t)n * h * w * sizeof(float);
    if (overlap(u_in, bytes, u_out, bytes)) return fail_arg(CARLE_E_ALIAS, (p + ": u_in and u_out overlap").c_str());
    return 0;
}

// The fused kernel alone (no epilogue): one generation + the RND2D bonus of the new state.
extern "C" int carle_ca_step_f32_rnd(const float *u_in, const float *action, float *u_out, int64_t n, int h,
                                     int w, int ah, int aw, uint32_t rule, int32_t *status, int slot,
                                     const float *pred_params, const float *rand_params, float *workspace,
                                     float *bonus, int net_flags, void *stream) {
    if (!pred_params || !rand_params || !workspace || !bonus)
        return fail_arg(CARLE_E_NULL, "carle_ca_step_f32_rnd: NULL reward-net pointer");
    if (h != 64 || w != 64 || (action && (aw % 8) != 0) || n <= 0)
        return fail_arg(CARLE_E_DIMS, "carle_ca_step_f32_rnd: 64x64 grids with AW % 8 == 0 only");
    if (int e = rnd_step_checks("carle_ca_step_f32_rnd", u_in, action, u_out, n, h, w, ah, aw, rule, slot, workspace))
        return e;
    return ca_step_rnd64(u_in, action, u_out, n, aw, rule, status, slot, pred_params, rand_params, workspace,
                         bonus, (net_flags & CARLE_NET_TABLES_VALID) != 0, static_cast<hipStream_t>(stream), nullptr);
}

static int env_step_f32_rnd_impl(const float *u_in, const float *action, float *u_out,
                                 float *reward, int64_t n, int h, int w, int ah, int aw,
                                 uint32_t rule, int32_t *status, int slot, int publish_flags,
                                 const float *pred_params, const float *rand_params,
                                 float *workspace, float *bonus, float reward_scale,
                                 int net_flags, double *moment_partials, void *event, void *stream_env,
                                 void *stream_reward, uint64_t *bits_out, int *bits_written, const uint64_t *bits_in = nullptr);

extern "C" int carle_env_step_f32_rnd(const float *u_in, const float *action, float *u_out,
                                      float *reward, int64_t n, int h, int w, int ah, int aw,
                                      uint32_t rule, int32_t *status, int slot, int publish_flags,
                                      const float *pred_params, const float *rand_params,
                                      float *workspace, float *bonus, float reward_scale,
                                      int net_flags, double *moment_partials, void *event, void *stream_env,
                                      void *stream_reward) {
    return env_step_f32_rnd_impl(u_in, action, u_out, reward, n, h, w, ah, aw, rule, status, slot, publish_flags, pred_params,
                                 rand_params, workspace, bonus, reward_scale, net_flags, moment_partials, event, stream_env,
                                 stream_reward, nullptr, nullptr);
}

// The same, and where the step is ONE launch (the fused 64x64 kernel, CARLE_STEP_DEFERRED) the kernel also leaves a
// bit-packed copy of the observation in bits_out (n x 64 row words, 16-byte aligned): *bits_written says whether it did.
extern "C" int carle_env_step_f32_rnd_bits(const float *u_in, const float *action, float *u_out,
                                           float *reward, int64_t n, int h, int w, int ah, int aw,
                                           uint32_t rule, int32_t *status, int slot, int publish_flags,
                                           const float *pred_params, const float *rand_params,
                                           float *workspace, float *bonus, float reward_scale,
                                           int net_flags, double *moment_partials, void *event, void *stream_env,
                                           void *stream_reward, uint64_t *bits_out, int *bits_written) {
    if (bits_written) *bits_written = 0;
    if (bits_out && !aligned16(bits_out)) return fail_arg(CARLE_E_ALIGN, "carle_env_step_f32_rnd_bits: bits_out must be 16-byte aligned");
    return env_step_f32_rnd_impl(u_in, action, u_out, reward, n, h, w, ah, aw, rule, status, slot, publish_flags, pred_params,
                                 rand_params, workspace, bonus, reward_scale, net_flags, moment_partials, event, stream_env,
                                 stream_reward, bits_out, bits_written);
}

// carle_env_step_f32_rnd_bits whose INPUT state is read from the bit-packed copy a previous launch left of it (bits_in, n x 64
// row words) instead of the fp32 cells -- see carle_env_step_f32_packed.  One launch or an error: n > 16 grids of 64x64, an
// 8-aligned window, CARLE_STEP_DEFERRED.  bits_in and bits_out must not overlap.
extern "C" int carle_env_step_f32_rnd_packed(const uint64_t *bits_in, const float *action, float *u_out,
                                             float *reward, int64_t n, int h, int w, int ah, int aw,
                                             uint32_t rule, int32_t *status, int slot, int publish_flags,
                                             const float *pred_params, const float *rand_params,
                                             float *workspace, float *bonus, float reward_scale,
                                             int net_flags, double *moment_partials, void *event, void *stream_env,
                                             void *stream_reward, uint64_t *bits_out) {
    if (!bits_in || !bits_out || !reward) return fail_arg(CARLE_E_NULL, "carle_env_step_f32_rnd_packed: NULL pointer");
    if (!aligned16(bits_in) || !aligned16(bits_out))
        return fail_arg(CARLE_E_ALIGN, "carle_env_step_f32_rnd_packed: bits_in / bits_out must be 16-byte aligned");
    if (h != 64 || w != 64 || n <= 16 || (action && (aw % 8) != 0))
        return fail_arg(CARLE_E_DIMS, "carle_env_step_f32_rnd_packed: more than 16 grids of 64x64, action width a multiple of 8");
    if (overlap(bits_in, (size_t)n * 512, bits_out, (size_t)n * 512))
        return fail_arg(CARLE_E_ALIAS, "carle_env_step_f32_rnd_packed: bits_in and bits_out overlap");
    return env_step_f32_rnd_impl(nullptr, action, u_out, reward, n, h, w, ah, aw, rule, status, slot,
                                 publish_flags | CARLE_STEP_DEFERRED, pred_params, rand_params, workspace, bonus, reward_scale,
                                 net_flags, moment_partials, event, stream_env, stream_reward, bits_out, nullptr, bits_in);
}

static int env_step_f32_rnd_impl(const float *u_in, const float *action, float *u_out,
                                 float *reward, int64_t n, int h, int w, int ah, int aw,
                                 uint32_t rule, int32_t *status, int slot, int publish_flags,
                                 const float *pred_params, const float *rand_params,
                                 float *workspace, float *bonus, float reward_scale,
                                 int net_flags, double *moment_partials, void *event, void *stream_env,
                                 void *stream_reward, uint64_t *bits_out, int *bits_written, const uint64_t *bits_in) {
    if (!status) return fail_arg(CARLE_E_NULL, "carle_env_step_f32_rnd: status is NULL");
    if (!pred_params || !rand_params || !workspace || !bonus)
        return fail_arg(CARLE_E_NULL, "carle_env_step_f32_rnd: NULL reward-net pointer");
    const bool tables_valid = (net_flags & CARLE_NET_TABLES_VALID) != 0;
    const bool fork = stream_reward && stream_reward != stream_env;
    if (fork && !event) return fail_arg(CARLE_E_NULL, "carle_env_step_f32_rnd: event is NULL");

    // 64x64 grids with an aligned action window: ONE kernel steps the automaton and evaluates the
    // reward net on the state it still holds in registers (ca_rnd_fused.hip); the epilogue turns
    // the bonus into the reward.  Everything runs on stream_env; stream_reward (if any) is ordered
    // behind it so that reward consumers enqueued there see the result.
    if (h == 64 && w == 64 && (!action || (aw % 8) == 0) && n > 0 && reward) {
        // (a packed input state: the checks see u_out in u_in's place -- the fp32 input does not exist)
        if (int e = rnd_step_checks("carle_env_step_f32_rnd", bits_in ? u_out + (size_t)n * 4096 : u_in, action, u_out, n, h, w,
                                    ah, aw, rule, slot, workspace))
            return e;
        const size_t bytes = (size_t)n * h * w * sizeof(float);
        const bool merge = (publish_flags & CARLE_STEP_MERGE_MOMENTS) != 0;
        if ((publish_flags & CARLE_STEP_DEFERRED) && n > 16 && knobs().two_launch_step <= 0 && aligned16(status) &&
            (!merge || (moment_partials && aligned16(moment_partials)))) {
            // ONE launch: master reset, reward and (as partial sums) the reward moments inside the fused kernel
            RndEnvArgs ea;
            ea.reward = reward;
            ea.reward_scale = reward_scale;
            ea.publish = (publish_flags & CARLE_STEP_PUBLISH_FLAGS) ? 1 : 0;
            ea.merge_moments = merge ? 1 : 0;
            ea.partials = merge ? moment_partials + (size_t)slot * 2 * (size_t)rnd64_blocks(n) : nullptr;
            ea.bits_out = bits_out;
            if (bits_out && bits_written) *bits_written = 1;
            if (int e = ca_step_rnd64(bits_in ? reinterpret_cast<const float *>(bits_in) : u_in, action, u_out, n, aw, rule, status,
                                      slot, pred_params, rand_params, workspace, bonus, tables_valid,
                                      static_cast<hipStream_t>(stream_env), &ea, bits_in != nullptr))
                return e;
            return fork ? carle_stream_fork(event, stream_env, stream_reward) : 0;
        }
        if (bits_in) return fail_arg(CARLE_E_DIMS, "carle_env_step_f32_rnd_packed: no one-launch step for these arguments "
                                                   "(n > 16, CARLE_STEP_DEFERRED, aligned status / moment_partials)");
        if (int e = ca_step_rnd64(u_in, action, u_out, n, aw, rule,
                                  (publish_flags & CARLE_STEP_PUBLISH_FLAGS) ? status : nullptr, slot,
                                  pred_params, rand_params, workspace, bonus, tables_valid,
                                  static_cast<hipStream_t>(stream_env), nullptr))
            return e;
        if (int e = ca_finish_impl(u_out, (int64_t)bytes, reward, n, status, slot, action != nullptr, nullptr, 0,
                                   bonus, reward_scale, publish_flags & CARLE_STEP_MERGE_MOMENTS, stream_env))
            return e;
        return fork ? carle_stream_fork(event, stream_env, stream_reward) : 0;
    }

    // other geometries: step + epilogue on stream_env, then the bonus kernels (reward += scale *
    // bonus fused into the last one) on stream_reward, ordered behind the step through `event`
    // (the bonus is added after the epilogue here: CARLE_STEP_MERGE_MOMENTS does not apply)
    if (int e = carle_ca_step_f32(u_in, action, u_out, n, h, w, ah, aw, rule,
                                  (publish_flags & CARLE_STEP_PUBLISH_FLAGS) ? status : nullptr, slot, stream_env))
        return e;
    if (int e = ca_finish_impl(u_out, (int64_t)n * h * w * (int64_t)sizeof(float), reward, n, status, slot,
                               action != nullptr, nullptr, 0, nullptr, 0.0f, 0, stream_env))
        return e;
    void *rs = stream_env;
    if (fork) {
        if (int e = carle_stream_fork(event, stream_env, stream_reward)) return e;
        rs = stream_reward;
    }
    return carle_rnd2d_bonus_f32(u_out, n, h, w, pred_params, rand_params, workspace, bonus, nullptr,
                                 nullptr, reward, reward_scale, net_flags | CARLE_NET_OBS_BINARY, rs);
}

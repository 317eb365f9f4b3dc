// reduce.hip -- per-instance (weighted) sums over a grid: the reductions behind the hand-coded
// reward wrappers of the reference: live-cell counts (mcl.py:769,832), masked
// sums (CornerBonus, mcl.py:222-223), centre-of-mass numerators (SpeedDetector, mcl.py:772-775) and
// per-instance action sums (ParsimonyBonus, mcl.py:102-103).  One pass over the grid computes the
// plain sum and up to three plane-weighted sums; fp64 accumulation, deterministic order.
#include "common.hpp"

namespace carle {

// One workgroup per instance.  WM = which weight planes exist (bit j = w_j): the loads are unconditional in every
// instantiation -- with `if (w0)` inside the loop every load was followed by its own s_waitcnt, four dependent
// memory round trips per element (round 2, seen in the disassembly).  VEC: float4 accesses (elems % 4 == 0 and
// 16-byte aligned planes), four independent quads per thread and iteration in flight.
// (body as a device function: the stand-alone kernel and speed_puffer_kernel, which finishes both wrappers behind it.
// COHERENT: the four sums leave as agent-scope stores -- a workgroup on another XCD reads them in the same launch.)
template <int WM, bool VEC, bool COHERENT>
__device__ __forceinline__ void instance_sums_body(double (&part)[4][16], const float *__restrict__ x, int64_t elems,
                                                   const float *__restrict__ w0, const float *__restrict__ w1,
                                                   const float *__restrict__ w2, float *__restrict__ out, const int64_t g) {
    const float *gx = x + g * elems;
    double s = 0.0, a = 0.0, b = 0.0, c = 0.0;
    if (VEC) {
        const int64_t quads = elems >> 2;
        const float4 *x4 = reinterpret_cast<const float4 *>(gx);
        const float4 *a4 = reinterpret_cast<const float4 *>(w0), *b4 = reinterpret_cast<const float4 *>(w1),
                     *c4 = reinterpret_cast<const float4 *>(w2);
        auto acc = [&](const float4 &v, const float4 &wa, const float4 &wb, const float4 &wc) {
            s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
            if (WM & 1) a = fma((double)v.x, (double)wa.x, fma((double)v.y, (double)wa.y, fma((double)v.z, (double)wa.z, fma((double)v.w, (double)wa.w, a))));
            if (WM & 2) b = fma((double)v.x, (double)wb.x, fma((double)v.y, (double)wb.y, fma((double)v.z, (double)wb.z, fma((double)v.w, (double)wb.w, b))));
            if (WM & 4) c = fma((double)v.x, (double)wc.x, fma((double)v.y, (double)wc.y, fma((double)v.z, (double)wc.z, fma((double)v.w, (double)wc.w, c))));
        };
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        const int nt = blockDim.x;                         // 256 (small instances) or 1024
        for (int64_t i = threadIdx.x; i < quads; i += 4 * nt) {
            float4 v[4], wa[4], wb[4], wc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t j = i + (int64_t)nt * u;
                const int64_t jc = j < quads ? j : i;              // unconditional, clamped; dropped below
                v[u] = x4[jc];
                wa[u] = (WM & 1) ? a4[jc] : zero;
                wb[u] = (WM & 2) ? b4[jc] : zero;
                wc[u] = (WM & 4) ? c4[jc] : zero;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i + (int64_t)nt * u < quads) acc(v[u], wa[u], wb[u], wc[u]);
        }
    } else {
        for (int64_t i = threadIdx.x; i < elems; i += blockDim.x) {
            const float xv = gx[i];
            const float fa = (WM & 1) ? w0[i] : 0.f, fb = (WM & 2) ? w1[i] : 0.f, fc = (WM & 4) ? w2[i] : 0.f;
            const double v = (double)xv;
            s += v;
            if (WM & 1) a = fma(v, (double)fa, a);
            if (WM & 2) b = fma(v, (double)fb, b);
            if (WM & 4) c = fma(v, (double)fc, c);
        }
    }
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) {
        s += __shfl_xor(s, k, 64); a += __shfl_xor(a, k, 64);
        b += __shfl_xor(b, k, 64); c += __shfl_xor(c, k, 64);
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { part[0][wv] = s; part[1][wv] = a; part[2][wv] = b; part[3][wv] = c; }
    __syncthreads();
    if (threadIdx.x < 4) {
        double t = 0.0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += part[threadIdx.x][k];
        if (COHERENT) __hip_atomic_store(out + g * 4 + threadIdx.x, (float)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else out[g * 4 + threadIdx.x] = (float)t;
    }
}

template <int WM, bool VEC>
__global__ __launch_bounds__(1024) void instance_sums_kernel(const float *__restrict__ x, int64_t elems,
                                                             const float *__restrict__ w0,
                                                             const float *__restrict__ w1,
                                                             const float *__restrict__ w2,
                                                             float *__restrict__ out) {
    __shared__ double part[4][16];
    instance_sums_body<WM, VEC, false>(part, x, elems, w0, w1, w2, out, (int64_t)blockIdx.x);
}


template <int WM>
static void launch_instance_sums(bool vec, unsigned n, hipStream_t st, const float *x, int64_t elems, const float *w0,
                                 const float *w1, const float *w2, float *out) {
    // small instances (64x64: one unrolled iteration of 256 threads): the 16-wave reduction would dominate
    const unsigned nt = elems <= 16384 ? 256u : 1024u;
    if (vec) hipLaunchKernelGGL((instance_sums_kernel<WM, true>), dim3(n), dim3(nt), 0, st, x, elems, w0, w1, w2, out);
    else hipLaunchKernelGGL((instance_sums_kernel<WM, false>), dim3(n), dim3(nt), 0, st, x, elems, w0, w1, w2, out);
}

// ---------------------------------------------------------------------------------------------------------------
// SpeedDetector.step after the reduction (carle/mcl.py:769-797), one single-block launch instead of ~8 tensor ops:
//   com_h = sum(obs * w_h) / (live + 1e-7), com_w likewise;  velocity = previous com - com;
//   speed = sqrt(sum over BOTH axes and ALL instances of velocity^2)  (batch-coupled, mcl.py:789);  reward += speed
// sums4 (n,4): [sum obs, sum obs*w_h, sum obs*w_w, -] of carle_instance_sums_f32; live4 (nullable): the sums of the
// inner universe where it is not the observation.  state[0] = 1 once a centre of mass has been stored (device side:
// the launch is identical every step, so a HIP graph can replay it).  fp32 quotients as the reference; the sum of
// squares is accumulated in fp64 in a fixed order.
// ---------------------------------------------------------------------------------------------------------------
// (bodies as device functions: the first 256 threads of the workgroup work, every thread takes the barriers; COHERENT:
// sums4 was stored by other workgroups of the same launch -> agent-scope loads)
template <bool COHERENT>
__device__ __forceinline__ float sums_ld(const float *p) {
    return COHERENT ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}

template <bool COHERENT>
__device__ __forceinline__ void speed_update_body(double (&part)[4], float &s_speed, const float *__restrict__ sums4,
                                                  const float *__restrict__ live4, int64_t n, float *__restrict__ com,
                                                  int32_t *__restrict__ state, float *__restrict__ velocity,
                                                  float *__restrict__ speed_out, float *__restrict__ live_out,
                                                  float *__restrict__ reward) {
    const int tid = threadIdx.x;
    const bool has_prev = state[0] != 0;
    double sq = 0.0;
    if (tid < 256)
        for (int64_t i = tid; i < n; i += 256) {
            const float s0 = sums_ld<COHERENT>(sums4 + 4 * i), s1 = sums_ld<COHERENT>(sums4 + 4 * i + 1),
                        s2 = sums_ld<COHERENT>(sums4 + 4 * i + 2);
            const float live = live4 ? live4[4 * i] : s0;
            const float ch = s1 / (live + 1e-7f), cw = s2 / (live + 1e-7f);
            if (has_prev) {
                const float vh = com[i] - ch, vw = com[n + i] - cw;
                velocity[i] = vh; velocity[n + i] = vw;
                sq = fma((double)vh, (double)vh, fma((double)vw, (double)vw, sq));
            }
            com[i] = ch; com[n + i] = cw;
            live_out[i] = live;
        }
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) sq += __shfl_xor(sq, k, 64);
    if (tid < 256 && (tid & 63) == 0) part[tid >> 6] = sq;
    __syncthreads();
    if (tid == 0) {
        const float sp = (float)sqrt((part[0] + part[1]) + (part[2] + part[3]));
        s_speed = sp;
        if (has_prev) speed_out[0] = sp;
        state[0] = 1;
    }
    __syncthreads();
    if (has_prev && reward && tid < 256) {
        const float sp = s_speed;
        for (int64_t i = tid; i < n; i += 256) reward[i] += sp;                      // mcl.py:795
    }
}

__global__ __launch_bounds__(256) void speed_update_kernel(const float *__restrict__ sums4, const float *__restrict__ live4,
                                                           int64_t n, float *__restrict__ com, int32_t *__restrict__ state,
                                                           float *__restrict__ velocity, float *__restrict__ speed_out,
                                                           float *__restrict__ live_out, float *__restrict__ reward) {
    __shared__ double part[4];
    __shared__ float s_speed;
    speed_update_body<false>(part, s_speed, sums4, live4, n, com, state, velocity, speed_out, live_out, reward);
}

// PufferDetector.step after the reduction (carle/mcl.py:832-851) without its two host round trips: the history of
// live-cell totals is a device ring (rstate = [head, count], cap = growth_threshold + 1 slots), the "no action this
// step" predicate is the flag the step kernel left in the status block (any_flag; NULL = no action).
template <bool COHERENT>
__device__ __forceinline__ void puffer_update_body(double (&part)[4], int &s_bonus, const float *__restrict__ sums4, int64_t n,
                                                   const int32_t *__restrict__ any_flag, float *__restrict__ ring,
                                                   int32_t *__restrict__ rstate, int cap, float *__restrict__ live_total,
                                                   float *__restrict__ reward) {
    const int tid = threadIdx.x;
    double s = 0.0;
    if (tid < 256)
        for (int64_t i = tid; i < n; i += 256) s += (double)sums_ld<COHERENT>(sums4 + 4 * i);
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) s += __shfl_xor(s, k, 64);
    if (tid < 256 && (tid & 63) == 0) part[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        const float total = (float)((part[0] + part[1]) + (part[2] + part[3]));     // torch.sum(universe), mcl.py:832
        live_total[0] = total;
        int head = rstate[0], count = rstate[1], bonus = 0;
        const bool any = any_flag != nullptr && any_flag[0] != 0;
        if (!any) {
            ring[(head + count) % cap] = total;                                      // cells.append
            count += 1;
            if (count > cap - 1) {                                                   // len(cells) > growth_threshold
                const float slope = total - ring[head];                              // cells[-1] - cells[0]
                head = (head + 1) % cap;                                             // cells.pop(0)
                count -= 1;
                bonus = slope > 0.01f ? 1 : 0;
            }
        } else {
            head = 0; count = 0;                                                     // cells = []
        }
        rstate[0] = head; rstate[1] = count;
        s_bonus = bonus;
    }
    __syncthreads();
    if (s_bonus && reward && tid < 256)
        for (int64_t i = tid; i < n; i += 256) reward[i] += 1.0f;                    // mcl.py:843
}

__global__ __launch_bounds__(256) void puffer_update_kernel(const float *__restrict__ sums4, int64_t n,
                                                            const int32_t *__restrict__ any_flag, float *__restrict__ ring,
                                                            int32_t *__restrict__ rstate, int cap,
                                                            float *__restrict__ live_total, float *__restrict__ reward) {
    __shared__ double part[4];
    __shared__ int s_bonus;
    puffer_update_body<false>(part, s_bonus, sums4, n, any_flag, ring, rstate, cap, live_total, reward);
}

// PufferDetector(SpeedDetector(env)) -- the top of the evaluation stack, evaluation/eval.py:86-87 -- in ONE launch instead
// of three: every workgroup reduces one instance (the two masked centre-of-mass planes, as SpeedDetector asks for them;
// column 0 = the live cells both wrappers use), the workgroup that finishes LAST runs SpeedDetector's update and then
// PufferDetector's (their bodies, their order: the reward gets the speed first, mcl.py:795, then the growth bonus,
// mcl.py:843).  `counter`: one int32, zero before the launch and zero again behind it.
template <bool VEC>
__global__ __launch_bounds__(1024) void speed_puffer_kernel(const float *__restrict__ x, int64_t elems, const float *__restrict__ w0,
                                                            const float *__restrict__ w1, float *__restrict__ sums4,
                                                            int32_t *__restrict__ counter, int64_t n, float *__restrict__ com,
                                                            int32_t *__restrict__ state, float *__restrict__ velocity,
                                                            float *__restrict__ speed_out, float *__restrict__ live_out,
                                                            const int32_t *__restrict__ any_flag, float *__restrict__ ring,
                                                            int32_t *__restrict__ rstate, int cap,
                                                            float *__restrict__ live_total, float *__restrict__ reward) {
    __shared__ double part[4][16];
    __shared__ double upart[4];
    __shared__ float s_speed;
    __shared__ int s_bonus, s_last;
    instance_sums_body<3, VEC, true>(part, x, elems, w0, w1, nullptr, sums4, (int64_t)blockIdx.x);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this instance's sums have been written through (no fence: an
    __syncthreads();                                        // agent-scope fence invalidates the XCD's whole L2)
    if (threadIdx.x == 0)
        s_last = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    __syncthreads();
    if (!s_last) return;                                    // workgroup-uniform
    speed_update_body<true>(upart, s_speed, sums4, nullptr, n, com, state, velocity, speed_out, live_out, reward);
    __syncthreads();
    puffer_update_body<true>(upart, s_bonus, sums4, n, any_flag, ring, rstate, cap, live_total, reward);
    if (threadIdx.x == 0) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------------------------
// The two updates under BATCH SHARDING (carle_amd/dist.py: one process per GPU owns a slice of the instances).  The
// reference's scalars couple the whole batch -- speed = sqrt(sum over ALL instances of velocity^2) (mcl.py:789), the
// live-cell total and the action's truthiness of PufferDetector (mcl.py:832, 835) -- so each update is split around ONE
// all-reduce(SUM) of exch = double[4] = { sum v^2, had a previous centre of mass, live cells, action had a non-zero }:
// `local` does the per-instance work of this shard and leaves its four terms, the caller reduces them over the ranks,
// `apply` finishes with the global values (same arithmetic as speed_update_body / puffer_update_body: fp64 sums, one
// rounding to fp32).  what: bit 0 = SpeedDetector's part, bit 1 = PufferDetector's.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wrapper_scalars_local_kernel(const float *__restrict__ sums4, const float *__restrict__ live4,
                                                                    int64_t n, float *__restrict__ com, int32_t *__restrict__ state,
                                                                    float *__restrict__ velocity, float *__restrict__ live_out,
                                                                    const int32_t *__restrict__ any_flag, double *__restrict__ exch,
                                                                    int what) {
    __shared__ double part[2][4];
    const int tid = threadIdx.x;
    const bool speed = (what & 1) != 0, puffer = (what & 2) != 0;
    const bool has_prev = speed && state[0] != 0;
    double sq = 0.0, cells = 0.0;
    for (int64_t i = tid; i < n; i += 256) {
        const float s0 = sums4[4 * i];
        const float live = live4 ? live4[4 * i] : s0;
        if (speed) {
            const float s1 = sums4[4 * i + 1], s2 = sums4[4 * i + 2];
            const float ch = s1 / (live + 1e-7f), cw = s2 / (live + 1e-7f);
            if (has_prev) {
                const float vh = com[i] - ch, vw = com[n + i] - cw;
                velocity[i] = vh; velocity[n + i] = vw;
                sq = fma((double)vh, (double)vh, fma((double)vw, (double)vw, sq));
            }
            com[i] = ch; com[n + i] = cw;
            live_out[i] = live;
        }
        cells += (double)live;
    }
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) { sq += __shfl_xor(sq, k, 64); cells += __shfl_xor(cells, k, 64); }
    if ((tid & 63) == 0) { part[0][tid >> 6] = sq; part[1][tid >> 6] = cells; }
    __syncthreads();                                        // every thread has read state[0] by now
    if (tid == 0) {
        exch[0] = speed ? (part[0][0] + part[0][1]) + (part[0][2] + part[0][3]) : 0.0;
        exch[1] = has_prev ? 1.0 : 0.0;
        exch[2] = puffer ? (part[1][0] + part[1][1]) + (part[1][2] + part[1][3]) : 0.0;
        exch[3] = (puffer && any_flag != nullptr && any_flag[0] != 0) ? 1.0 : 0.0;
        if (speed) state[0] = 1;
    }
}

__global__ __launch_bounds__(256) void wrapper_scalars_apply_kernel(const double *__restrict__ exch, int what,
                                                                    float *__restrict__ speed_out, float *__restrict__ ring,
                                                                    int32_t *__restrict__ rstate, int cap,
                                                                    float *__restrict__ live_total, float *__restrict__ reward,
                                                                    int64_t n) {
    __shared__ int s_bonus;
    const int tid = threadIdx.x;
    const bool has_prev = (what & 1) && exch[1] != 0.0;
    const float sp = has_prev ? (float)sqrt(exch[0]) : 0.0f;           // mcl.py:789 over the whole batch
    if (tid == 0) {
        if (has_prev) speed_out[0] = sp;
        int bonus = 0;
        if (what & 2) {
            const float total = (float)exch[2];                         // torch.sum(universe) of the whole batch, mcl.py:832
            live_total[0] = total;
            int head = rstate[0], count = rstate[1];
            if (exch[3] == 0.0) {                                       // no rank's action had a non-zero element, mcl.py:835
                ring[(head + count) % cap] = total;
                count += 1;
                if (count > cap - 1) {
                    const float slope = total - ring[head];
                    head = (head + 1) % cap;
                    count -= 1;
                    bonus = slope > 0.01f ? 1 : 0;
                }
            } else {
                head = 0; count = 0;
            }
            rstate[0] = head; rstate[1] = count;
        }
        s_bonus = bonus;
    }
    __syncthreads();
    if (reward && (has_prev || s_bonus))
        for (int64_t i = tid; i < n; i += 256) {
            float r = reward[i];
            if (has_prev) r += sp;                                      // mcl.py:795
            if (s_bonus) r += 1.0f;                                     // mcl.py:843
            reward[i] = r;
        }
}

// PredictionBonus / SurpriseBonus, behind the bonus (carle/mcl.py:686-694, 715-723): b_i = offset + sign * err_i (0.1 - err
// or err), b_i *= 0 where instance i has no live cell (sums4[4i] == 0: the reference's per-instance Python loop with a host
// synchronisation each), reward_i += scale * b_i -- product and sum rounded separately, as torch does.  live_total: the
// step's obs.sum() (mcl.py:697, 726), one float.
__global__ __launch_bounds__(256) void bonus_nonempty_kernel(float *__restrict__ reward, const float *__restrict__ err,
                                                             const float *__restrict__ sums4, float offset, float sign,
                                                             float scale, int64_t n, float *__restrict__ live_total) {
    __shared__ float part[4];
    float live = 0.0f;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const float cells = sums4[4 * i];
        live += cells;
        float b = offset + sign * err[i];             // sign = +-1: the product is exact, one rounding either way
        if (cells == 0.0f) b = b * 0.0f;
        reward[i] = reward_accumulate(reward[i], scale, b);   // no contraction into an fma (common.hpp)
    }
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) live += __shfl_xor(live, sft, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = live;
    __syncthreads();
    if (threadIdx.x == 0 && live_total) *live_total = (part[0] + part[1]) + (part[2] + part[3]);
}

}  // namespace carle

using namespace carle;

extern "C" int carle_speed_update_f32(const float *sums4, const float *live4, int64_t n, float *com, int32_t *state,
                                      float *velocity, float *speed, float *live_out, float *reward, void *stream) {
    if (!sums4 || !com || !state || !velocity || !speed || !live_out)
        return fail_arg(CARLE_E_NULL, "carle_speed_update_f32: NULL pointer");
    if (n < 0) return fail_arg(CARLE_E_DIMS, "carle_speed_update_f32: negative size");
    if (n == 0) return 0;
    hipLaunchKernelGGL(speed_update_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), sums4, live4, n, com,
                       state, velocity, speed, live_out, reward);
    return check_launch("carle_speed_update_f32");
}

extern "C" int carle_puffer_update_f32(const float *sums4, int64_t n, const int32_t *any_flag, float *ring,
                                       int32_t *ring_state, int cap, float *live_total, float *reward, void *stream) {
    if (!sums4 || !ring || !ring_state || !live_total)
        return fail_arg(CARLE_E_NULL, "carle_puffer_update_f32: NULL pointer");
    if (n < 0 || cap < 2) return fail_arg(CARLE_E_DIMS, "carle_puffer_update_f32: n >= 0 and cap >= 2 required");
    hipLaunchKernelGGL(puffer_update_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), sums4, n, any_flag,
                       ring, ring_state, cap, live_total, reward);
    return check_launch("carle_puffer_update_f32");
}

extern "C" int carle_wrapper_scalars_local_f32(const float *sums4, const float *live4, int64_t n, float *com, int32_t *state,
                                               float *velocity, float *live_out, const int32_t *any_flag, double *exch,
                                               int what, void *stream) {
    if (!sums4 || !exch) return fail_arg(CARLE_E_NULL, "carle_wrapper_scalars_local_f32: NULL pointer");
    if ((what & 1) && (!com || !state || !velocity || !live_out))
        return fail_arg(CARLE_E_NULL, "carle_wrapper_scalars_local_f32: SpeedDetector's buffers are NULL");
    if (n < 0 || (what & ~3) || what == 0) return fail_arg(CARLE_E_DIMS, "carle_wrapper_scalars_local_f32: n >= 0 and what in 1..3 required");
    hipLaunchKernelGGL(wrapper_scalars_local_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), sums4, live4, n,
                       com, state, velocity, live_out, any_flag, exch, what);
    return check_launch("carle_wrapper_scalars_local_f32");
}

extern "C" int carle_wrapper_scalars_apply_f32(const double *exch, int what, float *speed, float *ring, int32_t *ring_state,
                                               int cap, float *live_total, float *reward, int64_t n, void *stream) {
    if (!exch) return fail_arg(CARLE_E_NULL, "carle_wrapper_scalars_apply_f32: NULL pointer");
    if ((what & 1) && !speed) return fail_arg(CARLE_E_NULL, "carle_wrapper_scalars_apply_f32: speed is NULL");
    if ((what & 2) && (!ring || !ring_state || !live_total))
        return fail_arg(CARLE_E_NULL, "carle_wrapper_scalars_apply_f32: PufferDetector's buffers are NULL");
    if (n < 0 || (what & ~3) || what == 0 || ((what & 2) && cap < 2))
        return fail_arg(CARLE_E_DIMS, "carle_wrapper_scalars_apply_f32: n >= 0, what in 1..3 and cap >= 2 required");
    hipLaunchKernelGGL(wrapper_scalars_apply_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), exch, what, speed,
                       ring, ring_state, cap, live_total, reward, n);
    return check_launch("carle_wrapper_scalars_apply_f32");
}

extern "C" int carle_speed_puffer_f32(const float *obs, int64_t n, int64_t elems, const float *w_h, const float *w_w,
                                      float *sums4, int32_t *counter, float *com, int32_t *state, float *velocity,
                                      float *speed, float *live_out, const int32_t *any_flag, float *ring,
                                      int32_t *ring_state, int cap, float *live_total, float *reward, void *stream) {
    if (!obs || !w_h || !w_w || !sums4 || !counter || !com || !state || !velocity || !speed || !live_out || !ring ||
        !ring_state || !live_total)
        return fail_arg(CARLE_E_NULL, "carle_speed_puffer_f32: NULL pointer");
    if (n < 0 || elems < 0 || cap < 2) return fail_arg(CARLE_E_DIMS, "carle_speed_puffer_f32: n, elems >= 0 and cap >= 2 required");
    if (n == 0) return 0;
    if (n > 0x7FFFFFFFll) return fail_arg(CARLE_E_DIMS, "carle_speed_puffer_f32: batch too large");
    auto al16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    const bool vec = elems % 4 == 0 && al16(obs) && al16(w_h) && al16(w_w);
    const unsigned nt = elems <= 16384 ? 256u : 1024u;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (vec)
        hipLaunchKernelGGL(speed_puffer_kernel<true>, dim3((unsigned)n), dim3(nt), 0, st, obs, elems, w_h, w_w, sums4, counter, n, com,
                           state, velocity, speed, live_out, any_flag, ring, ring_state, cap, live_total, reward);
    else
        hipLaunchKernelGGL(speed_puffer_kernel<false>, dim3((unsigned)n), dim3(nt), 0, st, obs, elems, w_h, w_w, sums4, counter, n, com,
                           state, velocity, speed, live_out, any_flag, ring, ring_state, cap, live_total, reward);
    return check_launch("carle_speed_puffer_f32");
}

extern "C" int carle_bonus_nonempty_f32(float *reward, const float *err, const float *sums4, float offset, float sign,
                                        float scale, int64_t n, float *live_total, void *stream) {
    if (!reward || !err || !sums4) return fail_arg(CARLE_E_NULL, "carle_bonus_nonempty_f32: NULL pointer");
    if (n < 0) return fail_arg(CARLE_E_DIMS, "carle_bonus_nonempty_f32: negative size");
    if (n == 0) return 0;
    hipLaunchKernelGGL(bonus_nonempty_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), reward, err, sums4,
                       offset, sign, scale, n, live_total);
    return check_launch("carle_bonus_nonempty_f32");
}

extern "C" int carle_instance_sums_f32(const float *x, int64_t n, int64_t elems, const float *w0,
                                       const float *w1, const float *w2, float *out4, void *stream) {
    if (!x || !out4) return fail_arg(CARLE_E_NULL, "carle_instance_sums_f32: NULL pointer");
    if (n < 0 || elems < 0) return fail_arg(CARLE_E_DIMS, "carle_instance_sums_f32: negative size");
    if (n == 0) return 0;
    if (n > 0x7FFFFFFFll) return fail_arg(CARLE_E_DIMS, "carle_instance_sums_f32: batch too large");
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto al16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    const bool vec = elems % 4 == 0 && al16(x) && (!w0 || al16(w0)) && (!w1 || al16(w1)) && (!w2 || al16(w2));
    const int wm = (w0 ? 1 : 0) | (w1 ? 2 : 0) | (w2 ? 4 : 0);
    const unsigned nb = (unsigned)n;
    switch (wm) {
        case 0: launch_instance_sums<0>(vec, nb, st, x, elems, w0, w1, w2, out4); break;
        case 1: launch_instance_sums<1>(vec, nb, st, x, elems, w0, w1, w2, out4); break;
        case 2: launch_instance_sums<2>(vec, nb, st, x, elems, w0, w1, w2, out4); break;
        case 3: launch_instance_sums<3>(vec, nb, st, x, elems, w0, w1, w2, out4); break;
        case 4: launch_instance_sums<4>(vec, nb, st, x, elems, w0, w1, w2, out4); break;
        case 5: launch_instance_sums<5>(vec, nb, st, x, elems, w0, w1, w2, out4); break;
        case 6: launch_instance_sums<6>(vec, nb, st, x, elems, w0, w1, w2, out4); break;
        default: launch_instance_sums<7>(vec, nb, st, x, elems, w0, w1, w2, out4); break;
    }
    return check_launch("carle_instance_sums_f32");
}

#!/usr/bin/env python3
"""bench.py -- env-steps/s of the CARLE hot path on MI355X (BASELINE.json metric).

A "step" is one CARLE.step(action) over the whole batch: the fused HIP CA kernel
(toggle + 3x3 torus count + B/S rule) plus the epilogue kernel, driven through the drop-in
Python surface (carle_amd.env.CARLE).  Workload at N=1: BASELINE.json configs[1] -- B3/S23,
N=4096, 64x64 grids, fp32 drop-in format, a fresh device-resident Bernoulli(0.1) 64x64 action
every step (ring of action buffers larger than the 256 MiB Infinity Cache, so state and action
really stream from HBM).

Multi-GPU (--gpus N > 1): one process per GPU, every rank owns --instances universes (weak
scaling, no data-path collective: universes are independent); the only exchange is
the 24-byte all-gather of the reward moments every --stats-every steps (RCCL).  Two launch forms:
  * under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment):
    this process IS one rank;
  * plain `python bench.py --gpus N`: this process is only a launcher -- it spawns N fresh rank
    processes (before anything here has touched the GPU), waits for them and exits with the first
    non-zero code.  Rank 0 prints the JSON line.
--gpus must equal WORLD_SIZE, and the node must expose N devices: anything else exits non-zero
instead of silently measuring fewer GPUs.

Prints ONE JSON line on rank 0 (contract in the task description), with two extra objects:
  roofline      algorithmic bytes of the config's dominant kernel / its average launch duration
                measured here with HIP events on the launch stream, against 8 TB/s
  cpu_baseline  oracle/torch_port.py (a port of the reference's per-step ATen op sequence; the
                reference itself cannot travel) timed on this host's cores, bounded sample
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW"
FP32_PEAK = 157.3e12       # FLOP/s, same guide, "Peak FP32 (vector)"
MIN_TIMED_MS = 5.0         # below this the timed region is too short to quote (warned in the JSON)
MIN_BLOCKS, MAX_BLOCKS, MIN_TOTAL_MS = 5, 201, 50.0      # repeats of the timed block (odd cap: a true median)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--instances", type=int, default=4096, help="instances PER GPU")
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--rule", default="B3/S23")
    ap.add_argument("--format", default="f32", choices=["f32", "u8", "b1"],
                    help="grid storage: f32 = the reference's drop-in format (headline), "
                         "u8 = uint8 state and actions, b1 = bit-packed state and actions "
                         "(secondary configurations)")
    ap.add_argument("--action-ring", type=int, default=6,
                    help="number of distinct device action buffers cycled through")
    ap.add_argument("--agent", default="ring", choices=["ring", "random", "fused"],
                    help="action source: ring = pre-generated device tensors (BASELINE definition); "
                         "random = device-side RandomAgent tensor per step; fused = RandomAgent "
                         "toggles drawn inside the step kernel (no action tensor)")
    ap.add_argument("--rnd", action="store_true",
                    help="configs[3]: also run the RND2D intrinsic-reward forward (eval mode) on "
                         "every observation and feed ITS reward to the running statistics")
    ap.add_argument("--ae", action="store_true", help="additionally the AE2D forward")
    ap.add_argument("--train", action="store_true",
                    help="reward nets in training mode: every step also accumulates the predictor "
                         "gradient (fused forward+backward kernels), Adam update every 64 steps "
                         "(carle/train_mcl.py's per-step work)")
    ap.add_argument("--async-reward", action="store_true",
                    help="run the reward consumers (a stacked AE2D / training kernels, reward statistics) on a second "
                         "HIP stream beside the next CA step instead of on the CA stream (measured slower since the "
                         "kernels got faster: the cross-stream events cost more than they hide)")
    ap.add_argument("--sync-reward", action="store_true", help="(default since late round 2; kept for old command lines)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1")
    ap.add_argument("--per-rank-predictors", action="store_true",
                    help="--train with --gpus > 1: every rank trains its own predictor on its shard instead of the "
                         "data-parallel predictor (carle_amd.dist.data_parallel: gradient all-reduce per Adam step)")
    ap.add_argument("--share-gpus", action="store_true",
                    help="rehearsal only: let several ranks share a device when the node has fewer "
                         "than --gpus (needs --backend gloo; RCCL refuses two ranks on one device)")
    ap.add_argument("--launch-check", action="store_true",
                    help="ranks only join the process group, all-gather (rank, local_rank, device) "
                         "and rank 0 prints it as JSON: tests the launcher without a GPU")
    ap.add_argument("--stats-every", type=int, default=64,
                    help="steps between reward-statistics all-gathers (reporting interval)")
    ap.add_argument("--separate-stats-kernel", action="store_true",
                    help="reduce the rewards with the stand-alone moments kernel every step instead of "
                         "inside the step epilogue (A/B switch)")
    ap.add_argument("--replay", default="none", choices=["none", "calls", "graph"],
                    help="replay the step instead of calling env.step: 'calls' = the recorded library calls of a step "
                         "issued again (carle_amd.graphs.StepGraph backend='calls'), 'graph' = a HIP graph; both for "
                         "forward-only stacks under the action ring.  Off by default: the headline is the eager surface")
    ap.add_argument("--repeats", type=int, default=0,
                    help="number of timed blocks of --steps steps (default: >= 5, until >= 50 ms are timed)")
    ap.add_argument("--single-rank-group", action="store_true",
                    help="--gpus 1 only: join a process group of ONE rank (backend as --backend, nccl = RCCL) and send every "
                         "exchange of the path through it (carle_amd.dist.force_collectives): what a one-GPU box can execute "
                         "of the multi-GPU path")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="headline run only: skip the short timed blocks of the other BASELINE configs (key \"configs\")")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` outside torch.distributed.run
# ------------------------------------------------------------------------------------------------

def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """Spawns one fresh process per GPU and waits for them.  Runs BEFORE any HIP call in this
    process (torch.cuda.device_count() does not initialise the GPU) and never exec()s: the ranks
    are ordinary children, rank 0 inherits stdout for the JSON line."""
    n = args.gpus
    have = torch.cuda.device_count()
    if not args.launch_check and have < n and not args.share_gpus:
        print(f"bench.py: --gpus {n} but this node exposes {have} HIP device(s); refusing to run "
              f"fewer GPUs than asked (use --share-gpus --backend gloo for a rehearsal)", file=sys.stderr)
        return 2
    if args.share_gpus and args.backend == "nccl" and have < n:
        print("bench.py: --share-gpus needs --backend gloo (RCCL refuses two ranks on one device)",
              file=sys.stderr)
        return 2
    port = _free_port()
    procs = []
    for r in range(n):
        # HSA_ENABLE_IPC_MODE_LEGACY=0: this pool's host driver supports dmabuf IPC only; with the legacy mode RCCL's
        # intra-node transport setup fails in hipIpcGetMemHandle ("invalid argument").  The image exports it already;
        # the launcher keeps it for ranks started from a scrubbed environment (an explicit setting wins).
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.setdefault("OMP_NUM_THREADS", "1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    pending = dict(enumerate(procs))
    while pending:
        for r, p in list(pending.items()):
            code = p.poll()
            if code is None:
                continue
            del pending[r]
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr)
                for q in pending.values():
                    q.terminate()               # exactly the children started above
        time.sleep(0.05)
    return rc


# ------------------------------------------------------------------------------------------------
# CPU baseline
# ------------------------------------------------------------------------------------------------

def _host_threads():
    # the threads this process may actually run on (a 1-GPU box exposes a CPU share, not all cores)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:        # cgroup v2 CPU quota ("1600000 100000" = 16 CPUs): more threads only thrash
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, min(cores, int(os.environ.get("CARLE_CPU_THREADS", cores))))


def _time_port(n, size, rule, seconds, min_steps):
    from oracle.torch_port import TorchPortEnv
    b, s = rule
    asize = min(64, size)
    env = TorchPortEnv(n, size, asize, b, s)
    g = torch.Generator().manual_seed(1234)
    env.universe = (torch.rand(n, 1, size, size, generator=g) < 0.5).float()
    acts = [(torch.rand(n, 1, asize, asize, generator=g) <= 0.1).float() for _ in range(2)]
    env.step(acts[0])                                   # warm-up (allocator, thread pool)
    t0 = time.perf_counter()
    k = 0
    while True:
        env.step(acts[k % 2])
        k += 1
        if k >= min_steps and time.perf_counter() - t0 >= seconds:
            break
    dt = time.perf_counter() - t0
    return n * k / dt, k, dt


def cpu_baseline(n, size, rule, seconds):
    """Port of the reference step (oracle/torch_port.py) on the host cores, bounded sample: the
    SAME batch as the GPU leg (N per GPU, capped at 4096: the CPU rate falls with N), at
    least 3 steps; the N=256 rate (the CPU's cache-friendlier batch) is printed beside it."""
    cores = _host_threads()
    torch.set_num_threads(cores)
    n_big = min(n, 4096)
    v_big, k_big, dt_big = _time_port(n_big, size, rule, seconds, 3)
    out = {"value": v_big, "unit": "env-steps/s", "cores": cores, "kind": "port",
           "sample": f"oracle/torch_port.py (reference ATen op sequence), N={n_big} {size}x{size} "
                     f"{k_big} steps in {dt_big:.1f} s, torch {torch.__version__} {cores} threads"}
    if n_big > 256:
        v_small, k_small, dt_small = _time_port(256, size, rule, min(3.0, seconds), 3)
        out["value_n256"] = v_small
        out["sample_n256"] = f"N=256: {k_small} steps in {dt_small:.1f} s"
    return out


# ------------------------------------------------------------------------------------------------
# workload naming / PMC traffic
# ------------------------------------------------------------------------------------------------

def cfg_name(args, n, size):
    if args.format == "f32" and size == 64 and n == 4096:
        if args.rnd and args.ae:
            return "BASELINE configs[3] + AE2D: the AE2D(RND2D(CARLE)) stack of train_mcl.py:36-40, " + \
                ("training step" if args.train else "forward passes")
        return "BASELINE configs[3] (CA + RND2D forward)" if args.rnd else "BASELINE configs[1]"
    if args.format == "f32" and size == 64 and not (args.rnd or args.ae):
        return f"BASELINE configs[1] shape at N={n} (state arrays beyond the Infinity Cache)" \
            if n * size * size * 4 > (256 << 20) else "BASELINE configs[1] shape"
    if args.format == "b1" and size == 512 and n == 4096:
        return "BASELINE configs[4] (bit-packed)"
    if size == 256:
        return "BASELINE configs[2] shape (4096 instances per GPU)"
    return "custom"


def pmc_traffic(kernel, fmt, size, n):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE; tools/pmc_summary.py).  The
    counters cannot be collected from inside this process, so the value comes from
    profiles/pmc_latest.json -- and only when that record names THIS kernel, THIS workload and
    was taken from the kernel sources this library was built from (sha256 of carle_amd/csrc +
    include/); otherwise traffic is null."""
    from carle_amd import _lib
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        recs = json.load(open(path))
    except (OSError, ValueError):
        return {"traffic": None, "traffic_note": "no profiles/pmc_latest.json"}
    if isinstance(recs, dict):
        recs = [recs]
    sha = _lib.source_hash()
    stale = False
    for rec in recs:
        if (rec.get("size") == size and rec.get("instances") == n and rec.get("format", "f32") == fmt
                and rec.get("kernel") and kernel.startswith(rec["kernel"].split("<")[0])):
            if rec.get("csrc_sha256") == sha:
                return {"traffic": rec["traffic_bytes_per_launch"],
                        "traffic_source": rec.get("source", "profiles/pmc_latest.json")}
            stale = True
    return {"traffic": None,
            "traffic_note": "PMC record predates the current kernel sources" if stale
            else "no PMC record for this kernel / workload"}


# ------------------------------------------------------------------------------------------------
# one rank
# ------------------------------------------------------------------------------------------------

def launch_check(rank, world, dev, backend):
    import torch.distributed as dist
    mine = [rank, int(os.environ.get("LOCAL_RANK", "-1")), int(os.environ.get("WORLD_SIZE", "-1")),
            dev.index if dev.type == "cuda" else -1]
    if world > 1:
        t = torch.tensor(mine, dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        rows = [o.tolist() for o in out]
        dist.barrier()
        dist.destroy_process_group()
    else:
        rows = [mine]
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "backend": backend,
                          "ranks": [{"rank": r[0], "local_rank": r[1], "world_size": r[2], "device": r[3]}
                                    for r in rows],
                          "master": f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}"}),
              flush=True)


def time_kernel(launch, reps, dev):
    """Average duration (ms) of `reps` back-to-back launches, HIP events on the launch stream
    (torch's current stream is the stream the ABI is given)."""
    for i in range(4):
        launch(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    e0.record()
    for i in range(reps):
        launch(i)
    e1.record()
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / reps


def dominant_kernels(args, env, wenv, ring, hip, n, size, asize):
    """Candidates for the config's dominant kernel: (name, launch(i), algorithmic bytes per launch,
    FLOP per launch or None).  Each `launch` enqueues that ONE kernel on the current stream."""
    packed, bytes_fmt = args.format == "b1", args.format == "u8"
    state = env.universe_bits if packed else env.universe
    rule = env.rule_mask()
    bufs = [state, torch.empty_like(state)]
    cand = []
    if packed:
        alg = n * (size * size * 2 + asize * asize) // 8
        name = "ca_step_b1_kernel<..., ENV>"
        step_fn = hip.ca_step_b1
    elif bytes_fmt:
        alg = n * (size * size * 2 + asize * asize)
        name = ("ca_step_64_kernel" if size == 64 else "ca_step_wide_kernel") + "<carle::U8Cells, carle::U8Cells, true, ...>"
        step_fn = hip.ca_step_u8
    else:
        alg = n * (size * size * 8 + asize * asize * 4)
        name = ("ca_step_64_kernel" if size == 64 else "ca_step_wide_kernel") + "<carle::F32Cells, carle::F32Cells, true, ...>"
        step_fn = hip.ca_step_f32

    reward = torch.empty(n, 1, dtype=torch.float32, device=state.device)
    # CARLE_STEP_DEFERRED: the step IS one kernel (64x64 beyond the small-batch kernel's 16 instances; 256 / 512: any batch)
    one_launch = args.format in ("f32", "u8") and ((size == 64 and n > 16) or
                                                   (size in (256, 512) and state.numel() * state.element_size() <= 32 << 20))

    # fp32 64x64 beyond 16 instances (round 5): CARLE.step reads its input state from the bit-packed copy its previous launch
    # left (512 B per grid) instead of the fp32 cells (16 KiB) -- the kernel below is that launch, and its algorithmic bytes
    # are what it has to move: the fp32 action in, the fp32 observation out, the packed state in and out
    two_launch_knob = os.environ.get("CARLE_TWO_LAUNCH_STEP", "0") not in ("", "0")     # (A/B knob: no one-launch step, so no copy at 64x64)
    packed_state = (args.format == "f32" and getattr(env, "packed_state", False) and
                    ((size == 64 and one_launch and not two_launch_knob) or size in (256, 512)))
    if packed_state:
        sbits = [hip.pack_f32_b1(bufs[0]), torch.empty(n, 1, size, size // 64, dtype=torch.int64, device=state.device)]
        alg_reference = alg
        alg = n * (size * size * 4 + asize * asize * 4 + 2 * size * size // 8)
        name = ("ca_step_64_packed2_kernel<carle::F32Cells, ...> (= ca_step_64_kernel<..., ENV, BITS, PACKED_IN>, two grids per "
                "wavefront)" if size == 64 and 3072 < n <= 4096 and os.environ.get("CARLE_PACKED_GRIDS", "") != "1" else
                "ca_step_64_kernel<carle::F32Cells, carle::F32Cells, true, ..., ENV, BITS, PACKED_IN>" if size == 64 else
                "ca_step_wide_kernel<carle::F32Cells, carle::F32Cells, true, ..., ENV, PK=2>" if one_launch else
                "ca_step_wide_kernel<carle::F32Cells, carle::F32Cells, true, ..., PK=2> (timed with its 5 us epilogue launch)")
        dominant_kernels.notes = {name: {
            "traffic_model": "fp32 action window in + fp32 observation out (4 B/cell) + bit-packed state in and out "
                             "(2 x 1/8 B/cell): the fp32 state is written every step and not read back",
            "bytes_per_launch_if_the_fp32_state_were_read": alg_reference}}

    def ca(i):
        if packed:            # the kernel CARLEPacked.step launches (one launch for the 64-column window: resident slice etc.)
            hip.env_step_b1(bufs[i & 1], ring[i % len(ring)], bufs[(i & 1) ^ 1], reward, rule, env._status, i & 1, True, True)
        elif packed_state:    # the kernel CARLE.step launches from its second step on
            hip.env_step_f32_packed(sbits[i & 1], ring[i % len(ring)], bufs[(i & 1) ^ 1], reward, rule, env._status, i & 1,
                                    True, True, bits_out=sbits[(i & 1) ^ 1])
        elif one_launch:      # the kernel CARLE.step launches: step + master reset + reward + moments, no epilogue kernel
            (hip.env_step_f32 if args.format == "f32" else hip.env_step_u8)(
                bufs[i & 1], ring[i % len(ring)], bufs[(i & 1) ^ 1], reward, rule, env._status, i & 1, True, True)
        else:
            step_fn(bufs[i & 1], ring[i % len(ring)], bufs[(i & 1) ^ 1], rule, env._status, 0)

    if one_launch and not packed_state:
        name = name.replace("true, ...>", "true, ..., ENV>")
    nets = args.rnd or args.ae
    fused_rnd = args.rnd and not args.train and args.format == "f32" and size == 64
    if fused_rnd:
        # BASELINE configs[3]: the CA step and the RND2D forward are ONE kernel -- and, for more than 16 instances, the
        # whole RND2D.step is that one launch (reward and reward-moment partial sums included)
        rnd = wenv if not args.ae else wenv.env
        pb, rb = rnd._pred_blob.get(state.device), rnd._rand_blob.get(state.device)
        ws, bonus = rnd._ws, rnd._step_bonus
        partials = rnd.__dict__.get("_mom_partials")
        raw = hip._stream(state)

        def ca_rnd(i):
            if packed_state:      # what RND2D.step launches from its second step on: state in as row words (see above)
                hip.env_step_f32_rnd(bufs[i & 1], ring[i % len(ring)], bufs[(i & 1) ^ 1], reward, rule, env._status, i & 1,
                                     True, pb, rb, ws, bonus, 1.0, True, None, raw, None, partials is not None,
                                     deferred=True, partials=partials, bits_out=sbits[(i & 1) ^ 1], bits_in=sbits[i & 1])
            elif n > 16:
                hip.env_step_f32_rnd(bufs[i & 1], ring[i % len(ring)], bufs[(i & 1) ^ 1], reward, rule, env._status, i & 1,
                                     True, pb, rb, ws, bonus, 1.0, True, None, raw, None, partials is not None,
                                     deferred=True, partials=partials)
            else:
                hip.ca_step_f32_rnd(bufs[i & 1], ring[i % len(ring)], bufs[(i & 1) ^ 1], rule, env._status, 0,
                                    pb, rb, ws, bonus, True)
        rnd_flop = 2 * 9 * size * size * 6 + 2 * 9 * (size // 4) ** 2 * 6 + 2 * (size // 8) ** 2 * 16 * 2
        rname = ("ca_step_rnd64_kernel<true, ..., ENV, BITS, PACKED_IN>" if packed_state else
                 "ca_step_rnd64_kernel<true, ..., ENV>" if n > 16 else "ca_step_rnd64_kernel<true>")
        if packed_state:
            dominant_kernels.notes[rname] = dict(dominant_kernels.notes[name])
            dominant_kernels.notes[rname]["bytes_per_launch_if_the_fp32_state_were_read"] = alg_reference + 4 * n
        cand.append((rname, ca_rnd, alg + 4 * n, n * rnd_flop))
    else:
        cand.append((name, ca, alg, None))
    if nets and size == 64 and args.format == "f32":
        obs = state
        grid_bytes = n * (size * size * 4 + 4)
        # AE2D forward per grid: conv1 1->4 @64^2, conv2 4->2 @32^2, convT 2->1 (4 taps/output) @32^2,
        # convT 1->1 @64^2
        ae_flop = 2 * 9 * 4096 * 4 + 2 * 9 * 4 * 2 * 1024 + 2 * 4 * 2 * 1024 + 2 * 4 * 4096
        rnd_flop = 474112
        for w in ([wenv, wenv.env] if args.ae and args.rnd else [wenv]):
            is_ae = w.my_name == "AE2D"
            if args.train:
                bonus = torch.empty(n, dtype=torch.float32, device=obs.device)

                def tr(i, w=w, bonus=bonus):
                    w._fused_train(obs, None, bonus, None)
                cand.append(("ae_train_block64_kernel" if is_ae else "rnd_train_block64_kernel", tr,
                             grid_bytes, 3 * n * (ae_flop if is_ae else rnd_flop)))
            elif is_ae or not fused_rnd:
                bonus = torch.empty(n, dtype=torch.float32, device=obs.device)

                def fw(i, w=w, bonus=bonus):
                    w._hip_forward(obs, False, None, bonus, None)
                cand.append(("ae_block64_kernel" if is_ae else "rnd_block64_kernel", fw, grid_bytes,
                             n * (ae_flop if is_ae else rnd_flop)))
    return cand


def pin_rank_cpus():
    """One process per GPU: give every rank of the node its own contiguous slice of the CPUs this job may use, so
    that eight ranks' Python / HIP-runtime threads do not migrate over each other (the step costs ~14 us of host
    time against ~35 us of GPU time: a descheduled rank is the scaling risk SURVEY 8(e) names).  No-op for one
    rank, with CARLE_NO_PIN=1, or where the affinity call is refused.  Returns the CPU list or None."""
    lw = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    if lw <= 1 or os.environ.get("CARLE_NO_PIN") or not hasattr(os, "sched_setaffinity"):
        return None
    cpus = sorted(os.sched_getaffinity(0))
    per = len(cpus) // lw
    if per < 1:
        return None
    mine = cpus[lr * per:(lr + 1) * per]
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    return mine


def run_rank(args):
    pinned = pin_rank_cpus()
    from carle_amd import dist as cdist
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world != args.gpus:               # checked BEFORE the rendezvous: a mismatch must not hang
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}: refusing to report a "
              f"{env_world}-GPU number as a {args.gpus}-GPU one", file=sys.stderr)
        sys.exit(2)
    if args.single_rank_group:
        if args.gpus != 1:
            print("bench.py: --single-rank-group is for --gpus 1", file=sys.stderr)
            sys.exit(2)
        cdist.force_collectives(True)
    rank, world, dev = cdist.init_from_env(args.backend, share_devices=args.share_gpus,
                                           single_rank_group=args.single_rank_group)
    if args.launch_check:
        launch_check(rank, world, dev, args.backend)
        return
    if dev.type != "cuda":
        print("bench.py: no HIP device visible (carle_amd has no CPU path)", file=sys.stderr)
        sys.exit(2)
    if world > 1 and not args.share_gpus and torch.cuda.device_count() < int(os.environ.get("LOCAL_WORLD_SIZE", world)):
        print(f"bench.py: {world} ranks but only {torch.cuda.device_count()} HIP device(s)", file=sys.stderr)
        sys.exit(2)
    # proof of N ranks for the record: gathered THROUGH the group (a collective: every rank calls it)
    coll = cdist.collective_info(dev)
    out = measure(args, rank, world, dev, pinned)
    if rank == 0:
        out["collective"] = coll
        headline = (args.format == "f32" and args.size == 64 and not (args.rnd or args.ae or args.train)
                    and args.agent == "ring" and args.replay == "none")
        if world == 1 and headline and not args.no_extra_configs:
            # the other BASELINE configs, short blocks with the same estimator: one driver-timed line carries all five
            out["configs"] = extra_configs(args, rank, world, dev, pinned)
        print(json.dumps(out), flush=True)
    if cdist.exchanging() or world > 1:
        import torch.distributed as tdist
        tdist.barrier()
        tdist.destroy_process_group()


EXTRA_CONFIGS = (
    # (label, argument overrides): BASELINE.json configs[2] (one GPU's shard), [3], [4] and the stacked forward of the north star
    # (blocks as long as the headline's: with 64-step blocks the per-block barrier + flush showed as 1.5-2 us per step)
    ("configs[2] shard", dict(size=256, steps=32, warmup=4)),
    ("configs[3] + RND2D forward", dict(rnd=True, steps=256, warmup=16)),
    ("configs[4] bit-packed HighLife", dict(format="b1", size=512, rule="B36/S23", steps=128, warmup=8)),
    ("RND2D + AE2D forward stack (train_mcl.py:36-40)", dict(rnd=True, ae=True, steps=256, warmup=16)),
)


def extra_configs(args, rank, world, dev, pinned):
    import copy
    rows = []
    for label, over in EXTRA_CONFIGS:
        a = copy.copy(args)
        a.instances = 4096
        for k, v in over.items():
            setattr(a, k, v)
        a.repeats = 5
        t0 = time.perf_counter()
        try:
            o = measure(a, rank, world, dev, pinned, light=True)
        except Exception as e:                                   # a failing side config must not cost the headline line
            rows.append({"label": label, "error": f"{type(e).__name__}: {e}"})
            continue
        r = o["roofline"]
        rows.append({"label": label, "workload": o["config"]["workload"], "ms_per_step": o["ms_per_step"],
                     "env_steps_per_s": o["value"], "cell_updates_per_s": o["cell_updates_per_s"], "steps": o["steps"],
                     "repeats": o["timing"]["repeats"], "dtype": o["dtype"],
                     "roofline": {k: r[k] for k in ("kernel", "frac", "avg_launch_us", "achieved", "unit",
                                                    "algorithmic_bytes_per_launch", "launches_timed", "other_kernels_us",
                                                    "traffic") if k in r},
                     "wall_s": round(time.perf_counter() - t0, 2)})
        gc.collect()
        torch.cuda.empty_cache()
    return rows


def measure(args, rank, world, dev, pinned, light=False):
    """One workload: warm-up, timed blocks, roofline of its dominant kernel; returns the JSON object (rank 0) or None.
    ``light``: a side config of the headline run -- fewer roofline launches, no CPU baseline."""
    from carle_amd import dist as cdist
    dist = None
    if world > 1:
        import torch.distributed as dist
    from carle_amd import hip
    from carle_amd.env import CARLE

    n, size = args.instances, args.size
    asize = min(64, size)
    packed = args.format == "b1"
    bytes_fmt = args.format == "u8"
    env = CARLE(width=size, height=size, action_width=asize, action_height=asize, instances=n,
                device=str(dev), storage=args.format)
    env.rules_from_string(args.rule)
    env.reset()
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    env.universe = (torch.rand(n, 1, size, size, device=dev, generator=g) < 0.5).float()
    ga = torch.Generator(device=dev).manual_seed(4321 + rank)
    ring = [(torch.rand(n, 1, asize, asize, device=dev, generator=ga) <= 0.1).float()
            for _ in range(args.action_ring)]
    if packed:                      # actions handed over already packed (1 bit per toggle)
        ring = [hip.pack_f32_b1(a) for a in ring]
    elif bytes_fmt:                 # ... or as uint8 (1 byte per toggle)
        ring = [a.to(torch.uint8) for a in ring]

    # reward statistics shared by all shards: local moments every step (HIP kernel), one 24-B
    # all-gather per rank every --stats-every steps (the only collective of the path)
    stats = cdist.RunningRewardStats(dev, sync_every=args.stats_every)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    wenv = env
    if args.rnd or args.ae:
        from carle_amd.mcl import RND2D, AE2D
        torch.manual_seed(0)                 # construction order predictor -> random (mcl.py:248)
        # RND2D alone is fused into the CA kernel (one stream); a stacked AE2D / the training kernels run on the same
        # stream too unless --async-reward puts them on the reward stream beside the next CA step
        use_rs = (args.ae or args.train) and args.async_reward and not args.sync_reward
        kw = dict(async_reward=use_rs, train_predictor=args.train)
        if args.rnd:
            wenv = RND2D(env, **kw)
        if args.ae:
            wenv = AE2D(wenv, **kw)
        wenv.eval()
        wenv.to(dev)
        if args.train and world > 1 and not args.per_rank_predictors:
            # the reference's loss is the mean over the WHOLE batch: flat-gradient all-reduce before every Adam step
            cdist.data_parallel(wenv, broadcast=False)       # same seed on every rank: the networks already agree
    # Measured (late round 2, N=4096 64x64 / the reference's default 16 x 256x256): one stream 0.0767 / 0.0513 ms/step with
    # RND2D + AE2D forward, 0.171 / 0.088 in training mode; reward stream 0.0810 / 0.098 and 0.169 / 0.111 -- the
    # cross-stream events cost more than the overlap hides, most of all for small batches: one stream is the default
    rs = None
    if (args.ae or args.train) and args.async_reward and not args.sync_reward:
        from carle_amd.streams import RewardStream
        rs = env.__dict__.get("reward_stream")
        if rs is None:
            rs = RewardStream(dev)
            env.__dict__["reward_stream"] = rs

    agent = None
    if args.agent != "ring":
        from carle_amd.agents import RandomAgent
        agent = RandomAgent(action_width=asize, action_height=asize, fused=args.agent == "fused",
                            seed=4321 + rank)

    # the step epilogue reduces the rewards itself where the reward is final at that point (bare
    # CARLE, RND2D fused into the step): no third launch per step
    if not args.separate_stats_kernel:
        stats.attach(wenv)

    plan = None
    if args.replay != "none":
        if agent is not None or args.train or rs is not None:
            raise SystemExit("--replay needs the action ring, no predictor updates and one stream")
        from carle_amd.graphs import StepGraph
        static = ring[0].clone()
        plan = StepGraph(wenv, static, backend=args.replay)

    def one_step(i):
        action = ring[i % len(ring)] if agent is None else agent(env.universe)
        if plan is not None:
            if args.replay == "calls":
                obs, reward, done, info = plan(action)
            else:
                static.copy_(action)
                obs, reward, done, info = plan()
            stats.after_step(wenv, reward)
            return
        obs, reward, done, info = wenv.step(action)
        if rs is None:
            stats.after_step(wenv, reward)
        elif stats._pending_steps + 1 >= stats.sync_every:
            if wenv is env:
                rs.fork(reward)
            with rs.context():               # the periodic all-gather + merge (torch / RCCL ops)
                stats.update(reward)
        else:
            raw = rs.fork(reward) if wenv is env else rs.raw   # wrappers fork themselves
            stats.update(reward, raw)

    def join_rewards():
        if rs is not None:
            rs.join()

    if args.train and args.warmup < 136:
        # the first Adam step of a process loads torch's optimiser kernels (tens of milliseconds, once):
        # it must fall into the warm-up, i.e. the warm-up has to cover two update windows of 64 steps
        args.warmup = 136
    step_no = 0
    for _ in range(args.warmup):
        one_step(step_no)
        step_no += 1
    join_rewards()
    stats.flush()                       # first collective (RCCL setup) stays out of the timing
    # a full collection of the cyclic garbage collector takes 40-90 ms with torch imported (measured:
    # tools/big_n_steps.py, one such pause at step 16 of a bare step loop); collect now and move the survivors
    # to the permanent generation, so that no such pause can fall into a short timed region
    gc.collect()
    gc.freeze()

    # ---- timed region (median of >= 5 repeats).  One BLOCK = exactly --steps steps bracketed by
    # barrier + synchronize on both sides; the blocks repeat until >= MIN_TOTAL_MS have been timed (at least
    # MIN_BLOCKS, at most MAX_BLOCKS), `value` is the MEDIAN block (max over ranks per block).  The periodic
    # all-gather of the reward moments (every --stats-every steps) stays inside the blocks it falls into; the
    # final flush of whatever is pending is timed once, outside (flush_us).
    marks_all = []

    def timed_block():
        nonlocal step_no
        barrier()
        t0 = time.perf_counter()
        marks = [t0]
        for _ in range(args.steps):
            one_step(step_no)
            step_no += 1
            marks.append(time.perf_counter())
        join_rewards()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        marks_all.append(marks)
        return t1 - t0

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    block_dt = [timed_block()]
    first = max_over_ranks(block_dt[0])            # every rank derives the same number of repeats from it
    repeats = args.repeats if args.repeats > 0 else \
        min(MAX_BLOCKS, max(MIN_BLOCKS, int(MIN_TOTAL_MS * 1e-3 / max(first, 1e-9)) + 1))
    repeats |= 1                                   # odd: the median is a block that was actually timed
    while len(block_dt) < repeats:
        block_dt.append(timed_block())
    barrier()
    t0 = time.perf_counter()
    stats.flush()
    torch.cuda.synchronize(dev)
    flush_us = (time.perf_counter() - t0) * 1e6
    barrier()
    per_rank = torch.tensor(block_dt, dtype=torch.float64).reshape(1, -1)
    if dist is not None:                             # [world, repeats]: every rank's block times
        mine = per_rank.to(dev if args.backend == "nccl" else "cpu")
        rows = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(rows, mine)
        per_rank = torch.cat(rows).cpu()
    per_block = per_rank.max(dim=0).values           # a block lasts as long as its slowest rank
    dt = float(per_block.median())                   # torch.median: the lower middle element for even counts
    rank_ms = (per_rank.median(dim=1).values / args.steps * 1e3).tolist()
    pop = float(env.universe.sum().item())      # keeps the result live / sanity
    reward_count = stats.count                  # rewards seen by ALL ranks (the collective's result)

    # ---- roofline of the config's dominant kernel: back-to-back launches of that kernel alone
    warnings = []
    # host-side stalls inside the timed region (enqueue time of a single step far above the median: a garbage
    # collection, an allocator call, a page fault storm): reported, never hidden
    gaps = [(b, i, m[i + 1] - m[i]) for b, m in enumerate(marks_all) for i in range(len(m) - 1)]
    host = sorted(g_[2] for g_ in gaps)
    if host:
        med = host[len(host) // 2]
        slow = [(f"block {b} step {i}", round(d_ * 1e3, 2)) for b, i, d_ in gaps if d_ > max(20 * med, 5e-3)]
        if slow:
            warnings.append(f"host stalls inside the timed region (step, ms): {slow[:8]}; median host time per step "
                            f"{med * 1e6:.1f} us")
    dominant_kernels.notes = {}
    cands = dominant_kernels(args, env, wenv, ring, hip, n, size, asize)
    timed = []
    for name, launch, alg, flop in cands:
        reps = 64 if light else 256
        k_ms = time_kernel(launch, reps, dev)
        while k_ms * reps < (0.8 if light else 2) * MIN_TIMED_MS and reps < 8192:     # >= 10 ms of launches per estimate (side configs: 4)
            reps *= 4
            k_ms = time_kernel(launch, reps, dev)
        timed.append((k_ms, name, alg, flop, reps))
    k_ms, kname, alg_bytes, flop, reps = max(timed)
    achieved = alg_bytes / (k_ms * 1e-3)
    total_ms = float(per_block.sum()) * 1e3
    if total_ms < MIN_TIMED_MS:
        warnings.append(f"timed blocks sum to {total_ms:.2f} ms (< {MIN_TIMED_MS} ms): value is launch-jitter sensitive")

    if rank == 0:
        roof = {"bound": "hbm", "kernel": kname,
                "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": achieved / HBM_PEAK,
                **pmc_traffic(kname, args.format, size, n),
                "algorithmic_bytes_per_launch": alg_bytes,
                "avg_launch_us": k_ms * 1e3, "launches_timed": reps,
                "frac_of_measured_copy_ceiling": achieved / 6.29e12,
                **dominant_kernels.notes.get(kname, {})}
        state_bytes = int(env.universe_bits.numel() * 8 if packed else env.universe.numel() * env.universe.element_size())
        if state_bytes <= 256 << 20:
            # the library keeps the output of step t cacheable below 256 MiB per state array (write-back stores / cached
            # packed strips): part of what step t+1 reads comes out of the 256 MiB Infinity Cache, not HBM
            if "traffic_model" in roof:
                roof["cache_note"] = ("state array of %d MiB <= 256 MiB Infinity Cache: the fp32 observation is written with "
                                      "write-back stores and not read back by the next step; run --instances %d for the "
                                      "cache-free figure" % (state_bytes >> 20, max(n * 16, 65536) if size == 64 else n * 8))
            else:
                roof["cache_note"] = ("state array of %d MiB <= 256 MiB Infinity Cache: `achieved` counts algorithmic bytes, part "
                                      "of the re-read state is served by the cache; run --instances %d for the cache-free figure"
                                      % (state_bytes >> 20, max(n * 16, 65536) if size == 64 else n * 8))
        if flop:
            roof["flop_per_launch"] = flop
            roof["flop_frac_fp32_valu"] = flop / (k_ms * 1e-3) / FP32_PEAK
        if len(timed) > 1:
            roof["other_kernels_us"] = {nm: ms * 1e3 for ms, nm, _, _, _ in timed if nm != kname}
        out = {
            "metric": f"env-steps/sec at N={n}x64x64 ({args.rule}, fp32 drop-in grids, action every step)"
                      if args.format == "f32" and size == 64 else
                      f"env-steps/sec at N={n}x{size}x{size} ({args.rule}, {args.format} grids, action every step)",
            "value": world * n * args.steps / dt,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "timing": {"estimator": "median over `repeats` blocks of exactly `steps` steps, each bracketed by barrier + "
                                    "synchronize; per block the max over ranks",
                       "repeats": repeats, "block_ms_min": float(per_block.min()) * 1e3,
                       "block_ms_median": dt * 1e3, "block_ms_max": float(per_block.max()) * 1e3,
                       "ms_per_step_min": float(per_block.min()) / args.steps * 1e3,
                       "ms_per_step_max": float(per_block.max()) / args.steps * 1e3,
                       "rank_ms_per_step_min": min(rank_ms), "rank_ms_per_step_max": max(rank_ms),
                       "final_stats_flush_us": flush_us, "steps_run": step_no,
                       "rank0_cpus": (f"{pinned[0]}-{pinned[-1]}" if pinned else "unpinned")},
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64 bit-planes" if packed else args.format,
            "data": "synthetic",
            "config": {"workload": f"{cfg_name(args, n, size)}: {args.rule} N={n}/GPU {size}x{size} "
                                   f"{args.format} grids, Bernoulli(0.1) {asize}x{asize} {args.format} action per step "
                                   f"(ring of {len(ring)} device buffers), CARLE.step drop-in surface",
                       "instances_per_gpu": n, "height": size, "width": size,
                       "action": asize, "rule": args.rule, "sharding": f"batch x{world}",
                       "action_source": args.agent,
                       "step_surface": {"none": "env.step (eager)", "calls": "recorded library calls of env.step, replayed",
                                        "graph": "HIP graph of env.step"}[args.replay],
                       "reward_nets": ("RND2D" if args.rnd else "") + ("+AE2D" if args.ae else "") +
                                      ((" forward+backward, Adam every 64 steps" if args.train else " forward")
                                       if (args.rnd or args.ae) else ""),
                       "reward_consumers": "reward stream" if rs is not None else "same stream",
                       "collective": f"all-gather of 24 B/rank reward moments every {args.stats_every} steps"
                                     f" ({args.backend if world > 1 else 'single rank'})" +
                                     ("; all-reduce of the flat predictor gradients before every Adam step"
                                      if args.train and world > 1 and not args.per_rank_predictors else ""),
                       "state_input": ("the bit-packed copy of the state that the previous step's launch left beside its fp32 "
                                       "observation (1/32 of the bytes), read while env.universe is the untouched tensor that step "
                                       "returned; the fp32 observation is written every step, the fp32 action read every step"
                                       if (args.format == "f32" and size in (64, 256, 512) and getattr(env, "packed_state", False)
                                           and (size != 64 or n > 16)) else "the state array"),
                       "state_bytes_per_array": int(env.universe_bits.numel() * 8 if packed else
                                                    env.universe.numel() * env.universe.element_size())},
            "cell_updates_per_s": world * n * args.steps * size * size / dt,
            "final_population_rank0": pop,
            "rewards_counted_all_ranks": reward_count,
            "roofline": roof,
        }
        if warnings:
            out["warnings"] = warnings
        if world == 1 and not args.no_cpu_baseline and not light:
            b, s = env.birth, env.survive
            out["cpu_baseline"] = cpu_baseline(n, size, (b, s), args.cpu_seconds)
        return out
    return None


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    # torch sizes its OpenMP pool by the visible cores (256 on a GPU box whose container may use 16): any CPU tensor
    # op above the parallel grain then wakes 256 spinning workers, the cgroup quota runs out and the process is
    # frozen for the rest of the 100 ms period (92 ms stalls inside the timed region, measured at N=65536)
    torch.set_num_threads(max(1, _host_threads() // max(1, int(os.environ.get("WORLD_SIZE", "1")))))   # ranks share the node's quota
    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        sys.exit(2)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(launch_ranks(args, argv))
    run_rank(args)


if __name__ == "__main__":
    main()

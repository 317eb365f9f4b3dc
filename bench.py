#!/usr/bin/env python3
"""bench.py -- env-steps/s of the CARLE hot path on MI355X (BASELINE.json metric).

A "step" is one CARLE.step(action) over the whole batch: the fused HIP CA kernel
(toggle + 3x3 torus count + B/S rule) plus the epilogue kernel, driven through the drop-in
Python surface (carle_amd.env.CARLE).  Workload at N=1: BASELINE.json configs[1] -- B3/S23,
N=4096, 64x64 grids, fp32 drop-in format, a fresh device-resident Bernoulli(0.1) 64x64 action
every step (ring of action buffers larger than the 256 MiB Infinity Cache, so state and action
really stream from HBM).  For --gpus N > 1 the batch is sharded: every rank owns 4096 instances
(weak scaling, no data-path collective: universes are independent).

Prints ONE JSON line on rank 0 (contract in the task description), with two extra objects:
  roofline      algorithmic bytes of the dominant kernel (ca_step_f32_64) / its average launch
                duration measured here with HIP events on the launch stream, against 8 TB/s
  cpu_baseline  oracle/torch_port.py (a port of the reference's per-step ATen op sequence; the
                reference itself cannot travel) timed on this host's cores, bounded sample
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW"


def parse():
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
    ap.add_argument("--sync-reward", action="store_true",
                    help="run the reward consumers (reward nets, reward statistics) on the CA "
                         "stream instead of the reward stream")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1")
    ap.add_argument("--stats-every", type=int, default=64,
                    help="steps between reward-statistics all-gathers (reporting interval)")
    ap.add_argument("--separate-stats-kernel", action="store_true",
                    help="reduce the rewards with the stand-alone moments kernel every step instead of "
                         "inside the step epilogue (A/B switch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def cpu_baseline(size, rule, seconds):
    """Port of the reference step (oracle/torch_port.py) on the host cores, bounded sample."""
    from oracle.torch_port import TorchPortEnv
    # the threads this process may actually run on (a 1-GPU box exposes a CPU share, not all cores)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:        # cgroup v2 CPU quota ("1600000 100000" = 16 CPUs): more threads only thrash
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    cores = max(1, min(cores, int(os.environ.get("CARLE_CPU_THREADS", cores))))
    torch.set_num_threads(cores)
    n = 256
    b, s = rule
    env = TorchPortEnv(n, size, min(64, size), b, s)
    g = torch.Generator().manual_seed(1234)
    env.universe = (torch.rand(n, 1, size, size, generator=g) < 0.5).float()
    acts = [(torch.rand(n, 1, min(64, size), min(64, size), generator=g) <= 0.1).float()
            for _ in range(4)]
    for i in range(2):
        env.step(acts[i % 4])
    t0 = time.perf_counter()
    k = 0
    while True:
        env.step(acts[k % 4])
        k += 1
        if time.perf_counter() - t0 >= seconds and k >= 3:
            break
    dt = time.perf_counter() - t0
    return {"value": n * k / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"oracle/torch_port.py (reference ATen op sequence), N={n} {size}x{size} "
                      f"{k} steps in {dt:.1f} s, torch {torch.__version__} {cores} threads"}


def cfg_name(args, n, size):
    if args.format == "f32" and size == 64 and n == 4096:
        return "BASELINE configs[3] (CA + RND2D forward)" if args.rnd else "BASELINE configs[1]"
    if args.format == "b1" and size == 512 and n == 4096:
        return "BASELINE configs[4] (bit-packed)"
    if size == 256:
        return "BASELINE configs[2] shape (4096 instances per GPU)"
    return "custom"


def pmc_traffic(size, n):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE; tools/pmc_summary.py).  The
    counters cannot be collected from inside this process, so the value is read from
    profiles/pmc_latest.json when it describes the same workload, else null."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        rec = json.load(open(path))
        if rec.get("size") == size and rec.get("instances") == n:
            return {"traffic": rec["traffic_bytes_per_launch"],
                    "traffic_source": rec.get("source", "profiles/pmc_latest.json")}
    except (OSError, ValueError, KeyError):
        pass
    return {"traffic": None}


def main():
    args = parse()
    from carle_amd import dist as cdist
    rank, world, dev = cdist.init_from_env(args.backend)
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
        # forward passes only.  RND2D alone is fused into the CA kernel (one stream); a stacked AE2D
        # runs its kernels on the reward stream beside the next CA step unless --sync-reward
        kw = dict(async_reward=(args.ae or args.train) and not args.sync_reward, train_predictor=args.train)
        wenv = RND2D(env, **kw)
        if args.ae:
            wenv = AE2D(wenv, **kw)
        wenv.eval()
        wenv.to(dev)
    # with a stacked AE2D the reward consumers (its kernels, the per-step moments kernel) run on the
    # reward stream unless --sync-reward: they overlap the CA kernel of the next step.  Measured:
    # with no reward nets, or with RND2D alone (fused into the CA kernel), the cross-stream events
    # cost more than they hide (45.6 vs 39.1 and 52.8 vs 46.5 us/step): those stay on one stream
    rs = None
    if (args.ae or args.train) and not args.sync_reward:
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

    def one_step(i):
        action = ring[i % len(ring)] if agent is None else agent(env.universe)
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

    for i in range(args.warmup):
        one_step(i)
    join_rewards()
    stats.flush()                       # first collective (RCCL setup) stays out of the timing
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(i)
    join_rewards()
    stats.flush()
    torch.cuda.synchronize(dev)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    pop = float(env.universe.sum().item())      # keeps the result live / sanity
    state = env.universe_bits if packed else env.universe
    step_fn = hip.ca_step_b1 if packed else hip.ca_step_u8 if bytes_fmt else hip.ca_step_f32

    # ---- roofline of the dominant kernel: back-to-back launches of the step kernel alone,
    # HIP events on the launch stream (torch's current stream is the stream the ABI is given)
    rule = env.rule_mask()
    a_buf, b_buf = state, torch.empty_like(state)
    reps = max(32, min(args.steps, 256))
    for i in range(4):
        step_fn(a_buf, ring[i % len(ring)], b_buf, rule, env._status, 0)
        a_buf, b_buf = b_buf, a_buf
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    e0.record()
    for i in range(reps):
        step_fn(a_buf, ring[i % len(ring)], b_buf, rule, env._status, 0)
        a_buf, b_buf = b_buf, a_buf
    e1.record()
    torch.cuda.synchronize(dev)
    k_ms = e0.elapsed_time(e1) / reps
    if packed:
        alg_bytes = n * (size * size * 2 + asize * asize) // 8
    elif bytes_fmt:
        alg_bytes = n * (size * size * 2 + asize * asize)
    else:
        alg_bytes = n * (size * size * 8 + asize * asize * 4)
    achieved = alg_bytes / (k_ms * 1e-3)

    cells = "carle::U8Cells" if bytes_fmt else "carle::F32Cells"
    if rank == 0:
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
                       "reward_nets": ("RND2D" if args.rnd else "") + ("+AE2D" if args.ae else "") +
                                      ((" forward+backward, Adam every 64 steps" if args.train else " forward")
                                       if (args.rnd or args.ae) else ""),
                       "reward_consumers": "reward stream" if rs is not None else "same stream",
                       "collective": f"all-gather of 24 B/rank reward moments every {args.stats_every} steps"},
            "cell_updates_per_s": world * n * args.steps * size * size / dt,
            "final_population_rank0": pop,
            "roofline": {"bound": "hbm", "kernel": ("ca_step_b1_kernel" if packed else
                                    f"ca_step_64_kernel<{cells}, {cells}, true, ...>" if size == 64 else
                                    f"ca_step_wide_kernel<{cells}, {cells}, true, ...>"),
                         "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK,
                         **(pmc_traffic(size, n) if args.format == "f32" else {"traffic": None}),
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "avg_launch_us": k_ms * 1e3, "launches_timed": reps},
        }
        if world == 1 and not args.no_cpu_baseline:
            b, s = env.birth, env.survive
            out["cpu_baseline"] = cpu_baseline(size, (b, s), args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

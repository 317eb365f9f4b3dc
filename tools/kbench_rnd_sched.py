#!/usr/bin/env python3
"""Same-process A/B of the workgroup count of the fused CA + RND2D kernel at N x 64x64 (CARLE_RND_BLOCKS; 0 = the library's
choice; KBENCH_RND_BLOCKS=a,b,c: the counts to compare) -- kernel alone and the one-launch step.
    python tools/kbench_rnd_sched.py [N ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd import hip, _lib, dist as cd   # noqa: E402
from carle_amd.env import CARLE                # noqa: E402
from carle_amd.mcl import RND2D                # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [4096]:
    torch.manual_seed(0)
    env = CARLE(width=64, height=64, instances=n, device="cuda:0")
    w = RND2D(env, train_predictor=False)
    w.eval(); w.reset()
    st = cd.RunningRewardStats("cuda:0", sync_every=1 << 30)
    st.attach(w)
    g = torch.Generator(device="cuda").manual_seed(1)
    env.universe = (torch.rand(n, 1, 64, 64, device="cuda", generator=g) < 0.5).float()
    ring = [(torch.rand(n, 1, 64, 64, device="cuda", generator=g) <= 0.1).float() for _ in range(6)]
    w.step(ring[0])
    dev = env.universe.device
    pb, rb = w._pred_blob.get(dev), w._rand_blob.get(dev)
    bufs = [env.universe, torch.empty_like(env.universe)]
    rule = env.rule_mask()
    reward = torch.empty(n, 1, device=dev)
    partials = torch.zeros(_lib.load().carle_rnd_step_partials_doubles(n), dtype=torch.float64, device=dev)
    raw = hip._stream(reward)
    alg = n * (64 * 64 * 12 + 4)

    def kernel(i):
        hip.ca_step_f32_rnd(bufs[i & 1], ring[i % 6], bufs[(i & 1) ^ 1], rule, env._status, 0, pb, rb, w._ws, w._step_bonus, True)

    def step(i):
        hip.env_step_f32_rnd(bufs[i & 1], ring[i % 6], bufs[(i & 1) ^ 1], reward, rule, env._status, i & 1, True, pb, rb, w._ws,
                             w._step_bonus, 1.0, True, None, raw, None, True, deferred=True, partials=partials)

    for rep in range(int(os.environ.get("KBENCH_REPS", "4"))):
        for name, fn in (("kernel alone", kernel), ("one-launch step", step)):
            counts = os.environ.get("KBENCH_RND_BLOCKS", "1024,768,512,896").split(",")      # workgroup counts to compare
            for form, env_kv in [("library's choice", {})] + [(f"{c} workgroups", {"CARLE_RND_BLOCKS": c}) for c in counts]:
                os.environ.pop("CARLE_RND_BLOCKS", None)
                os.environ.update(env_kv)
                hip.reload_knobs()
                for i in range(6):
                    fn(i)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for i in range(300):
                    fn(i)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / 300 * 1e3
                print(f"N={n:6d} {name:16s} {form:26s}: {us:7.2f} us  frac {alg / us / 8e6:.3f}", flush=True)
    del env, w, ring, bufs
    torch.cuda.empty_cache()

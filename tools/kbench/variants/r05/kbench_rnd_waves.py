#!/usr/bin/env python3
"""Same-process A/B of the one-launch CA + RND2D step (carle_env_step_f32_rnd, CARLE_STEP_DEFERRED): four wavefronts per
grid / four workgroups per CU (ca_step_rnd64_kernel, shipped) against eight wavefronts per grid / two workgroups per CU
(ca_step_rnd64x8_kernel, CARLE_RND_WAVES=8).  Alternating blocks of 300 back-to-back calls, HIP events; first the two forms'
outputs (grids, rewards, bonuses) are compared bit for bit.   python tools/kbench_rnd_waves.py [N ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd import hip, _lib, dist as cd   # noqa: E402
from carle_amd.env import CARLE                # noqa: E402
from carle_amd.mcl import RND2D                # noqa: E402


VARIANTS = {4: {}, 8: {"CARLE_RND_WAVES": "8"}, 16: {"CARLE_RND_PAIR": "1"}}
NAMES = {4: "4 waves per grid, 512-entry tables (shipped) ", 8: "8 waves per grid, 2 workgroups per CU         ",
         16: "pair table, 1 workgroup of 4 groups per CU     "}


def waves(k):
    for key in ("CARLE_RND_WAVES", "CARLE_RND_PAIR"):
        os.environ.pop(key, None)
    os.environ.update(VARIANTS[k])
    hip.reload_knobs()


for n in [int(a) for a in sys.argv[1:]] or [4096]:
    waves(4)
    torch.manual_seed(0)
    env = CARLE(width=64, height=64, instances=n, device="cuda:0")
    w = RND2D(env, train_predictor=False)
    w.eval(); w.reset()
    st = cd.RunningRewardStats("cuda:0", sync_every=1 << 30)
    st.attach(w)
    g = torch.Generator(device="cuda").manual_seed(1)
    u0 = (torch.rand(n, 1, 64, 64, device="cuda", generator=g) < 0.5).float()
    env.universe = u0.clone()
    ring = [(torch.rand(n, 1, 64, 64, device="cuda", generator=g) <= 0.1).float() for _ in range(6)]
    w.step(ring[0])                                 # builds the tables / workspace
    dev = u0.device
    pb, rb = w._pred_blob.get(dev), w._rand_blob.get(dev)
    rule = env.rule_mask()
    partials = torch.zeros(_lib.load().carle_rnd_step_partials_doubles(n), dtype=torch.float64, device=dev)
    raw = hip._stream(u0)
    alg = n * (64 * 64 * 12 + 4)

    def run(k, steps, keep=False):
        bufs = [u0.clone(), torch.empty_like(u0)]
        reward = torch.empty(n, 1, device=dev)
        outs = []
        for i in range(steps):
            hip.env_step_f32_rnd(bufs[i & 1], ring[i % 6], bufs[(i & 1) ^ 1], reward, rule, env._status, i & 1, True, pb, rb,
                                 w._ws, w._step_bonus, 1.0, True, None, raw, None, True, deferred=True, partials=partials)
            if keep:
                outs.append((bufs[(i & 1) ^ 1].clone(), reward.clone(), w._step_bonus.clone()))
        return outs

    ref = None
    kinds = [int(x) for x in os.environ.get("KBENCH_KINDS", "4,16").split(",")]
    for k in kinds:
        waves(k)
        env.reset()
        outs = run(k, 5, keep=True)
        torch.cuda.synchronize()
        if ref is None:
            ref = outs
        else:
            same = all(torch.equal(a, b) for x, y in zip(ref, outs) for a, b in zip(x, y))
            print(f"N={n}: [{NAMES[k].strip()}] bit-identical to the four-wave form over 5 steps: {same}", flush=True)
    for rep in range(3):
        for k in kinds:
            waves(k)
            run(k, 12)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            run(k, 300)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 300 * 1e3
            print(f"N={n} {NAMES[k]}: {us:6.2f} us per step  {alg / us / 8e6:.3f} of 8 TB/s", flush=True)
    del env, w, ring, u0, partials
    torch.cuda.empty_cache()

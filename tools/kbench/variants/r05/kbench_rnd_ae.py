#!/usr/bin/env python3
"""AE2D(RND2D(CARLE)).step at N x 64x64, forward passes: the one-launch form (carle_env_step_f32_rnd_ae, round-5 experiment)
against RND2D's one-launch step + AE2D's bonus kernel (CARLE_FUSED_RND_AE=0), 200 steps each, HIP events, alternating.
    python tools/kbench_rnd_ae.py [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd.env import CARLE                # noqa: E402
from carle_amd.mcl import RND2D, AE2D          # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
g = torch.Generator(device="cuda").manual_seed(1)
u0 = (torch.rand(n, 1, 64, 64, device="cuda", generator=g) < 0.5).float()
ring = [(torch.rand(n, 1, 64, 64, device="cuda", generator=g) <= 0.1).float() for _ in range(6)]
stacks = {}
for fused in ("1", "0"):
    os.environ["CARLE_FUSED_RND_AE"] = fused
    torch.manual_seed(0)
    env = CARLE(width=64, height=64, instances=n, device="cuda:0")
    w = AE2D(RND2D(env, train_predictor=False), train_predictor=False)
    w.eval(); w.reset()
    env.universe = u0.clone()
    stacks[fused] = w
for rep in range(3):
    for fused, w in stacks.items():
        os.environ["CARLE_FUSED_RND_AE"] = fused
        for i in range(8):
            w.step(ring[i % 6])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(200):
            w.step(ring[i % 6])
        e1.record()
        torch.cuda.synchronize()
        print(f"N={n} {'one launch (CA + RND2D + AE2D)' if fused == '1' else 'CA + RND2D launch, AE2D launch  '}: "
              f"{e0.elapsed_time(e1) / 200 * 1e3:7.2f} us per step", flush=True)

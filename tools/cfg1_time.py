#!/usr/bin/env python3
"""BASELINE configs[0] (N=1, 64x64 Life, 256 steps) and small batches: env-steps/s through
CARLE.step (the drop-in loop) and through CARLE.run (k steps per library call)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd.agents import RandomAgent     # noqa: E402
from carle_amd.env import CARLE              # noqa: E402

for n in (1, 16, 64, 1024):
    env = CARLE(width=64, height=64, instances=n, device="cuda:0")
    env.reset()
    env.universe = (torch.rand(n, 1, 64, 64, device="cuda") < 0.5).float()
    agent = RandomAgent(action_width=64, action_height=64, fused=True, seed=1)
    zero = torch.zeros(n, 1, 64, 64, device="cuda")
    acts = (torch.rand(256, n, 1, 64, 64, device="cuda") <= 0.1).float() if n <= 64 else None
    rows = []
    for name, fn in (("step(zero action)", lambda: [env.step(zero) for _ in range(256)]),
                     ("step(fused RandomAgent)", lambda: [env.step(agent(env.universe)) for _ in range(256)]),
                     ("run(steps=256, no action)", lambda: env.run(steps=256)),
                     ("run(agent, steps=256)", lambda: env.run(agent=agent, steps=256)),
                     ("run(actions[256])", (lambda: env.run(actions=acts)) if acts is not None else None)):
        if fn is None:
            continue
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        rows.append(f"   {name:28s} {best / 256 * 1e6:7.2f} us/step  {n * 256 / best:.3e} env-steps/s")
    print(f"N={n} 64x64 B3/S23, 256 steps (best of 5):")
    print("\n".join(rows), flush=True)

#!/usr/bin/env python3
"""Per-step time of PredictionBonus / SurpriseBonus / AE2D stepping on the environment with predictor updates
(batch_size 64): the reference's default geometry (1 and 16 instances of 256x256) and N=1024 x 64x64.  GPU box."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd.env import CARLE                      # noqa: E402
import carle_amd.mcl as M                            # noqa: E402

for size, n in ((256, 1), (256, 16), (64, 1024)):
    for cls in ("AE2D", "SurpriseBonus", "PredictionBonus"):
        torch.manual_seed(0)
        env = CARLE(width=size, height=size, instances=n, device="cuda:0")
        w = getattr(M, cls)(env)
        w.eval(); w.reset()
        env.universe = (torch.rand(n, 1, size, size, device="cuda") < 0.3).float()
        acts = [(torch.rand(n, 1, 64, 64, device="cuda") < 0.1).float() for _ in range(4)]
        for t in range(70):
            w.step(acts[t % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        steps = 256
        for t in range(steps):
            w.step(acts[t % 4])
        torch.cuda.synchronize()
        print(f"N={n:5d} {size}x{size} {cls:16s}: {(time.perf_counter() - t0) / steps * 1e6:8.1f} us per step (4 Adam updates inside)", flush=True)

#!/usr/bin/env python3
"""Host-side cost of one CARLE.step (Python + ctypes + launch), measured on the GPU box:
runs steps with a tiny batch so the GPU is never the bottleneck, then cProfile."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd.env import CARLE          # noqa: E402
from carle_amd import dist as cdist       # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
env = CARLE(width=64, height=64, instances=n, device="cuda:0")
env.reset()
acts = [(torch.rand(n, 1, 64, 64, device="cuda") < 0.1).float() for _ in range(4)]
stats = cdist.RunningRewardStats("cuda:0", sync_every=16)


def loop(k, with_stats):
    for i in range(k):
        obs, reward, done, info = env.step(acts[i & 3])
        if with_stats:
            stats.update(reward)


loop(200, True)
torch.cuda.synchronize()
for ws in (False, True):
    t0 = time.perf_counter()
    loop(2000, ws)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"with_stats={ws}: host {1e6 * (t1 - t0) / 2000:.1f} us/step")
pr = cProfile.Profile()
pr.enable()
loop(2000, True)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)

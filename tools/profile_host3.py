#!/usr/bin/env python3
"""Pure host enqueue cost per step: short bursts (150 steps < launch-queue depth) from an idle GPU."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd.env import CARLE
from carle_amd.mcl import RND2D
from carle_amd.agents import RandomAgent
from carle_amd import dist as cdist

n = 4096
for label, async_reward, wrap in (("CA only", False, False), ("CA+RND sync", False, True), ("CA+RND reward stream", True, True)):
    env = CARLE(width=64, height=64, instances=n, device="cuda:0")
    w = RND2D(env, async_reward=async_reward, train_predictor=False) if wrap else env
    if wrap:
        w.eval()
    w.reset()
    acts = [(torch.rand(n, 1, 64, 64, device="cuda") < 0.1).float() for _ in range(4)]
    stats = cdist.RunningRewardStats("cuda:0", sync_every=16)

    def loop(k):
        for i in range(k):
            obs, reward, done, info = w.step(acts[i & 3])
            rs = env.__dict__.get("reward_stream")
            if rs is None:
                stats.update(reward)
            elif stats._pending_steps + 1 >= stats.sync_every:
                with rs.context():
                    stats.update(reward)
            else:
                stats.update(reward, rs.raw)
    loop(50); torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter(); loop(100); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        best = min(best, (t1 - t0) / 100)
    print(f"{label}: host enqueue {1e6 * best:.1f} us/step; last burst incl. drain {1e6 * (t2 - t0) / 100:.1f} us/step")

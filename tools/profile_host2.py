#!/usr/bin/env python3
"""Host-side cost per step of the wrapped environment (tiny batch: GPU never the bottleneck)."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd.env import CARLE
from carle_amd.mcl import RND2D
from carle_amd import dist as cdist

n = 8
for async_reward in (False, True):
    env = CARLE(width=64, height=64, instances=n, device="cuda:0")
    w = RND2D(env, async_reward=async_reward, train_predictor=False)
    w.eval(); w.reset()
    acts = [(torch.rand(n, 1, 64, 64, device="cuda") < 0.1).float() for _ in range(4)]
    stats = cdist.RunningRewardStats("cuda:0", sync_every=16)

    def loop(k):
        for i in range(k):
            obs, reward, done, info = w.step(acts[i & 3])
            rs = env.__dict__.get("reward_stream")
            if rs is not None:
                with torch.cuda.stream(rs):
                    stats.update(reward)
            else:
                stats.update(reward)
    loop(200); torch.cuda.synchronize()
    t0 = time.perf_counter(); loop(2000); t1 = time.perf_counter(); torch.cuda.synchronize()
    print(f"async_reward={async_reward}: host {1e6 * (t1 - t0) / 2000:.1f} us/step")
    pr = cProfile.Profile(); pr.enable(); loop(1000); pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)

#!/usr/bin/env python3
"""cProfile of the cfg3 loop (CA + RND2D forward, reward stream) -- where the host time goes."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd.env import CARLE
from carle_amd.mcl import RND2D
from carle_amd import dist as cdist

n = 4096
env = CARLE(width=64, height=64, instances=n, device="cuda:0")
w = RND2D(env, async_reward=True, train_predictor=False)
w.eval(); w.reset()
acts = [(torch.rand(n, 1, 64, 64, device="cuda") < 0.1).float() for _ in range(4)]
stats = cdist.RunningRewardStats("cuda:0", sync_every=16)
rs = None


def loop(k):
    global rs
    for i in range(k):
        obs, reward, done, info = w.step(acts[i & 3])
        rs = env.__dict__.get("reward_stream")
        if stats._pending_steps + 1 >= stats.sync_every:
            with rs.context():
                stats.update(reward)
        else:
            stats.update(reward, rs.raw)


loop(64); torch.cuda.synchronize()
t0 = time.perf_counter(); loop(1024); rs.join(); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"plain: {1e6 * (t1 - t0) / 1024:.1f} us/step")
pr = cProfile.Profile()
pr.enable(); loop(1024); pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)

#!/usr/bin/env python3
"""cProfile of the reference's default geometry (16 x 256x256, RND2D+AE2D forward-only): host time per step."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd.agents import RandomAgent
from carle_amd.mcl import RND2D, AE2D
from carle_amd.env import CARLE
env = AE2D(RND2D(CARLE(width=256, height=256, instances=16, device="cuda:0"), train_predictor=False), train_predictor=False)
env.eval(); obs = env.reset()
agent = RandomAgent(action_width=64, action_height=64)
def loop(k):
    for i in range(k): env.step(agent(obs))
loop(50); torch.cuda.synchronize()
t0 = time.perf_counter(); loop(500); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"plain: {1e6 * (t1 - t0) / 500:.1f} us/step")
pr = cProfile.Profile(); pr.enable(); loop(500); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)

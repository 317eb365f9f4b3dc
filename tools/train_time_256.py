#!/usr/bin/env python3
"""The reference's DEFAULT configuration (carle/train_mcl.py: 16 instances of 256x256, RND2D + AE2D with
predictor updates every 64 steps) on this build: per-step time of the training loop and of its parts."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd.harness import train
from carle_amd.agents import RandomAgent
from carle_amd.mcl import RND2D, AE2D
from carle_amd.env import CARLE

for mcl, label in (() if os.environ.get("SKIP_TRAIN") else (((RND2D,), "RND2D"), ((RND2D, AE2D), "RND2D+AE2D"))):
    for instances in (16, 64):
        kw = dict(instances=instances, out_dir=None, verbose=False, width=256, height=256, mcl=mcl)
        train(RandomAgent, steps=[1, 130], **kw)
        torch.cuda.synchronize()
        t0 = time.time()
        train(RandomAgent, steps=[1, 256], **kw)
        torch.cuda.synchronize()
        dt = time.time() - t0
        print(f"train loop {label} N={instances} 256x256: 256 steps in {dt:.3f} s -> "
              f"{instances * 256 / dt:.3e} env-steps/s ({1e3 * dt / 256:.3f} ms/step)")
for instances in (16, 256):
    env = CARLE(width=256, height=256, instances=instances, device="cuda:0")
    env.reset()
    agent = RandomAgent(action_width=64, action_height=64)
    obs = env.reset()
    for i in range(20): env.step(agent(obs))
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(256): env.step(agent(obs))
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"bare CARLE N={instances} 256x256: {1e3 * dt / 256:.3f} ms/step")
# forward-only (evaluation/eval.py style)
for instances in (16, 256):
    env = AE2D(RND2D(CARLE(width=256, height=256, instances=instances, device="cuda:0"), train_predictor=False),
               train_predictor=False)
    env.eval(); env.reset()
    agent = RandomAgent(action_width=64, action_height=64)
    obs = env.reset()
    for i in range(20): env.step(agent(obs))
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(256): env.step(agent(obs))
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"forward-only RND2D+AE2D N={instances} 256x256: {1e3 * dt / 256:.3f} ms/step")

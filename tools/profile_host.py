#!/usr/bin/env python3
"""Host-side cost of one step (Python + ctypes + launches), measured on the GPU box.

One script, several configurations (was five near-duplicate scripts in round 1):
  --stack ca | rnd | rnd+ae     environment stack (bare CARLE, RND2D on it, AE2D(RND2D))
  --size / --instances          geometry (default: 64x64, 8 instances: the GPU is never the bottleneck)
  --async-reward                reward nets on the reward stream
  --agent ring | random | fused action source
  --burst K                     time K steps enqueued from an idle GPU (pure enqueue cost), no profile
  --profile                     cProfile of 500 steps, 25 hottest functions
"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd.agents import RandomAgent     # noqa: E402
from carle_amd.env import CARLE              # noqa: E402
from carle_amd.mcl import AE2D, RND2D        # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--stack", default="ca", choices=["ca", "rnd", "rnd+ae"])
ap.add_argument("--size", type=int, default=64)
ap.add_argument("--instances", type=int, default=8)
ap.add_argument("--async-reward", action="store_true")
ap.add_argument("--agent", default="ring", choices=["ring", "random", "fused"])
ap.add_argument("--burst", type=int, default=0)
ap.add_argument("--profile", action="store_true")
args = ap.parse_args()

n, size = args.instances, args.size
env = CARLE(width=size, height=size, instances=n, device="cuda:0")
kw = dict(async_reward=args.async_reward, train_predictor=False)
if args.stack != "ca":
    env = RND2D(env, **kw)
if args.stack == "rnd+ae":
    env = AE2D(env, **kw)
if args.stack != "ca":
    env.eval()
obs = env.reset()
ring = [(torch.rand(n, 1, 64, 64, device="cuda") <= 0.1).float() for _ in range(4)]
agent = None if args.agent == "ring" else RandomAgent(action_width=64, action_height=64, fused=args.agent == "fused")


def loop(k):
    for i in range(k):
        env.step(ring[i & 3] if agent is None else agent(obs))


loop(50)
torch.cuda.synchronize()
label = f"{args.stack} {n} x {size}x{size} agent={args.agent} async={args.async_reward}"
if args.burst:
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop(args.burst)
        best = min(best, time.perf_counter() - t0)
        torch.cuda.synchronize()
    print(f"{label}: {1e6 * best / args.burst:.1f} us/step of host enqueue time ({args.burst}-step bursts from an idle GPU)")
else:
    t0 = time.perf_counter()
    loop(500)
    torch.cuda.synchronize()
    print(f"{label}: {1e6 * (time.perf_counter() - t0) / 500:.1f} us/step")
if args.profile:
    pr = cProfile.Profile()
    pr.enable()
    loop(500)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(25)

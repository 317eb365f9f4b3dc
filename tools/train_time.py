#!/usr/bin/env python3
"""Throughput of the reference's training loop (carle/train_mcl.py) on this build: RND2D + AE2D
wrappers WITH predictor updates every 64 steps, device-side RandomAgent."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd.harness import train
from carle_amd.agents import RandomAgent

for instances, size, steps in ((16, 256, 256), (8, 256, 256), (1024, 64, 128), (4096, 64, 128)):
    train(RandomAgent, instances=instances, steps=[1, 8], out_dir=None, verbose=False, width=size, height=size)
    torch.cuda.synchronize()
    t0 = time.time()
    train(RandomAgent, instances=instances, steps=[1, steps], out_dir=None, verbose=False, width=size, height=size)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"train loop N={instances} {size}x{size}: {steps} steps in {dt:.2f} s -> {instances * steps / dt:.3e} env-steps/s "
          f"({1e3 * dt / steps:.2f} ms/step, updates every 64 steps)")

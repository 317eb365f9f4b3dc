#!/usr/bin/env python3
"""Throughput of the reference's training loop (carle/train_mcl.py) on this build, WITH predictor
updates every 64 steps, device-side RandomAgent.  A warm-up pass of 2 windows absorbs MIOpen's
one-time kernel compilation for the autograd replay of the AE2D wrapper."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd.harness import train
from carle_amd.agents import RandomAgent
from carle_amd.mcl import RND2D, AE2D

for mcl, label in (((RND2D,), "RND2D"), ((RND2D, AE2D), "RND2D+AE2D")):
    for instances, size, steps in ((16, 64, 256), (1024, 64, 256), (4096, 64, 256)):
        kw = dict(instances=instances, out_dir=None, verbose=False, width=size, height=size, mcl=mcl)
        train(RandomAgent, steps=[1, 130], **kw)
        torch.cuda.synchronize()
        t0 = time.time()
        train(RandomAgent, steps=[1, steps], **kw)
        torch.cuda.synchronize()
        dt = time.time() - t0
        print(f"train loop {label} N={instances} {size}x{size}: {steps} steps in {dt:.3f} s -> "
              f"{instances * steps / dt:.3e} env-steps/s ({1e3 * dt / steps:.3f} ms/step)")

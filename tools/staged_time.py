#!/usr/bin/env python3
"""Per-call time of the STAGED reward-net paths (sizes that are not multiples of 64: direct fp32 convolution
kernels): forward bonus and forward + backward, RND2D and AE2D."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd.env import CARLE
from carle_amd.mcl import RND2D, AE2D
for n, size in ((256, 96), (64, 160)):
    env = CARLE(width=size, height=size, instances=n, device="cuda:0")
    rnd = RND2D(env); ae = AE2D(rnd)
    ae.eval(); ae.reset()
    g = (torch.rand(n, 1, size, size, device="cuda") < 0.3).float()
    for w in (rnd, ae):
        w.batch_size = 10 ** 9
        bonus = torch.empty(n, device="cuda")
        reward = torch.zeros(n, 1, device="cuda")
        for name, fn in (("forward", lambda: w._hip_forward(g, False, reward, bonus, None)),
                         ("forward + backward", lambda: w._fused_train(g, None, bonus, None))):
            for _ in range(3): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            print(f"N={n:5d} {size}x{size} {w.my_name} staged {name:18s}: {e0.elapsed_time(e1) * 50:8.1f} us per call", flush=True)

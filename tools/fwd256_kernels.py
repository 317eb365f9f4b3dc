#!/usr/bin/env python3
"""RND2D / AE2D forward at 256x256: tiled kernels (binary observations) against the staged path, HIP events."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd.env import CARLE
from carle_amd.mcl import RND2D, AE2D
for n in (16, 256, 2048):
    env = CARLE(width=256, height=256, instances=n, device="cuda:0")
    rnd = RND2D(env, train_predictor=False); ae = AE2D(rnd, train_predictor=False)
    ae.eval(); ae.reset()
    g = (torch.rand(n, 1, 256, 256, device="cuda") < 0.3).float()
    from carle_amd import hip
    gp = hip.pack_f32_b1(g)
    for _ in range(3): hip.pack_f32_b1(g)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): hip.pack_f32_b1(g)
    e1.record(); torch.cuda.synchronize()
    print(f"N={n:5d} 256x256 pack_f32_b1: {e0.elapsed_time(e1) * 50:9.1f} us per call")
    for w in (rnd, ae):
        w.assume_binary_obs = True
        for _ in range(3): w.get_bonus_only(gp)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): w.get_bonus_only(gp)
        e1.record(); torch.cuda.synchronize()
        print(f"N={n:5d} 256x256 {w.my_name:5s} tiled, bit-packed input: {e0.elapsed_time(e1) * 50:9.1f} us per call")
    for w in (rnd, ae):
        for binary in (False, True):
            w.assume_binary_obs = binary
            for _ in range(3): w.get_bonus_only(g)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(20): w.get_bonus_only(g)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 50
            print(f"N={n:5d} 256x256 {w.my_name:5s} {'tiled ' if binary else 'staged'}: {us:9.1f} us per call "
                  f"({n * 262144 / us / 1e6:6.2f} TB/s of observation reads)")

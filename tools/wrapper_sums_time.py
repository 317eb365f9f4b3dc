#!/usr/bin/env python3
"""Per-call time of carle_instance_sums_f32 (the reduction behind CornerBonus / SpeedDetector / ParsimonyBonus)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd import hip
for n, size in ((16, 256), (256, 256), (4096, 64)):
    x = (torch.rand(n, 1, size, size, device="cuda") < 0.3).float()
    w = [torch.rand(1, 1, size, size, device="cuda") for _ in range(3)]
    for name, args in (("plain sum", ()), ("2 weight planes", (w[0], w[1])), ("3 weight planes", (w[0], w[1], w[2]))):
        for _ in range(3): hip.instance_sums_f32(x, *args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50): hip.instance_sums_f32(x, *args)
        e1.record(); torch.cuda.synchronize()
        print(f"N={n:5d} {size}x{size} {name:16s}: {e0.elapsed_time(e1) * 20:8.1f} us per call", flush=True)

#!/usr/bin/env python3
"""Same-process A/B of the ONE-LAUNCH packed step (carle_env_step_b1, 64x64 window, B36/S23) at N x SxS: half-grid units
(CARLE_B1_DBG=16) against single strips (CARLE_B1_DBG=8) against the library's choice by size.  HIP events around 200
back-to-back calls, packed action ring of 6.      python tools/kbench_b1_env.py [N] [S]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd import hip                      # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
rule = (1 << 12) | (1 << 15) | 0xC              # B36/S23
g = torch.Generator(device="cuda").manual_seed(1)
a = torch.randint(-2 ** 63, 2 ** 63 - 1, (n, 1, size, size // 64), device="cuda", generator=g, dtype=torch.int64)
b = torch.empty_like(a)
ring = [hip.pack_f32_b1((torch.rand(n, 1, 64, 64, device="cuda", generator=g) <= 0.1).float()) for _ in range(6)]
st = hip.new_status(a.device)
reward = torch.empty(n, 1, device="cuda")
bufs = [a, b]
alg = n * (size * size // 4 + 512)


def call(i):
    hip.env_step_b1(bufs[i & 1], ring[i % 6], bufs[(i & 1) ^ 1], reward, rule, st, i & 1)


for rep in range(4):
    for name, knob in (("half-grid units", "16"), ("single strips", "8"), ("by size", None)):
        if knob is None:
            os.environ.pop("CARLE_B1_DBG", None)
        else:
            os.environ["CARLE_B1_DBG"] = knob
        hip.reload_knobs()
        for i in range(6):
            call(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(200):
            call(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 200 * 1e3
        print(f"N={n} {size}x{size} {name:16s}: {us:7.2f} us  {alg / us / 1e6:5.2f} TB/s  frac {alg / us / 8e6:.3f}", flush=True)

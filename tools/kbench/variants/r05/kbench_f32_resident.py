#!/usr/bin/env python3
"""Same-process A/B of CARLE_F32_RESIDENT_BYTES (EXPERIMENTS.md 5.3b): fp32 state arrays beyond the Infinity Cache stream
their stores; with the knob the first R MiB of the array are stored write-back and stay cached for the next step's read.
The whole CARLE.step library call (carle_env_step_f32), 64 x 64 window action ring, HIP events.
    python tools/kbench_f32_resident.py [N@size ...]      default: 65536@64 32768@64 4096@256 1024@512"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd import hip                      # noqa: E402

rule = (1 << 12) | 0xC                          # B3/S23
for arg in sys.argv[1:] or ["65536@64", "32768@64", "4096@256", "1024@512"]:
    n, size = (int(x) for x in arg.split("@"))
    g = torch.Generator(device="cuda").manual_seed(1)
    a = (torch.rand(n, 1, size, size, device="cuda", generator=g) < 0.5).float()
    b = torch.empty_like(a)
    k = max(2, min(6, (512 << 20) // (n * 64 * 64 * 4)))
    ring = [(torch.rand(n, 1, 64, 64, device="cuda", generator=g) <= 0.1).float() for _ in range(k)]
    st = hip.new_status(a.device)
    reward = torch.empty(n, 1, device="cuda")
    bufs = [a, b]
    alg = n * (size * size * 8 + 64 * 64 * 4)
    reps = max(20, int(20e3 / (alg / 5e6)))

    def call(i):
        hip.env_step_f32(bufs[i & 1], ring[i % k], bufs[(i & 1) ^ 1], reward, rule, st, i & 1, True, True)

    for rep in range(2):
        for mib in (0, 64, 96, 112, 128, 160):
            os.environ["CARLE_F32_RESIDENT_BYTES"] = str(mib << 20)
            hip.reload_knobs()
            for i in range(4):
                call(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for i in range(reps):
                call(i)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / reps * 1e3
            print(f"N={n} {size}x{size} f32 ({n * size * size * 4 >> 20} MiB per array) resident {mib:3d} MiB: {us:8.2f} us per step  "
                  f"{alg / us / 8e6:.3f} of 8 TB/s", flush=True)
    os.environ.pop("CARLE_F32_RESIDENT_BYTES", None)
    hip.reload_knobs()
    del a, b, bufs, ring
    torch.cuda.empty_cache()

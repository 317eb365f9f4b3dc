#!/usr/bin/env python3
"""Same-process sweep of the launch geometry of the ONE-LAUNCH packed step (carle_env_step_b1: 64x64 window or no action,
B36/S23): the unit walk (CARLE_B1_DBG 16 = half-grid units, 8 = single strips) and the cap on the number of one-wave
workgroups (CARLE_B1_BLOCKS; 8168 = the persistent launch of rounds 2-3, 0 = the library's default: one wave per strip for
grids taller than 64 rows since round 4).  HIP events around 200 back-to-back calls, action ring of 6.
    python tools/kbench_b1_cap.py            (the table of profiles/r04_b1_launch_cap.txt)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd import hip                      # noqa: E402

rule = (1 << 12) | (1 << 15) | 0xC              # B36/S23


def bench(n, size, configs, reps=2, act=True):
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randint(-2 ** 63, 2 ** 63 - 1, (n, 1, size, size // 64), device="cuda", generator=g, dtype=torch.int64)
    b = torch.empty_like(a)
    ring = [hip.pack_f32_b1((torch.rand(n, 1, 64, 64, device="cuda", generator=g) <= 0.1).float()) for _ in range(6)] \
        if act else [None] * 6
    st = hip.new_status(a.device)
    reward = torch.empty(n, 1, device="cuda")
    bufs = [a, b]
    alg = n * (size * size // 4 + (512 if act else 0))
    for rep in range(reps):
        for dbg, cap in configs:
            os.environ["CARLE_B1_DBG"] = dbg
            if cap:
                os.environ["CARLE_B1_BLOCKS"] = str(cap)
            else:
                os.environ.pop("CARLE_B1_BLOCKS", None)
            hip.reload_knobs()
            for i in range(6):
                hip.env_step_b1(bufs[i & 1], ring[i % 6], bufs[(i & 1) ^ 1], reward, rule, st, i & 1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for i in range(200):
                hip.env_step_b1(bufs[i & 1], ring[i % 6], bufs[(i & 1) ^ 1], reward, rule, st, i & 1)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 200 * 1e3
            walk = {"16": "half-grid units", "8": "single strips", "0": "library's walk"}[dbg]
            print(f"N={n:6d} {size}x{size} {'window   ' if act else 'no action'} {walk:15s} cap={cap if cap else 'default':>7}: "
                  f"{us:7.2f} us  frac {alg / us / 8e6:.3f}", flush=True)
    del a, b, ring


if __name__ == "__main__":
    old = 8168
    bench(4096, 512, [("16", old), ("8", old), ("8", 16384), ("8", 24576), ("0", 0)], reps=3)
    bench(8192, 512, [("16", old), ("8", old), ("8", 32768), ("0", 0)])
    bench(1024, 512, [("16", old), ("8", old), ("0", 0)])
    bench(256, 512, [("16", old), ("0", 0)])
    bench(32768, 256, [("16", old), ("8", old), ("8", 65536), ("0", 0)])
    bench(16384, 256, [("16", old), ("8", old), ("0", 0)])
    bench(4096, 256, [("16", old), ("0", 0)])
    bench(65536, 128, [("16", old), ("0", 0)])
    bench(4096, 512, [("0", old), ("0", 0)], act=False)
    bench(8192, 512, [("0", old), ("0", 0)], act=False)
    bench(16384, 256, [("0", old), ("0", 0)], act=False)
    bench(65536, 128, [("0", old), ("0", 0)], act=False)
    bench(262144, 64, [("0", 0), ("0", 262144)], act=False)

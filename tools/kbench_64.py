#!/usr/bin/env python3
"""Times the 64x64 step kernels alone (carle_ca_step_f32 / _u8, no epilogue): HIP events around back-to-back launches,
ring of 6 action tensors.  Usage: kbench_64.py [N,N,...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd import hip                      # noqa: E402

rule = (1 << 12) | 0xC
ns = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [4096, 65536]
g = torch.Generator(device="cuda").manual_seed(1)
for n in ns:
    for fmt, afmt in (("f32", "f32"), ("u8", "f32"), ("u8", "u8")):
        dt = torch.float32 if fmt == "f32" else torch.uint8
        adt = torch.float32 if afmt == "f32" else torch.uint8
        a = (torch.rand(n, 1, 64, 64, device="cuda", generator=g) < 0.5).to(dt)
        b = torch.empty_like(a)
        ring = [(torch.rand(n, 1, 64, 64, device="cuda", generator=g) <= 0.1).to(adt) for _ in range(6)]
        st = hip.new_status(a.device)
        bufs = [a, b]
        step = hip.ca_step_f32 if fmt == "f32" else hip.ca_step_u8
        for name, fn, alg in (("noact", lambda i: step(bufs[i & 1], None, bufs[(i & 1) ^ 1], rule, st, 0),
                               n * 4096 * 2 * a.element_size()),
                              ("act", lambda i: step(bufs[i & 1], ring[i % 6], bufs[(i & 1) ^ 1], rule, st, i & 1),
                               n * 4096 * (2 * a.element_size() + ring[0].element_size()))):
            for i in range(6):
                fn(i)
            reps = 200 if n <= 8192 else 40
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for i in range(reps):
                fn(i)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / reps * 1e3
            print(f"{fmt} state / {afmt} action N={n:6d} {name:6s}: {us:8.2f} us  {alg / us / 1e6:5.2f} TB/s  frac {alg / us / 8e6:.3f}",
                  flush=True)
        del a, b, ring, bufs
        torch.cuda.empty_cache()

#!/usr/bin/env python3
"""Times the wavefront-per-strip step kernels (256x256 / 512x512, fp32 and uint8; no epilogue) for
several launch sizes (CARLE_WIDE_BLOCKS experiment knob; 0 = the library's default)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd import hip                      # noqa: E402

blocks = sys.argv[1].split(",") if len(sys.argv) > 1 else ["0"]
rule = (1 << 12) | 0xC
g = torch.Generator(device="cuda").manual_seed(1)
for fmt, n, size in (("f32", 4096, 256), ("f32", 1024, 512), ("u8", 4096, 256), ("u8", 4096, 512)):
    dt = torch.float32 if fmt == "f32" else torch.uint8
    a = (torch.rand(n, 1, size, size, device="cuda", generator=g) < 0.5).to(dt)
    b = torch.empty_like(a)
    ring = [(torch.rand(n, 1, 64, 64, device="cuda", generator=g) <= 0.1).to(dt) for _ in range(6)]
    st = hip.new_status(a.device)
    bufs = [a, b]
    step = hip.ca_step_f32 if fmt == "f32" else hip.ca_step_u8
    es = a.element_size()
    for blk in blocks:
        if blk == "0":
            os.environ.pop("CARLE_WIDE_BLOCKS", None)
        else:
            os.environ["CARLE_WIDE_BLOCKS"] = blk
        hip.reload_knobs()                  # the library reads its knobs once
        for name, fn, alg in (("noact", lambda i: step(bufs[i & 1], None, bufs[(i & 1) ^ 1], rule, st, 0), n * size * size * 2 * es),
                              ("act", lambda i: step(bufs[i & 1], ring[i % 6], bufs[(i & 1) ^ 1], rule, st, 0),
                               n * (size * size * 2 + 4096) * es)):
            for i in range(4):
                fn(i)
            reps = 100
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for i in range(reps):
                fn(i)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / reps * 1e3
            print(f"{fmt} N={n} {size}x{size} blocks={blk:>6} {name:6s}: {us:8.2f} us  {alg / us / 1e6:5.2f} TB/s  frac {alg / us / 8e6:.3f}",
                  flush=True)
    del a, b, ring, bufs
    torch.cuda.empty_cache()

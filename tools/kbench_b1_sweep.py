#!/usr/bin/env python3
"""Same-process A/B of the packed one-launch step's cache residency (EXPERIMENTS.md 5.3): the whole state array cacheable
(CARLE_B1_RESIDENT_BYTES=0: until round 4) against a resident slice of R MiB per array with the rest streamed (nt), over
every pair of four separately allocated state buffers -- the two timing levels of BASELINE configs[4] belonged to the pair.
    python tools/kbench_b1_sweep.py [N ...]      (512 x 512, B36/S23, 64 x 64 window)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd import hip                      # noqa: E402

rule = (1 << 12) | (1 << 15) | 0xC
# arguments: N or N@size (default size 512)
for arg in sys.argv[1:] or ["4096"]:
    n, size = (int(x) for x in arg.split("@")) if "@" in arg else (int(arg), 512)
    g = torch.Generator(device="cuda").manual_seed(1)
    words = n * size * size // 64
    bufs = []
    for _ in range(4):
        t = torch.empty(words, dtype=torch.int64, device="cuda")
        t.random_(generator=g)
        bufs.append(t.view(n, 1, size, size // 64))
    ring = [hip.pack_f32_b1((torch.rand(n, 1, 64, 64, device="cuda", generator=g) <= 0.1).float()) for _ in range(6)]
    st = hip.new_status(bufs[0].device)
    reward = torch.empty(n, 1, device="cuda")
    alg = n * (size * size // 4 + 512)

    def timed(a, b, reps=150):
        def call(i):
            hip.env_step_b1(a if i % 2 == 0 else b, ring[i % 6], b if i % 2 == 0 else a, reward, rule, st, i & 1)
        for i in range(6):
            call(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(reps):
            call(i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    for rep in range(2):
        sizes = [int(x) for x in os.environ.get("KBENCH_RESIDENT_MIB", "0,112,96,120").split(",")]
        for name, env in [("all cacheable      " if m == 0 else f"resident {m:3d} MiB   ", {"CARLE_B1_RESIDENT_BYTES": str(m << 20)})
                          for m in sizes] + [("r4 policy          ", {"CARLE_B1_RESIDENT_BYTES": "0", "CARLE_B1_HYBRID_ALL": "0"})]:
            for k in ("CARLE_B1_RESIDENT_BYTES", "CARLE_B1_HYBRID_ALL"):
                os.environ.pop(k, None)
            os.environ.update(env)
            hip.reload_knobs()
            us = [timed(bufs[i], bufs[j]) for i in range(4) for j in range(i + 1, 4)]
            print(f"N={n} {size}x{size} {name}: pairs " + " ".join(f"{u:6.2f}" for u in us) +
                  f"  | min {min(us):.2f} max {max(us):.2f} us = {alg / max(us) / 8e6:.3f} .. {alg / min(us) / 8e6:.3f} of 8 TB/s", flush=True)
    for k in ("CARLE_B1_RESIDENT_BYTES", "CARLE_B1_HYBRID_ALL"):
        os.environ.pop(k, None)
    hip.reload_knobs()
    del bufs, ring
    torch.cuda.empty_cache()

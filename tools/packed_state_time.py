#!/usr/bin/env python3
"""The one-launch 64x64 fp32 step alone, N x 64x64: reading the fp32 state, the same + the packed copy, and reading the packed
copy (carle_env_step_f32_packed).  HIP events over back-to-back launches, action ring beyond the Infinity Cache.
    python tools/packed_state_time.py [N ...]          (CARLE_STREAM_STORE_MIN_BYTES etc. apply)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd import hip                      # noqa: E402

dev = torch.device("cuda", 0)
for n in [int(a) for a in sys.argv[1:]] or [4096, 8192, 65536]:
    g = torch.Generator(device="cuda").manual_seed(1)
    bufs = [(torch.rand(n, 1, 64, 64, device=dev, generator=g) < 0.4).float(), torch.empty(n, 1, 64, 64, device=dev)]
    nr = 6 if n <= 8192 else 2
    ring = [(torch.rand(n, 1, 64, 64, device=dev, generator=g) <= 0.1).float() for _ in range(nr)]
    reward = torch.empty(n, 1, device=dev)
    st = hip.new_status(dev)
    rule = hip.rule_mask([3], [2, 3])
    bits = [hip.pack_f32_b1(bufs[0]), torch.empty(n, 1, 64, 1, dtype=torch.int64, device=dev)]
    for rep in range(2):
        for mode in ("fp32 in", "fp32 in + copy", "packed in + copy"):
            def call(i):
                if mode == "packed in + copy":
                    hip.env_step_f32_packed(bits[i & 1], ring[i % nr], bufs[(i & 1) ^ 1], reward, rule, st, i & 1, True, False,
                                            bits_out=bits[(i & 1) ^ 1])
                else:
                    hip.env_step_f32(bufs[i & 1], ring[i % nr], bufs[(i & 1) ^ 1], reward, rule, st, i & 1, True, False,
                                     bits_out=bits[(i & 1) ^ 1] if "copy" in mode else None)
            for i in range(6):
                call(i)
            reps = 400 if n <= 8192 else 60
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for i in range(reps):
                call(i)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / reps * 1e3
            print(f"N={n} one-launch step, {mode:18s}: {us:8.2f} us  ({n / us:.1f} env-steps/us)", flush=True)

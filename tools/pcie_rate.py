#!/usr/bin/env python3
"""PCIe-inclusive step rate: the reference's agents draw the action on the host and the wrapper copies it to the device
every step (carle/agents.py:38-40, carle/mcl.py:400) -- 64 MiB per step at BASELINE configs[1].  env.step(host tensor)
for pageable and pinned fp32 actions, and for the same toggles handed over as uint8 (16 MiB) or bit-packed (2 MiB)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd import hip                      # noqa: E402
from carle_amd.env import CARLE                # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
g = torch.Generator().manual_seed(1)
ring = [(torch.rand(n, 1, 64, 64, generator=g) <= 0.1).float() for _ in range(4)]


def timed(env, actions, steps=40):
    for i in range(6):
        env.step(actions[i % len(actions)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        env.step(actions[i % len(actions)])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


rows = []
env = CARLE(width=64, height=64, instances=n, device="cuda:0")
env.reset()
env.universe = (torch.rand(n, 1, 64, 64, device="cuda") < 0.5).float()
rows.append(("fp32 action on the device (bench.py's value)", timed(env, [a.cuda() for a in ring]), 0))
rows.append(("fp32 action, pageable host memory", timed(env, ring), ring[0].numel() * 4))
rows.append(("fp32 action, pinned host memory", timed(env, [a.pin_memory() for a in ring]), ring[0].numel() * 4))
env8 = CARLE(width=64, height=64, instances=n, device="cuda:0", storage="u8")
env8.reset()
env8.universe = (torch.rand(n, 1, 64, 64, device="cuda") < 0.5).to(torch.uint8)
rows.append(("uint8 state, uint8 action, pinned host memory", timed(env8, [a.to(torch.uint8).pin_memory() for a in ring]),
             ring[0].numel()))
envb = CARLE(width=64, height=64, instances=n, device="cuda:0", storage="b1")
envb.reset()
packed = [hip.pack_f32_b1(a.cuda()).cpu().pin_memory() for a in ring]
rows.append(("bit-packed state, bit-packed action, pinned host memory", timed(envb, packed), packed[0].numel() * 8))
for name, dt, nbytes in rows:
    bw = f"{nbytes / dt / 1e9:6.1f} GB/s over PCIe" if nbytes else " " * 22
    print(f"N={n} 64x64  {name:58s}: {dt * 1e3:8.3f} ms/step  {n / dt:10.3e} env-steps/s  {bw}", flush=True)

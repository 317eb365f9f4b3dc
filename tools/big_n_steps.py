#!/usr/bin/env python3
"""Per-step host / total time of CARLE.step with a synchronize after every step: looks for host-side stalls
(device allocations of the caching allocator) inside the step loop.  usage: big_n_steps.py [N] [steps]"""
import sys, time, os, gc, torch
if os.environ.get("NOGC"): gc.disable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd.env import CARLE
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
env = CARLE(width=64, height=64, instances=n, device="cuda:0")
env.reset()
env.universe = (torch.rand(n, 1, 64, 64, device="cuda") < 0.5).float()
ring = [(torch.rand(n, 1, 64, 64, device="cuda") <= 0.1).float() for _ in range(6)]
torch.cuda.synchronize()
ts = []
for i in range(steps):
    a0 = torch.cuda.memory_stats()["num_device_alloc"]
    t0 = time.perf_counter()
    env.step(ring[i % 6])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ts.append((i, round((t1 - t0) * 1e3, 3), round((t2 - t0) * 1e3, 3), torch.cuda.memory_stats()["num_device_alloc"] - a0))
slow = [t for t in ts if t[1] > 1.0]
print(f"N={n}: {steps} steps, median host {sorted(t[1] for t in ts)[steps // 2]:.3f} ms, median total {sorted(t[2] for t in ts)[steps // 2]:.3f} ms")
print("steps with > 1 ms of host time (step, host ms, total ms, device allocations during the step):", slow)
print("reserved GiB:", torch.cuda.memory_reserved() / 2**30, "allocated GiB:", torch.cuda.memory_allocated() / 2**30)

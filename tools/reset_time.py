"""Cost of a whole-batch master reset (all-ones action) vs a normal step (development tool)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd.env import CARLE
for n, size in ((4096, 64), (4096, 256)):
    env = CARLE(width=size, height=size, instances=n, device="cuda:0")
    env.reset()
    ones = torch.ones(n, 1, 64, 64, device="cuda")
    rnd = (torch.rand(n, 1, 64, 64, device="cuda") < 0.1).float()
    for name, a in (("normal", rnd), ("reset", ones)):
        for _ in range(5): env.step(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50): env.step(a)
        e1.record(); torch.cuda.synchronize()
        print(f"N={n} {size}x{size} {name} step: {e0.elapsed_time(e1) * 20:.1f} us")

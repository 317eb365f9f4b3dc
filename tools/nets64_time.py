#!/usr/bin/env python3
"""Per-call time of the fused 64x64 reward-net kernels at N=4096: RND2D / AE2D forward (fp32 and packed observations)
and forward + backward."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd import hip
from carle_amd.env import CARLE
from carle_amd.mcl import RND2D, AE2D
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = CARLE(width=64, height=64, instances=n, device="cuda:0")
rnd = RND2D(env); ae = AE2D(rnd)
ae.eval(); ae.reset()
g = (torch.rand(n, 1, 64, 64, device="cuda") < 0.3).float()
gp = hip.pack_f32_b1(g)
for w in (rnd, ae):
    w.batch_size = 10 ** 9
    w.assume_binary_obs = True
    bonus = torch.empty(n, device="cuda")
    reward = torch.zeros(n, 1, device="cuda")
    for obs, oname in ((g, "fp32 obs"), (gp, "packed obs")):
        for name, fn in (("forward", lambda: w._hip_forward(obs, False, reward, bonus, None)),
                         ("forward + backward", lambda: w._fused_train(obs, None, bonus, None))):
            for _ in range(5): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(50): fn()
            e1.record(); torch.cuda.synchronize()
            print(f"N={n} 64x64 {w.my_name} {oname:10s} {name:18s}: {e0.elapsed_time(e1) * 20:8.1f} us per call", flush=True)

#!/usr/bin/env python3
"""Times RandomNetworkAgent.forward (carle_agent_rn_forward_f32: features kernel + MFMA linear kernel) against
the same network run through PyTorch-ROCm modules, at the reference's default geometry (256x256
observations, 64x64 actions) and at 64x64 observations."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd.agents import RandomNetworkAgent
for n, size in ((16, 256), (256, 256), (4096, 64), (1024, 256)):
    agent = RandomNetworkAgent(observation_width=size, observation_height=size).cuda()
    obs = (torch.rand(n, 1, size, size, device="cuda") < 0.3).float()
    def hip_fwd(): return agent(obs)
    def torch_fwd():
        with torch.no_grad():
            return 1.0 * (agent.network(obs) <= agent.toggle_rate)
    for name, fn in (("HIP kernels", hip_fwd), ("torch modules", torch_fwd)):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50
        dense = (size // 4) ** 2
        print(f"N={n:5d} {size}x{size}: {name:13s} {us:9.1f} us per call  ({2.0 * n * dense * 4096 / us / 1e6:7.2f} TFLOP/s in the linear layer, weights {dense * 4096 * 4 / 2**20:.0f} MiB)")

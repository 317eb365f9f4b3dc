"""BASELINE configs[0]: Conway's Life, N=1, 64x64, 256 steps on the CPU port of the reference step
(oracle/torch_port.py = the reference's ATen op sequence), and the same through this build on the GPU."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.torch_port import TorchPortEnv
g = torch.Generator().manual_seed(1234)
acts = [(torch.rand(1, 1, 64, 64, generator=g) <= 0.1).float() for _ in range(4)]
for threads in (1, 16):
    torch.set_num_threads(threads)
    env = TorchPortEnv(1, 64, 64)
    for i in range(8): env.step(acts[i % 4])
    t0 = time.perf_counter()
    for i in range(256): env.step(acts[i % 4])
    dt = time.perf_counter() - t0
    print(f"CPU port, N=1 64x64 B3/S23, 256 steps, {threads} thread(s): {dt * 1e3:.1f} ms -> {256 / dt:.0f} env-steps/s")
if torch.cuda.is_available():
    from carle_amd.env import CARLE
    env = CARLE(width=64, height=64, instances=1, device="cuda:0")
    env.reset()
    dacts = [a.cuda() for a in acts]
    for i in range(16): env.step(dacts[i % 4])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(256): env.step(dacts[i % 4])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"this build, N=1 64x64 B3/S23, 256 steps on the MI355X: {dt * 1e3:.2f} ms -> {256 / dt:.0f} env-steps/s (host-bound)")

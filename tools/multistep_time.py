"""Throughput of carle_ca_multistep_b1 (k action-free generations per launch, grid resident in registers)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd import hip
for size, n, rule, name in ((64, 4096, ([3], [2, 3]), "B3/S23"), (64, 65536, ([3], [2, 3]), "B3/S23"),
                            (256, 4096, ([3], [2, 3]), "B3/S23"), (512, 4096, ([3, 6], [2, 3]), "B36/S23"),
                            (512, 4096, ([3, 6, 8], [2, 4, 5]), "B368/S245 (generic rule path)")):
    mask = hip.rule_mask(*rule)
    u = hip.pack_f32_b1((torch.rand(n // 8, 1, size, size, device="cuda") < 0.4).float()).repeat(8, 1, 1, 1)
    out = torch.empty_like(u)
    for k in (1, 16, 256):
        hip.ca_multistep_b1(u, out, mask, k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20 if k < 256 else 5
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): hip.ca_multistep_b1(u, out, mask, k)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print(f"N={n} {size}x{size} {name} k={k}: {us:.1f} us per launch, {us / k:.2f} us per generation, "
              f"{n * size * size * k / us * 1e6:.3e} cell-updates/s")

#!/usr/bin/env python3
"""Per-step wall time of env.step(action) eager vs replayed as a HIP graph (carle_amd.graphs.StepGraph) for the
launch-bound configurations: BASELINE configs[0] (N=1, 64x64), the reference's default geometry (16 x 256x256) bare
and under the RND2D + AE2D evaluation stack, and small 64x64 batches."""
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd.env import CARLE                 # noqa: E402
from carle_amd.graphs import StepGraph          # noqa: E402
from carle_amd.mcl import AE2D, RND2D, PufferDetector, SpeedDetector           # noqa: E402


def timed(fn, steps):
    for _ in range(32):
        fn()
    torch.cuda.synchronize()
    gc.collect()
    gc.freeze()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    gc.unfreeze()
    return dt


for name, n, size, stack, storage in (("configs[0]: N=1 64x64", 1, 64, False, "f32"),
                                      ("N=16 64x64", 16, 64, False, "f32"),
                                      ("N=256 64x64", 256, 64, False, "f32"),
                                      ("N=4096 64x64 packed", 4096, 64, False, "b1"),
                                      ("N=4096 64x64 uint8", 4096, 64, False, "u8"),
                                      ("reference default: N=16 256x256", 16, 256, False, "f32"),
                                      ("N=16 256x256 + RND2D + AE2D (forward)", 16, 256, True, "f32"),
                                      ("N=16 64x64 + RND2D + AE2D (forward)", 16, 64, True, "f32"),
                                      ("N=16 256x256 evaluation stack (eval.py:83-87)", 16, 256, "eval", "f32"),
                                      ("configs[1]: N=4096 64x64", 4096, 64, False, "f32")):
    torch.manual_seed(0)
    env = CARLE(instances=n, width=size, height=size, device="cuda:0", storage=storage)
    if stack:
        env = AE2D(RND2D(env, train_predictor=False), train_predictor=False)
    if stack == "eval":                          # RND2D -> AE2D -> SpeedDetector -> PufferDetector
        env = PufferDetector(SpeedDetector(env))
        env.eval()
    env.reset()
    ring = [(torch.rand(n, 1, 64, 64, device="cuda") < 0.1).float() for _ in range(6)]
    static = torch.zeros_like(ring[0])
    i = [0]

    def eager():
        i[0] += 1
        env.step(ring[i[0] % 6])

    t_e = timed(eager, 2000)
    env.reset()
    graph = StepGraph(env, static)

    def replay():
        i[0] += 1
        static.copy_(ring[i[0] % 6])
        graph()

    def replay_inplace():                        # an agent that writes its action into the static buffer itself
        graph()

    t_g = timed(replay, 2000)
    t_i = timed(replay_inplace, 2000)
    # the recorded library calls of a step issued again (backend="calls"): a fresh action tensor per call, no copy
    t_c = float("nan")
    try:
        env.reset()
        plan = StepGraph(env, static, backend="calls")

        def replay_calls():
            i[0] += 1
            plan(ring[i[0] % 6])

        t_c = timed(replay_calls, 2000)
    except NotImplementedError:
        pass
    print(f"{name:45s} eager {t_e * 1e6:7.1f} us/step ({n / t_e:9.3e} env-steps/s) | graph + action copy {t_g * 1e6:7.1f} us "
          f"| graph {t_i * 1e6:7.1f} us | recorded calls {t_c * 1e6:7.1f} us ({n / t_c:9.3e} env-steps/s)", flush=True)

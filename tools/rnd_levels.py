#!/usr/bin/env python3
"""Does the one-launch CA + RND2D step (BASELINE configs[3]) have timing levels that belong to the allocations, as the
packed step had (tools/b1_levels.py)?  One process, four separately allocated state buffers, every pair timed over 300
ping-pong steps; then the same with the streaming hint on the state loads (CARLE_RND_CACHED_LOAD_MAX_BYTES=0).
    python tools/rnd_levels.py [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd import hip, _lib, dist as cd   # noqa: E402
from carle_amd.env import CARLE                # noqa: E402
from carle_amd.mcl import RND2D                # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
env = CARLE(width=64, height=64, instances=n, device="cuda:0")
w = RND2D(env, train_predictor=False)
w.eval(); w.reset()
g = torch.Generator(device="cuda").manual_seed(1)
bufs = [(torch.rand(n, 1, 64, 64, device="cuda", generator=g) < 0.5).float() for _ in range(4)]
ring = [(torch.rand(n, 1, 64, 64, device="cuda", generator=g) <= 0.1).float() for _ in range(6)]
env.universe = bufs[0].clone()
w.step(ring[0])
dev = bufs[0].device
pb, rb = w._pred_blob.get(dev), w._rand_blob.get(dev)
rule = env.rule_mask()
reward = torch.empty(n, 1, device=dev)
raw = hip._stream(reward)
alg = n * (64 * 64 * 12 + 4)
for i, b in enumerate(bufs):
    p = b.data_ptr()
    print(f"buffer {i}: 0x{p:012x} (2M page {(p % (2 << 20)) >> 12}, mod 256M {(p % (256 << 20)) >> 20} MiB)")


def timed(a, b, reps=300):
    def call(i):
        hip.env_step_f32_rnd(a if i % 2 == 0 else b, ring[i % 6], b if i % 2 == 0 else a, reward, rule, env._status, i & 1, True,
                             pb, rb, w._ws, w._step_bonus, 1.0, True, None, raw, None, False, deferred=True)
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
    for name, knob in (("cached state loads (library's choice)", None), ("streaming state loads                ", "0")):
        if knob is None:
            os.environ.pop("CARLE_RND_CACHED_LOAD_MAX_BYTES", None)
        else:
            os.environ["CARLE_RND_CACHED_LOAD_MAX_BYTES"] = knob
        hip.reload_knobs()
        us = [timed(bufs[i], bufs[j]) for i in range(4) for j in range(i + 1, 4)]
        print(f"N={n} {name}: pairs " + " ".join(f"{u:6.2f}" for u in us) +
              f"  | {alg / max(us) / 8e6:.3f} .. {alg / min(us) / 8e6:.3f} of 8 TB/s", flush=True)
os.environ.pop("CARLE_RND_CACHED_LOAD_MAX_BYTES", None)

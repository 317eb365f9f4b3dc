#!/usr/bin/env python3
"""What the GPU's clocks and power do under the three BASELINE 64x64 / 512x512 kernels -- the fused CA + RND2D step runs at
36.4-36.7 us on some boxes and 38.3-39.8 us on others with the same library, while the plain step kernel beside it is
31.5-32.0 us everywhere.  Runs each kernel back to back for ~4 s while a thread samples
`rocm-smi --showclocks --showpower` twice a second; prints the kernel's time beside the sampled sclk / mclk / power.
    python tools/clock_probe.py"""
import json
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd import hip, _lib                # noqa: E402
from carle_amd.env import CARLE                # noqa: E402
from carle_amd.mcl import RND2D                # noqa: E402


def sample():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(out)
        card = d[sorted(d)[0]]
        pick = {k: v for k, v in card.items() if any(t in k.lower() for t in ("sclk", "mclk", "fclk", "power"))}
        return pick
    except Exception as e:                       # noqa: BLE001
        return {"error": repr(e)}


n = 4096
torch.manual_seed(0)
env = CARLE(width=64, height=64, instances=n, device="cuda:0")
w = RND2D(env, train_predictor=False)
w.eval(); w.reset()
g = torch.Generator(device="cuda").manual_seed(1)
u = (torch.rand(n, 1, 64, 64, device="cuda", generator=g) < 0.5).float()
env.universe = u.clone()
ring = [(torch.rand(n, 1, 64, 64, device="cuda", generator=g) <= 0.1).float() for _ in range(6)]
w.step(ring[0])
dev = u.device
pb, rb = w._pred_blob.get(dev), w._rand_blob.get(dev)
rule = env.rule_mask()
reward = torch.empty(n, 1, device=dev)
raw = hip._stream(reward)
bufs = [u.clone(), torch.empty_like(u)]
st2 = hip.new_status(dev)


def rnd(i):
    hip.env_step_f32_rnd(bufs[i & 1], ring[i % 6], bufs[(i & 1) ^ 1], reward, rule, env._status, i & 1, True, pb, rb, w._ws,
                         w._step_bonus, 1.0, True, None, raw, None, False, deferred=True)


def plain(i):
    hip.env_step_f32(bufs[i & 1], ring[i % 6], bufs[(i & 1) ^ 1], reward, rule, st2, i & 1, True, False)


print("idle:", sample(), flush=True)
for name, fn in (("plain step kernel (configs[1])", plain), ("fused CA + RND2D step (configs[3])", rnd), ("plain again", plain)):
    samples, stop = [], threading.Event()

    def watch():
        while not stop.is_set():
            samples.append(sample())
            time.sleep(0.4)
    th = threading.Thread(target=watch)
    th.start()
    times = []
    t_end = time.time() + 4.0
    while time.time() < t_end:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(1000):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    stop.set()
    th.join()
    print(f"{name}: {min(times):.2f} .. {max(times):.2f} us per launch over {len(times)} blocks of 1000", flush=True)
    for s_ in samples[1:6]:
        print("    ", s_, flush=True)

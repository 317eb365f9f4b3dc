#!/usr/bin/env python3
"""Times the fused CA + RND2D kernel (carle_ca_step_f32_rnd, no epilogue) alone at N x 64x64 for
several workgroup counts (CARLE_RND_BLOCKS experiment knob): persistent (1024 = 4 per CU) down to
one workgroup per grid.  HIP events around back-to-back launches, action ring beyond the Infinity
Cache."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd import hip                      # noqa: E402
from carle_amd.env import CARLE                # noqa: E402
from carle_amd.mcl import RND2D                # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
settings = sys.argv[2].split(",") if len(sys.argv) > 2 else ["0", "1024", "1280", "2048", "4097"]
staggers = sys.argv[3].split(",") if len(sys.argv) > 3 else ["0"]
torch.manual_seed(0)
env = CARLE(width=64, height=64, instances=n, device="cuda:0")
w = RND2D(env, train_predictor=False)
w.eval(); w.reset()
g = torch.Generator(device="cuda").manual_seed(1)
env.universe = (torch.rand(n, 1, 64, 64, device="cuda", generator=g) < 0.5).float()
ring = [(torch.rand(n, 1, 64, 64, device="cuda", generator=g) <= 0.1).float() for _ in range(6)]
w.step(ring[0])                                 # builds the tables / workspace
dev = env.universe.device
pb, rb = w._pred_blob.get(dev), w._rand_blob.get(dev)
bufs = [env.universe, torch.empty_like(env.universe)]
rule = env.rule_mask()


def launch(i):
    hip.ca_step_f32_rnd(bufs[i & 1], ring[i % 6], bufs[(i & 1) ^ 1], rule, env._status, 0, pb, rb, w._ws,
                        w._step_bonus, True)


alg = n * (64 * 64 * 12 + 4)
for rep in range(2):
  for stg in staggers:
    os.environ["CARLE_RND_STAGGER"] = stg
    for s in settings:
        if s == "0":
            os.environ.pop("CARLE_RND_BLOCKS", None)
        else:
            os.environ["CARLE_RND_BLOCKS"] = s
        hip.reload_knobs()                  # the library reads its knobs once
        for i in range(6):
            launch(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(300):
            launch(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 300 * 1e3
        print(f"N={n} stagger={stg:>2} blocks={s:>5}: {us:6.2f} us  {alg / us / 1e6:5.2f} TB/s  frac {alg / us / 8e6:.3f}", flush=True)

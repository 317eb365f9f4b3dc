#!/usr/bin/env python3
"""Times the whole RND2D.step library call (carle_env_step_f32_rnd) at N x 64x64, back to back: the ONE-launch form
(CARLE_STEP_DEFERRED) against step kernel + epilogue kernel, with / without the reward moments.  HIP events around 300
calls, action ring beyond the Infinity Cache.   python tools/kbench_rnd_env.py [N] [--bits]
--bits: the instantiation that also leaves the observation bit-packed (the hand-off to a reward net stacked on RND2D)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd import hip, _lib, dist as cd   # noqa: E402
from carle_amd.env import CARLE                # noqa: E402
from carle_amd.mcl import RND2D                # noqa: E402

want_bits = "--bits" in sys.argv
argv = [a for a in sys.argv[1:] if a != "--bits"]
n = int(argv[0]) if argv else 4096
torch.manual_seed(0)
env = CARLE(width=64, height=64, instances=n, device="cuda:0")
w = RND2D(env, train_predictor=False)
w.eval(); w.reset()
st = cd.RunningRewardStats("cuda:0", sync_every=1 << 30)
st.attach(w)
g = torch.Generator(device="cuda").manual_seed(1)
env.universe = (torch.rand(n, 1, 64, 64, device="cuda", generator=g) < 0.5).float()
ring = [(torch.rand(n, 1, 64, 64, device="cuda", generator=g) <= 0.1).float() for _ in range(6)]
w.step(ring[0])                                 # builds the tables / workspace
dev = env.universe.device
pb, rb = w._pred_blob.get(dev), w._rand_blob.get(dev)
bufs = [env.universe, torch.empty_like(env.universe)]
rule = env.rule_mask()
reward = torch.empty(n, 1, device=dev)
partials = torch.zeros(_lib.load().carle_rnd_step_partials_doubles(n), dtype=torch.float64, device=dev)
raw = hip._stream(reward)
bits = torch.empty(n, 1, 64, 1, dtype=torch.int64, device=dev) if want_bits else None
alg = n * (64 * 64 * 12 + 4 + (512 if want_bits else 0))


def call(i, deferred, merge):
    hip.env_step_f32_rnd(bufs[i & 1], ring[i % 6], bufs[(i & 1) ^ 1], reward, rule, env._status, i & 1, True, pb, rb, w._ws,
                         w._step_bonus, 1.0, True, None, raw, None, merge, deferred=deferred, partials=partials if merge else None,
                         bits_out=bits if deferred else None)


for rep in range(2):
    for deferred, merge in ((True, True), (True, False), (False, True), (False, False)):
        for i in range(6):
            call(i, deferred, merge)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(300):
            call(i, deferred, merge)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 300 * 1e3
        print(f"N={n} {'bits ' if want_bits and deferred else ''}{'one launch ' if deferred else 'two launches'} moments={int(merge)}: {us:6.2f} us per step  "
              f"{alg / us / 8e6:.3f} of 8 TB/s", flush=True)

#!/usr/bin/env python3
"""The reference's training loop (carle/train_mcl.py:15-89 through carle_amd.harness.train) at its default geometry -- 16
instances of 256x256, AE2D(RND2D(CARLE)) with predictor updates, RandomAgent -- for a rocprofv3 --kernel-trace --stats pass:
which kernels a step launches.  Round 5: the per-step `reward.sum()` (a torch reduction kernel + an allocation per step,
train_mcl.py:68-69) is gone; the step's finish kernel leaves the reward moments in a device ring that is drained once per
inner loop.      python tools/harness_launches.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd import harness                   # noqa: E402
from carle_amd.agents import RandomAgent        # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(0)
t0 = time.time()
rewards = harness.train(lambda **kw: RandomAgent(seed=3, **kw), instances=16, steps=(2, steps), out_dir=None, verbose=False)
torch.cuda.synchronize()
print(f"harness.train: 2 x {steps} steps of 16 x 256x256 in {time.time() - t0:.2f} s; last totals {[round(float(r), 4) for r in rewards[-3:]]}")

#!/usr/bin/env python3
"""Actual differences behind the tolerances of the two reference-trajectory tests (tests/test_gpu_nets.py): rewards per
step and updated weights per tensor against the trajectories recorded from the genuine reference (CPU, fp32 autograd +
torch Adam) at 64x64 and 256x256, 3 Adam updates per wrapper.  Run on the GPU box; output kept under profiles/."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as G                   # noqa: E402
from carle_amd.env import CARLE           # noqa: E402
from carle_amd.mcl import RND2D, AE2D     # noqa: E402


def run(size, n, z, init_key, act_key, rew_key, sd_keys, steps=6):
    torch.manual_seed(0)
    env = CARLE(width=size, height=size, action_width=64, action_height=64, instances=n, device="cuda:0")
    rnd = RND2D(env)
    ae = AE2D(rnd)
    rnd.eval(); ae.eval()
    rnd.batch_size = ae.batch_size = 2
    ae.reset()
    env.universe = torch.from_numpy(G.unpack(z[init_key], (n, 1, size, size))).cuda()
    acts = G.unpack(z[act_key], (steps, n, 1, 64, 64))
    print(f"--- {size}x{size}, N={n}: rewards (RND2D + AE2D bonus), |ours - reference| per step; updates after steps 2, 4, 6")
    for t in range(steps):
        obs, reward, done, info = ae.step(torch.from_numpy(acts[t]))
        ref = z[rew_key][t]
        d = np.abs(reward.cpu().numpy() - ref)
        print(f"step {t}: reward ~{np.abs(ref).max():.3e}  max abs diff {d.max():.2e}  relative {d.max() / np.abs(ref).max():.2e}")
    for prefix, mod in zip(sd_keys, (rnd, ae)):
        ref = G.load_sd(z, prefix)
        for k, v in mod.state_dict().items():
            if "predictor" not in k:
                continue
            d = np.abs(v.cpu().numpy() - ref[k])
            moved = np.abs(ref[k]).max()
            print(f"{mod.my_name:5s} {k:22s} max |w| {moved:.3e}  max abs diff {d.max():.2e}  elements over 1e-5: {(d > 1e-5).sum()} / {d.size}")


z64 = G.nets(64)
run(64, 4, z64, "upd_init", "upd_actions", "upd_rewards", ("upd_rnd_sd", "upd_ae_sd"))
z256 = np.load(os.path.join(G.GOLDEN, "nets_256_train.npz"))
run(256, 2, z256, "init", "actions", "rewards", ("final_rnd_sd", "final_ae_sd"))


def run_prediction(name):
    """PredictionBonus / SurpriseBonus trajectories of tests/golden/prediction.npz (tests/test_gpu_prediction.py)."""
    import carle_amd.mcl as M
    c = G.prediction_case(name)
    env = CARLE(width=c.size, height=c.size, instances=c.n, device="cuda:0")
    w = getattr(M, c.cls)(env)
    w.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in c.init_sd.items()}, strict=True)
    w.eval()
    w.batch_size, w.reward_scale = c.batch_size, c.scale
    w.reset()
    env.universe = torch.from_numpy(c.init).cuda()
    print(f"--- {name}: {c.cls} {c.size}x{c.size}, N={c.n}, batch_size {c.batch_size}, {c.steps} steps")
    worst = 0.0
    for t in range(c.steps):
        obs, reward, done, info = w.step(torch.from_numpy(c.actions[t]))
        d = np.abs(reward.cpu().numpy() - c.rewards[t]).max()
        worst = max(worst, d)
        print(f"step {t}: reward ~{np.abs(c.rewards[t]).max():.3e}  max abs diff {d:.2e}")
    for k, v in w.state_dict().items():
        if "predictor" in k:
            d = np.abs(v.cpu().numpy() - c.final_sd[k])
            print(f"{k:22s} max |w| {np.abs(c.final_sd[k]).max():.3e}  max abs diff {d.max():.2e}")
    print(f"worst reward difference {worst:.2e}")


for nm in G.PREDICTION_CASES:
    run_prediction(nm)

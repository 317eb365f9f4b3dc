#!/usr/bin/env python3
"""Differential fuzzing of the hand-coded reward wrappers (CornerBonus / SpeedDetector / PufferDetector / ParsimonyBonus /
MorphoBonus, carle/mcl.py:86-231, 730-853) against oracle/wrappers.py + the CA oracle, run on the GPU box: random stacks in
random order (incl. PufferDetector directly on a SpeedDetector: the one-launch pair of eval.py:86-87), sizes 64 .. 512, all
three storages, batches of 1 .. 40, action windows of 8 .. 64, growth thresholds of 1 .. 5 with action-free stretches,
master resets.  Rewards within 1e-4 (the SpeedDetector's fp32 centre of mass; exact for the
others), grids bit-exact.  Not part of the test suite (open-ended).      python tests/tools/fuzz_wrappers.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from carle_amd.env import CARLE                # noqa: E402
import carle_amd.mcl as M                      # noqa: E402
from oracle import ca as O                     # noqa: E402
from oracle import wrappers as OW              # noqa: E402



def one_case(case, rng):
    storage = str(rng.choice(["f32", "f32", "u8", "b1"]))
    size = int(rng.choice([64, 64, 128, 256, 512]))
    asize = 64 if storage == "b1" else int(rng.choice([8, 32, 64, 64]))
    n = int(rng.choice([1, 2, 3, 16, 17, 40])) if size <= 128 else int(rng.integers(1, 5))
    env = CARLE(width=size, height=size, action_width=asize, action_height=asize, instances=n, device="cuda:0",
                storage=storage, **({"obs": "f32"} if storage != "f32" else {}))
    kinds = [k for k in ("corner", "speed", "puffer") if rng.random() < 0.7] or ["puffer"]
    if rng.random() < 0.4:
        kinds.append("parsimony")
    if size == 64 and asize == 64 and rng.random() < 0.4:
        kinds.append("morpho")                               # (mcl.py:172: the action must have the universe's shape)
    rng.shuffle(kinds)
    if rng.random() < 0.4 and "speed" in kinds and "puffer" in kinds:        # the one-launch pair: Puffer directly on Speed
        kinds.remove("puffer")
        kinds.insert(kinds.index("speed") + 1, "puffer")
    top, mine = env, []
    for k in kinds:
        if k == "corner":
            top = M.CornerBonus(top); mine.append(OW.CornerBonus(size, size))
        elif k == "speed":
            top = M.SpeedDetector(top); mine.append(OW.SpeedDetector(size, size, asize, asize))
        elif k == "puffer":
            top = M.PufferDetector(top); top.growth_threshold = int(rng.integers(1, 6))
            mine.append(OW.PufferDetector(top.growth_threshold))
        elif k == "morpho":
            top = M.MorphoBonus(top); top.reward_scale = float(rng.choice([1.0, 0.25]))
            mine.append(OW.MorphoBonus(top.target_patterns.cpu().numpy(), top.reward_scale))
        else:
            top = M.ParsimonyBonus(top); mine.append(OW.ParsimonyBonus())
    growing = rng.random() < 0.5
    env.birth, env.survive = ([3, 4, 5], list(range(9))) if growing else ([3], [2, 3])
    rule = O.rule_mask(env.birth, env.survive)
    top.reset()
    u = (rng.random((n, 1, size, size)) < 0.15).astype(np.float32)
    env.universe = torch.from_numpy(u).cuda()
    desc = f"case {case}: {kinds} storage={storage} size={size} asize={asize} n={n}"
    for t in range(int(rng.integers(6, 18))):
        p = 0.0 if rng.random() < 0.6 else float(rng.choice([0.01, 0.2, 1.0]))
        a = (rng.random((n, 1, asize, asize)) < p).astype(np.float32)
        obs, reward, done, info = top.step(torch.from_numpy(a).cuda())
        before = u
        u, reset, _ = O.c_env_step(u, a, rule)
        got_obs = obs.float().cpu().numpy()
        if not np.array_equal(got_obs, u):
            raise AssertionError(f"{desc}: grids differ at step {t}")
        r = np.zeros((n, 1), np.float32)
        for w in mine:
            r = w.step(before if isinstance(w, OW.MorphoBonus) else u, r, a)
        got = reward.detach().float().cpu().numpy().reshape(n, -1)
        # (512 x 512: the weighted sums exceed 2^24, their fp32 rounding depends on the summation order)
        if not np.allclose(got, r.reshape(n, -1), rtol=2e-6, atol=1e-4 if size <= 256 else 2e-3):
            raise AssertionError(f"{desc}: rewards differ at step {t} (p={p}, reset={reset}): got {got.ravel()[:6]} expected "
                                 f"{r.ravel()[:6]}")
    return desc


SEED_MULT = 15485863


def main(budget=60.0, seed0=0, cases=None):
    """Runs cases until `budget` seconds have passed (or exactly `cases` of them: tests/test_gpu_fuzz.py)."""
    t_end = time.time() + budget
    case = 0
    while (case < cases) if cases is not None else (time.time() < t_end):
        d = one_case(case, np.random.default_rng(seed0 * SEED_MULT + case))
        case += 1
        if case % 25 == 0 and cases is None:
            print(f"{case} cases ok (last: {d})", flush=True)
    print(f"fuzz_wrappers: {case} cases, no difference")
    return case


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)

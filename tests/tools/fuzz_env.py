#!/usr/bin/env python3
"""Differential fuzzing of the environment SURFACE against the C oracle (oracle/ca_oracle.c), run on the GPU box:
random geometries (even squares 2 .. 130 and 256 / 512), batches of 1 .. 300 instances, all three storages, random
18-bit rules (changed between steps), actions with arbitrary values (0 / 1, 0.5, -1, inf, NaN, -0.0, denormals),
oversize actions (centre crop, env.py:164-167), all-ones actions (master reset), no-toggle steps, state assignment
and reset() between steps, CARLE.run chunks, an RND2D wrapper on top for 64x64.  Grids bit-exact, counters exact.
Not part of the test suite (open-ended); every failure it ever finds becomes a regression test.
    python tests/tools/fuzz_env.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from carle_amd.env import CARLE                # noqa: E402
from carle_amd.mcl import RND2D                # noqa: E402
from oracle import ca as O                     # noqa: E402

VALUES = np.array([1.0, 1.0, 1.0, 0.5, -1.0, 2.0, np.inf, -np.inf, np.nan, 1e-40, -1e-41], np.float32)


def rand_action(rng, n, asize, kind):
    if kind == "ones":
        return np.ones((n, 1, asize, asize), np.float32)
    if kind == "zeros":
        return np.zeros((n, 1, asize, asize), np.float32)
    if kind == "negzero":
        return np.full((n, 1, asize, asize), -0.0, np.float32)
    p = {"sparse": 0.05, "dense": 0.6, "values": 0.3, "almost": 1.0}[kind]
    a = (rng.random((n, 1, asize, asize)) < p).astype(np.float32)
    if kind == "values":
        a[a != 0] = rng.choice(VALUES, int((a != 0).sum()))
    if kind == "almost":                                     # all ones but one element somewhere
        i = tuple(rng.integers(0, s) for s in a.shape)
        a[i] = rng.choice(np.array([0.0, 2.0, -0.0, np.nan], np.float32))
    return a


def one_case(case, rng):
    storage = rng.choice(["f32", "f32", "u8", "b1"])
    if storage == "b1":
        size = int(rng.choice([64, 128, 256, 512]))
        asize = int(rng.choice([64, size])) if rng.random() < 0.8 else int(rng.choice([a for a in (64, 128, 256, 512) if a <= size]))
    else:
        size = int(rng.choice([64, 64, 256, 512])) if rng.random() < 0.5 else 2 * int(rng.integers(1, 66))
        asize = 2 * int(rng.integers(1, size // 2 + 1))
        if rng.random() < 0.4:
            asize = min(64, size) if size % 2 == 0 else size
    n = int(rng.choice([1, 2, 3, 5, 16, 17, 33, 64, 100, 257])) if size <= 130 else int(rng.integers(1, 7))
    wrap = storage == "f32" and size == 64 and rng.random() < 0.3
    steps = int(rng.integers(3, 12))
    env = CARLE(width=size, height=size, action_width=asize, action_height=asize, instances=n, device="cuda:0",
                storage=str(storage), **({"obs": "f32"} if storage != "f32" else {}))
    top = RND2D(env, train_predictor=False) if wrap else env
    if wrap:
        top.eval()
    top.reset()
    u = (rng.random((n, 1, size, size)) < rng.choice([0.0, 0.1, 0.4, 0.9])).astype(np.float32)
    if storage == "f32" and rng.random() < 0.3:
        u[u != 0] = rng.choice(VALUES, int((u != 0).sum()))      # any non-zero value is alive
    env.universe = torch.from_numpy(u).cuda()
    sn = ssa = resets = 0
    desc = f"case {case}: storage={storage} size={size} asize={asize} n={n} wrap={wrap}"
    t = 0
    while t < steps:
        if rng.random() < 0.5:
            rule = int(rng.integers(0, 1 << 18))
            env.birth = [k for k in range(9) if (rule >> (9 + k)) & 1]
            env.survive = [k for k in range(9) if (rule >> k) & 1]
        rule = O.rule_mask(env.birth, env.survive)
        kind = str(rng.choice(["sparse", "sparse", "dense", "values", "ones", "zeros", "almost", "negzero"]))
        # an oversize action is cropped by (W - AW) // 2 on every side (env.py:164-167): consistent only for an action of
        # the universe's own size
        over = storage != "b1" and not wrap and asize < size and rng.random() < 0.15
        a = rand_action(rng, n, size if over else asize, kind)
        off = (size - asize) // 2
        a_eff = a[:, :, off:size - off, off:size - off] if over else a
        # reference predicates on the UNCROPPED action (env.py:191, 208); kernel predicates: any non-zero / all == 1
        any_nz = bool((a != 0).any())                        # NaN != 0 is True, -0.0 != 0 False: the kernels' test
        all_one = bool((a == 1.0).all())
        use_run = storage == "f32" and not wrap and not over and rng.random() < 0.2 and kind != "values"
        if use_run:
            k = int(rng.integers(1, 4))
            acts = [a_eff] + [rand_action(rng, n, asize, str(rng.choice(["sparse", "zeros", "ones"]))) for _ in range(k - 1)]
            obs = env.run(actions=torch.from_numpy(np.stack(acts)).cuda())[0]
        else:
            acts = [a_eff]
            obs = top.step(torch.from_numpy(a).cuda())[0]
        for j, aj in enumerate(acts):
            if j > 0:
                any_nz, all_one = bool((aj != 0).any()), bool((aj == 1.0).all())
            if not any_nz:
                ssa += 1
            if all_one:                                      # master reset (env.py:208-216)
                u = np.zeros_like(u)
                sn, ssa, resets = 0, 0, resets + 1
            else:
                # (oracle_env_step decides the reset by the reference's sum == count, which values other than 0 / 1 can
                # satisfy without being all ones -- the documented deviation of the default predicates: toggle and
                # generation are taken separately here)
                toggled = O.np_apply_action(u, aj).astype(np.float32)
                u = O.c_env_step(toggled, None, rule)[0]
                sn += 1
        t += len(acts)
        got = obs.float().cpu().numpy()
        if not np.array_equal(got, u):
            bad = np.argwhere(got != u)
            raise AssertionError(f"{desc}: grids differ at step {t} (kind={kind}, over={over}, run={use_run}, rule={rule:#x}): "
                                 f"{len(bad)} cells, first {bad[0].tolist()}")
        if (env.step_number, env.steps_since_action, env.master_resets) != (sn, ssa, resets):
            raise AssertionError(f"{desc}: counters {(env.step_number, env.steps_since_action, env.master_resets)} != "
                                 f"{(sn, ssa, resets)} at step {t} (kind={kind}, over={over}, run={use_run})")
        r = rng.random()
        if r < 0.08:
            top.reset()
            u = np.zeros_like(u)
            sn = ssa = resets = 0                        # reset() zeroes the status block (the build-side reset counter too)
        elif r < 0.16:
            u = (rng.random(u.shape) < 0.3).astype(np.float32)
            env.universe = torch.from_numpy(u).cuda()
    return desc


SEED_MULT = 1000003


def main(budget=60.0, seed0=0, cases=None):
    """Runs cases until `budget` seconds have passed (or exactly `cases` of them: tests/test_gpu_fuzz.py)."""
    t_end = time.time() + budget
    case = 0
    while (case < cases) if cases is not None else (time.time() < t_end):
        d = one_case(case, np.random.default_rng(seed0 * SEED_MULT + case))
        case += 1
        if case % 25 == 0 and cases is None:
            print(f"{case} cases ok (last: {d})", flush=True)
    print(f"fuzz_env: {case} cases, no difference")
    return case


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)

#!/usr/bin/env python3
"""Differential fuzzing of the reward-network wrappers STEPPING on the environment (RND2D.step / AE2D.step,
carle/mcl.py:398-407) against the CA oracle + oracle/nets.py, run on the GPU box: stacks RND2D / AE2D / AE2D(RND2D) over
CARLE in all three storages, sizes 64 / 128 / 256, batches of 1 .. 300 (64x64) or 1 .. 20, forward only -- the paths the
surface takes by itself: the fused CA + RND2D kernel (one launch for N > 16), the tiled pair kernel behind the one-launch
wide step (bit-packed copy of the new state), packed state, uint8 state, the stand-alone kernels otherwise.  Rewards within
1e-5 of the oracle's (relative beyond 1), grids bit-exact, master resets included (every reward then equals the all-dead
grid's).  Not part of the test suite (open-ended).      python tests/tools/fuzz_stack.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from carle_amd.env import CARLE                # noqa: E402
import carle_amd.mcl as M                      # noqa: E402
from oracle import ca as O                     # noqa: E402
from oracle import nets as ON                  # noqa: E402



def one_case(case, rng):
    storage = str(rng.choice(["f32", "f32", "u8", "b1"]))
    size = int(rng.choice([64, 64, 128, 256]))
    n = int(rng.choice([1, 3, 16, 17, 40, 100, 300])) if size == 64 else int(rng.choice([1, 2, 5, 16, 20]))
    asize = 64 if storage == "b1" or rng.random() < 0.7 else int(rng.choice([8, 32]))
    kind = str(rng.choice(["rnd", "ae", "rnd+ae"]))
    torch.manual_seed(int(rng.integers(0, 2 ** 31 - 1)))
    env = CARLE(width=size, height=size, action_width=asize, action_height=asize, instances=n, device="cuda:0",
                storage=storage, **({"obs": "f32"} if storage != "f32" else {}))
    kw = dict(train_predictor=False)
    top = env
    nets = []
    if "rnd" in kind:
        top = M.RND2D(top, **kw); nets.append(("rnd", top))
    if "ae" in kind:
        top = M.AE2D(top, **kw); nets.append(("ae", top))
    for _, w in nets:
        w.reward_scale = float(rng.choice([1.0, 0.5, 2.0]))
    top.eval(); top.reset()
    rule_kind = rng.random()
    env.birth, env.survive = ([3], [2, 3]) if rule_kind < 0.6 else ([3, 6], [2, 3]) if rule_kind < 0.8 else ([3, 4, 5], [4, 5, 6, 7])
    rule = O.rule_mask(env.birth, env.survive)
    u = (rng.random((n, 1, size, size)) < rng.choice([0.05, 0.3, 0.6])).astype(np.float32)
    env.universe = torch.from_numpy(u).cuda()
    sds = {name: {k: v.detach().cpu().numpy() for k, v in w.state_dict().items()} for name, w in nets}
    desc = f"case {case}: {kind} storage={storage} size={size} asize={asize} n={n}"
    for t in range(int(rng.integers(3, 8))):
        p = float(rng.choice([0.0, 0.1, 0.1, 0.4, 1.0]))
        a = (rng.random((n, 1, asize, asize)) < p).astype(np.float32)
        obs, reward, done, info = top.step(torch.from_numpy(a).cuda())
        u, reset, _ = O.c_env_step(u, a, rule)
        if not np.array_equal(obs.float().cpu().numpy(), u):
            raise AssertionError(f"{desc}: grids differ at step {t}")
        exp = np.zeros(n, np.float32)
        with torch.no_grad():
            for name, w in nets:
                b = (ON.rnd_bonus if name == "rnd" else ON.ae_bonus)(torch.from_numpy(u), sds[name]).numpy()
                exp = exp + np.float32(w.reward_scale) * b
        got = reward.detach().float().cpu().numpy().reshape(-1)
        tol = 1e-5 * np.maximum(1.0, np.abs(exp)) * len(nets)
        if not bool((np.abs(got - exp) <= tol).all()):
            i = int(np.argmax(np.abs(got - exp) - tol))
            raise AssertionError(f"{desc}: reward of instance {i} at step {t} is {got[i]:.8g}, oracle {exp[i]:.8g} (p={p}, reset={reset})")
    return desc


SEED_MULT = 32452843


def main(budget=60.0, seed0=0, cases=None):
    """Runs cases until `budget` seconds have passed (or exactly `cases` of them: tests/test_gpu_fuzz.py)."""
    t_end = time.time() + budget
    case = 0
    while (case < cases) if cases is not None else (time.time() < t_end):
        d = one_case(case, np.random.default_rng(seed0 * SEED_MULT + case))
        case += 1
        if case % 20 == 0 and cases is None:
            print(f"{case} cases ok (last: {d})", flush=True)
    print(f"fuzz_stack: {case} cases, no difference")
    return case


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)

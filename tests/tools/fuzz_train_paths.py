#!/usr/bin/env python3
"""Self-consistency fuzzing of the TRAINING step of the wrapper stack (train_mcl.py:34-38, 64-69), run on the GPU box:
AE2D(RND2D(CARLE)) in training mode stepped twice from the same seed -- once through whatever fast path the surface picks
(the paired four-launch call at tiled sizes, the one-launch forward pair, the fused 64x64 kernels, packed copies of the
state) and once with the pairing switched off (each wrapper steps by itself) -- must give the same rewards and the same
weights after several update windows: bit-identical where both runs use the same kernels per net (the pairing only
changes how launches are grouped), within 2e-6 otherwise.  Random sizes 64 / 128 / 192 / 256, storages, batches, window
lengths 1 .. 4, reward scales, master resets inside windows.  And against oracle/nets.py: the rewards of the FIRST window
(before any update) within 1e-5.  Not part of the test suite (open-ended).   python tests/tools/fuzz_train_paths.py [seconds] [seed]"""
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



def build(seed, storage, size, n, bs, scales, paired):
    torch.manual_seed(seed)
    env = CARLE(width=size, height=size, instances=n, device="cuda:0", storage=storage,
                **({"obs": "f32"} if storage != "f32" else {}))
    rnd = M.RND2D(env)
    ae = M.AE2D(rnd)
    rnd.batch_size, ae.batch_size = bs
    rnd.reward_scale, ae.reward_scale = scales
    if not paired:
        ae._can_pair_step = lambda: 0                        # each wrapper steps by itself
    ae.eval(); ae.reset()
    return env, rnd, ae


def one_case(case, rng):
    storage = str(rng.choice(["f32", "f32", "u8", "b1"]))
    size = int(rng.choice([64, 128, 192, 256]))
    if storage == "b1" and size == 192:
        size = 128                                           # packed state: 64 / 128 / 256 / 512 only
    n = int(rng.choice([1, 3, 16, 17, 40])) if size == 64 else int(rng.choice([1, 2, 5, 16]))
    bs = (int(rng.integers(1, 5)), int(rng.integers(1, 5)))
    scales = (float(rng.choice([1.0, 0.5])), float(rng.choice([1.0, 2.0])))
    seed = int(rng.integers(0, 2 ** 31 - 1))
    steps = int(rng.integers(4, 10))
    u0 = (rng.random((n, 1, size, size)) < 0.3).astype(np.float32)
    acts = [(rng.random((n, 1, 64, 64)) < float(rng.choice([0.0, 0.1, 0.1, 1.0]))).astype(np.float32) for _ in range(steps)]
    desc = f"case {case}: storage={storage} size={size} n={n} batch_size={bs} scales={scales}"
    runs = []
    for paired in (True, False):
        env, rnd, ae = build(seed, storage, size, n, bs, scales, paired)
        env.universe = torch.from_numpy(u0).cuda()
        sd0 = {k: v.detach().cpu().numpy().copy() for k, v in ae.state_dict().items()} if paired else None
        rewards = []
        for a in acts:
            obs, reward, done, info = ae.step(torch.from_numpy(a).cuda())
            rewards.append(reward.detach().float().cpu().numpy().reshape(-1).copy())
        runs.append((rewards, {k: v.detach().cpu().numpy().copy() for k, v in ae.state_dict().items()},
                     obs.float().cpu().numpy(), sd0))
    (ra, sa, oa, sd0), (rb, sb, ob, _) = runs
    assert np.array_equal(oa, ob), desc
    for t, (x, y) in enumerate(zip(ra, rb)):
        if not np.allclose(x, y, rtol=2e-6, atol=2e-7):
            raise AssertionError(f"{desc}: rewards of the paired and the sequential run differ at step {t}: {np.abs(x - y).max():.3e}")
    for k in sa:
        if not np.allclose(sa[k], sb[k], rtol=2e-6, atol=2e-7):
            raise AssertionError(f"{desc}: {k} differs after {steps} steps: {np.abs(sa[k] - sb[k]).max():.3e}")
    # the first window against the oracle (no update has happened before min(batch sizes) steps are complete)
    life = O.rule_mask([3], [2, 3])
    u = u0
    rsd = {k[len("env."):]: v for k, v in sd0.items() if k.startswith("env.")}
    for t in range(min(min(bs), steps)):
        u, reset, _ = O.c_env_step(u, acts[t], life)
        with torch.no_grad():
            exp = np.float32(scales[0]) * ON.rnd_bonus(torch.from_numpy(u), rsd).numpy() + \
                np.float32(scales[1]) * ON.ae_bonus(torch.from_numpy(u), sd0).numpy()
        if not np.allclose(ra[t], exp, rtol=1e-5, atol=2e-5):
            raise AssertionError(f"{desc}: reward at step {t} differs from the oracle by {np.abs(ra[t] - exp).max():.3e}")
    return desc


SEED_MULT = 49979687


def main(budget=60.0, seed0=0, cases=None):
    """Runs cases until `budget` seconds have passed (or exactly `cases` of them: tests/test_gpu_fuzz.py)."""
    t_end = time.time() + budget
    case = 0
    while (case < cases) if cases is not None else (time.time() < t_end):
        d = one_case(case, np.random.default_rng(seed0 * SEED_MULT + case))
        case += 1
        if case % 10 == 0 and cases is None:
            print(f"{case} cases ok (last: {d})", flush=True)
    print(f"fuzz_train_paths: {case} cases, no difference")
    return case


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)

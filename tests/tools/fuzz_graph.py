#!/usr/bin/env python3
"""Self-consistency fuzzing of carle_amd.graphs.StepGraph (both backends) against the eager loop, run on the GPU box: a twin
pair of environments from the same seed -- one stepped eagerly, one through a StepGraph -- is driven through a random
sequence of replayed steps, eager steps in between, reset(), assigned universes, rule changes, master resets, with random
storages, sizes, batches and forward-only wrapper stacks (RND2D, AE2D(RND2D), + SpeedDetector / PufferDetector); reward
statistics attached at random.  Observations bit-identical, rewards within 1e-6, counters and statistics equal after
every step.  Not part of the test suite (open-ended).      python tests/tools/fuzz_graph.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from carle_amd import dist as cd               # noqa: E402
from carle_amd.env import CARLE                # noqa: E402
from carle_amd.graphs import StepGraph         # noqa: E402
import carle_amd.mcl as M                      # noqa: E402



def build(seed, storage, size, n, stack, with_stats):
    torch.manual_seed(seed)
    env = CARLE(width=size, height=size, instances=n, device="cuda:0", storage=storage,
                **({"obs": "f32"} if storage != "f32" else {}))
    top = env
    kw = dict(train_predictor=False)
    for k in stack:
        if k == "rnd":
            top = M.RND2D(top, **kw)
        elif k == "ae":
            top = M.AE2D(top, **kw)
        elif k == "speed":
            top = M.SpeedDetector(top)
        elif k == "puffer":
            top = M.PufferDetector(top); top.growth_threshold = 3
    if top is not env:
        top.eval()
    stats = None
    if with_stats:
        stats = cd.RunningRewardStats("cuda:0", sync_every=1 << 30)
        stats.attach(top)
    top.reset()
    return env, top, stats


def one_case(case, rng):
    storage = str(rng.choice(["f32", "f32", "u8", "b1"]))
    size = int(rng.choice([64, 64, 128, 256]))
    n = int(rng.choice([1, 3, 16, 17, 40])) if size == 64 else int(rng.choice([1, 2, 5, 16]))
    stacks = [[], ["rnd"], ["rnd", "ae"], ["rnd", "ae", "speed", "puffer"], ["speed"], ["puffer"], ["speed", "puffer"]]
    stack = stacks[int(rng.integers(0, len(stacks)))]
    backend = str(rng.choice(["graph", "calls"]))
    with_stats = bool(rng.random() < 0.4)
    seed = int(rng.integers(0, 2 ** 31 - 1))
    desc = f"case {case}: backend={backend} storage={storage} size={size} n={n} stack={stack} stats={with_stats}"
    env_a, top_a, st_a = build(seed, storage, size, n, stack, with_stats)
    env_b, top_b, st_b = build(seed, storage, size, n, stack, with_stats)
    u0 = (torch.rand(n, 1, size, size, device="cuda", generator=torch.Generator(device="cuda").manual_seed(seed)) < 0.3).float()
    env_a.universe, env_b.universe = u0.clone(), u0.clone()
    pre = int(rng.integers(0, 3))                            # eager steps before the graph is built (pending one-launch steps)
    gen = torch.Generator(device="cuda").manual_seed(seed + 1)

    def action(p):
        return (torch.rand(n, 1, 64, 64, device="cuda", generator=gen) < p).float()

    def check(t, ra, rb, oa, ob):
        if not torch.equal(oa.float(), ob.float()):
            raise AssertionError(f"{desc}: observations differ at {t}")
        if not torch.allclose(ra, rb, rtol=1e-6, atol=1e-6):
            raise AssertionError(f"{desc}: rewards differ at {t}: {(ra - rb).abs().max().item():.3e}")
        ca = (env_a.step_number, env_a.steps_since_action, env_a.master_resets)
        cb = (env_b.step_number, env_b.steps_since_action, env_b.master_resets)
        if ca != cb:
            raise AssertionError(f"{desc}: counters {ca} != {cb} at {t}")

    for t in range(pre):
        a = action(0.1)
        oa, ra = top_a.step(a)[:2]
        ob, rb = top_b.step(a)[:2]
        check(f"pre {t}", ra, rb, oa, ob)
    static = torch.zeros(n, 1, 64, 64, device="cuda")
    try:
        graph = StepGraph(top_b, static, backend=backend)
    except NotImplementedError:
        return desc + " (refused)"
    for t in range(int(rng.integers(5, 14))):
        r = rng.random()
        if r < 0.08:
            top_a.reset(); top_b.reset()
            continue
        if r < 0.16:
            pat = (torch.rand(n, 1, size, size, device="cuda", generator=gen) < 0.2).float()
            env_a.universe, env_b.universe = pat.clone(), pat.clone()
            continue
        if r < 0.24:
            rule = "B36/S23" if env_a.birth == [3] else "B3/S23"
            top_a.rules_from_string(rule); top_b.rules_from_string(rule)
        a = action(float(rng.choice([0.0, 0.1, 0.1, 1.0])))
        oa, ra = top_a.step(a)[:2]
        if rng.random() < 0.2:
            ob, rb = top_b.step(a)[:2]                       # an eager step between replays
        elif backend == "calls" and rng.random() < 0.5:
            ob, rb = graph(a)[:2]
        else:
            graph.action.copy_(a)
            ob, rb = graph()[:2]
        check(t, ra, rb, oa, ob)
    if with_stats:
        env_a.settled_status(); env_b.settled_status()
        pa, pb = st_a._pending.cpu().numpy(), st_b._pending.cpu().numpy()
        if pa[0] != pb[0] or not np.allclose(pa, pb, rtol=1e-10, atol=1e-12):
            raise AssertionError(f"{desc}: reward statistics differ: {pa} != {pb}")
    return desc


SEED_MULT = 67867967


def main(budget=60.0, seed0=0, cases=None):
    """Runs cases until `budget` seconds have passed (or exactly `cases` of them: tests/test_gpu_fuzz.py)."""
    t_end = time.time() + budget
    case = 0
    while (case < cases) if cases is not None else (time.time() < t_end):
        d = one_case(case, np.random.default_rng(seed0 * SEED_MULT + case))
        case += 1
        if case % 10 == 0 and cases is None:
            print(f"{case} cases ok (last: {d})", flush=True)
    print(f"fuzz_graph: {case} cases, no difference")
    return case


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)

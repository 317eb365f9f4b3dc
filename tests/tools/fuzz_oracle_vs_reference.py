#!/usr/bin/env python3
"""Differential fuzzing of the CPU ORACLE (oracle/ca.py: C restatement + independent numpy restatement) and of the RLE
codec against the GENUINE reference, in the build container only (the reference is mounted read-only at /root/reference
and never travels; import recipe as tools/make_golden.py).  The oracle is pinned by the committed golden fixtures; this
widens the pin with open-ended random cases: sizes 4 .. 40 (even squares), 1 .. 3 instances, random rules given as the
reference takes them (lists of counts, duplicates and single-element lists included), action values 0 / 1 / 0.5 / -1 /
2 / NaN / inf / -0.0, oversize actions (centre crop), all-ones actions, values that cancel or average to exactly 1.0
(env.py:191, 208) -- grids, reset decision, step_number, steps_since_action after every step; get_rle / rle_to_grid of
random grids byte for byte; the hand-coded wrappers and PredictionBonus / SurpriseBonus (with predictor updates, eval
mode) in random configurations against oracle/wrappers.py / oracle/prediction.py.  Any difference found becomes a golden
fixture.

    cd /tmp && MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference \
        python /root/repo/tests/tools/fuzz_oracle_vs_reference.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
import make_golden as MG                        # noqa: E402   (import_reference)

torch, ref_env, ref_mcl = MG.import_reference()
sys.path.append("/root/repo")
from oracle import ca as O                      # noqa: E402
from oracle import wrappers as OW               # noqa: E402
from oracle import prediction as OP             # noqa: E402
from carle_amd import rle as R                  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
VALUES = np.array([1.0, 1.0, 0.5, -1.0, 2.0, np.inf, np.nan, -0.0], np.float32)


def ref_lists(rng):
    """birth / survive as a caller of the reference may assign them (train_mcl.py:56-57, eval.py:58-59)."""
    def one():
        k = int(rng.integers(1, 6))
        vals = [int(v) for v in rng.integers(0, 9, k)]
        return vals if rng.random() < 0.3 else sorted(set(vals))
    return one(), one()


def one_case(case, rng):
    size = 2 * int(rng.integers(2, 21))
    asize = 2 * int(rng.integers(1, size // 2 + 1))
    n = int(rng.integers(1, 4))
    env = ref_env.CARLE(width=size, height=size, action_width=asize, action_height=asize, instances=n, device="cpu")
    env.reset()
    u = (rng.random((n, 1, size, size)) < rng.choice([0.1, 0.4, 0.8])).astype(np.float32)
    env.universe = torch.from_numpy(u.copy())
    sn = ssa = 0
    for t in range(int(rng.integers(2, 8))):
        birth, survive = ref_lists(rng)
        if len(set(birth)) < len(birth) and len(birth) >= 3 or len(set(survive)) < len(survive) and len(survive) >= 3:
            birth, survive = sorted(set(birth)), sorted(set(survive))    # duplicates in >= 3-element lists give 2.0 cells
        env.birth, env.survive = list(birth), list(survive)
        rule = O.rule_mask(birth, survive)
        kind = str(rng.choice(["sparse", "values", "ones", "zeros", "cancel", "mean1"]))
        over = asize < size and rng.random() < 0.2
        a_sz = size if over else asize
        a = (rng.random((n, 1, a_sz, a_sz)) < 0.2).astype(np.float32)
        if kind == "values":
            a[a != 0] = rng.choice(VALUES, int((a != 0).sum()))
        elif kind == "ones":
            a[:] = 1.0
        elif kind == "zeros":
            a[:] = 0.0
        elif kind == "cancel":                               # toggles whose values sum to zero
            a[:] = 0.0
            a[0, 0, 0, 0], a[0, 0, 0, 1] = 1.0, -1.0
        elif kind == "mean1":                                # values averaging to exactly 1.0 without all being 1
            a[:] = 1.0
            a[0, 0, 0, 0], a[0, 0, 0, 1] = 0.5, 1.5
        off = (size - asize) // 2
        a_eff = a[:, :, off:size - off, off:size - off] if over else a
        with np.errstate(all="ignore"):
            s = float(torch.sum(torch.from_numpy(a)))
            reset_ref_pred = bool(torch.mean(torch.from_numpy(a)) == 1.0)
        obs, reward, done, info = env.step(torch.from_numpy(a.copy()))
        # the oracle gets the cropped action; the reference's predicates look at the uncropped tensor (env.py:191, 208)
        exp_c, reset_c, flags = O.c_env_step(u, a_eff, rule)
        exp_np, reset_np = O.np_env_step(u, a_eff, rule)
        if over:
            if reset_ref_pred and not reset_c:
                exp_c, exp_np, reset_c, reset_np = np.zeros_like(u), np.zeros_like(u), True, True
            elif not reset_ref_pred and reset_c:
                continue_case = True                          # cropped all ones, uncropped not: no oracle entry point; skip the step
                env.universe = torch.from_numpy(u.copy())
                env.step_number, env.steps_since_action = sn, ssa
                continue
        got = obs.detach().numpy()
        if not (np.array_equal(got, exp_c) and np.array_equal(got, exp_np)):
            raise AssertionError(f"case {case} step {t}: grids differ (size {size}, action {a_sz}, kind {kind}, over {over}, "
                                 f"birth {birth}, survive {survive})")
        assert reset_c == reset_np
        if not over:
            assert reset_c == reset_ref_pred, (case, t, kind)
        if not bool(s):
            ssa += 1
        if reset_c:
            sn, ssa = 0, 0
        else:
            sn += 1
        if (env.step_number, env.steps_since_action) != (sn, ssa):
            raise AssertionError(f"case {case} step {t}: counters {(env.step_number, env.steps_since_action)} != {(sn, ssa)} "
                                 f"(kind {kind}, over {over})")
        u = got.copy()
    # RLE: this build's codec against the reference's, byte for byte, on the final grid and on a random one
    for grid in (u[0, 0], (rng.random((size, size)) < rng.choice([0.02, 0.3, 0.97])).astype(np.float32)):
        ref_rle = env.get_rle(torch.from_numpy(grid.copy()))
        body = ref_rle.split("\n", 3)[3]
        mine = R.encode_body(grid)
        if mine != body:
            raise AssertionError(f"case {case}: RLE body differs (size {size})")
        back_ref = env.rle_to_grid(body).numpy()
        back = R.decode_body(body, size, size).astype(np.float32)
        if not np.array_equal(back, back_ref):
            raise AssertionError(f"case {case}: rle_to_grid differs (size {size})")


def wrapper_case(case, rng):
    """oracle/wrappers.py against the reference's CornerBonus / SpeedDetector / PufferDetector / ParsimonyBonus stacked
    in random order over a reference CARLE (0 / 1 actions, with action-free stretches for the growth detector)."""
    size = int(rng.choice([64, 64, 96, 128]))
    n = int(rng.integers(1, 4))
    env = ref_env.CARLE(width=size, height=size, instances=n, device="cpu")
    kinds = [k for k in ("corner", "speed", "puffer") if rng.random() < 0.7] or ["speed"]
    if n == 1 and rng.random() < 0.5:
        kinds.append("parsimony")                            # (N > 1: the reference's division broadcasts to (N, N))
    rng.shuffle(kinds)
    top, mine = env, []
    for k in kinds:
        if k == "corner":
            top = ref_mcl.CornerBonus(top); mine.append(OW.CornerBonus(size, size))
        elif k == "speed":
            top = ref_mcl.SpeedDetector(top); mine.append(OW.SpeedDetector(size, size, 64, 64))
        elif k == "puffer":
            top = ref_mcl.PufferDetector(top); top.growth_threshold = int(rng.integers(1, 5))
            mine.append(OW.PufferDetector(top.growth_threshold))
        else:
            top = ref_mcl.ParsimonyBonus(top); mine.append(OW.ParsimonyBonus())
    env.birth, env.survive = [[3], [2, 3]] if rng.random() < 0.5 else [[3, 4, 5], [0, 1, 2, 3, 4, 5, 6, 7, 8]]   # Life / a growing rule
    top.reset()
    u = (rng.random((n, 1, size, size)) < 0.2).astype(np.float32)
    env.universe = torch.from_numpy(u.copy())
    rule = O.rule_mask(env.birth, env.survive)
    for t in range(int(rng.integers(6, 16))):
        a = (rng.random((n, 1, 64, 64)) < (0.0 if rng.random() < 0.6 else rng.choice([0.01, 0.2]))).astype(np.float32)
        obs, reward, done, info = top.step(torch.from_numpy(a.copy()))
        u, reset, _ = O.c_env_step(u, a, rule)
        assert np.array_equal(obs.detach().numpy(), u)
        r = np.zeros((n, 1), np.float32)
        for w in mine:
            r = w.step(u, r, a)
        got = reward.detach().numpy().reshape(n, -1)
        if not np.allclose(got, r.reshape(n, -1), rtol=1e-6, atol=2e-5):
            raise AssertionError(f"wrapper case {case} step {t}: rewards differ ({kinds}, size {size}, n {n}): {got.ravel()} != {r.ravel()}")


def prediction_case(case, rng):
    """oracle/prediction.py against the reference's PredictionBonus / SurpriseBonus (mcl.py:558-728) in eval mode with
    predictor updates: random sizes (multiples of 4: the decoder's geometry), batches, windows, scales, empty instances."""
    import contextlib
    import io
    cls = str(rng.choice(["PredictionBonus", "SurpriseBonus"]))
    size = 4 * int(rng.integers(2, 17))
    asize = 2 * int(rng.integers(1, size // 2 + 1))
    n = int(rng.integers(1, 4))
    bs = int(rng.integers(1, 4))
    scale = float(rng.choice([1.0, 0.5, 2.0]))
    torch.manual_seed(int(rng.integers(0, 2 ** 31 - 1)))
    with contextlib.redirect_stdout(io.StringIO()):
        env = ref_env.CARLE(width=size, height=size, action_width=asize, action_height=asize, instances=n, device="cpu")
        w = getattr(ref_mcl, cls)(env)
    w.eval()
    w.batch_size, w.reward_scale = bs, scale
    w.reset()
    sd = {k: v.detach().numpy().copy() for k, v in w.state_dict().items()}
    ow = getattr(OP, cls)(sd, batch_size=bs, reward_scale=scale)
    u = (rng.random((n, 1, size, size)) < 0.3).astype(np.float32)
    empty = rng.random(n) < 0.3
    u[empty] = 0
    env.universe = torch.from_numpy(u.copy())
    rule = O.rule_mask([3], [2, 3])
    for t in range(min(3 * bs + 1, int(rng.integers(3, 10)))):
        a = (rng.random((n, 1, asize, asize)) < float(rng.choice([0.0, 0.1, 0.3]))).astype(np.float32)
        a[empty] = 0
        obs, reward, done, info = w.step(torch.from_numpy(a.copy()))
        u, reset, _ = O.c_env_step(u, a, rule)
        assert np.array_equal(obs.detach().numpy(), u)
        exp = ow.step(u, np.zeros((n, 1), np.float32))
        got = reward.detach().numpy().reshape(n, -1)
        if not np.allclose(got, exp, rtol=0, atol=5e-6):
            raise AssertionError(f"prediction case {case} step {t}: {cls} size {size} n {n} batch_size {bs}: "
                                 f"{got.ravel()} != {exp.ravel()}")
        assert float(w.live_cells) == ow.live_cells


def morpho_case(case, rng):
    """oracle/wrappers.py::MorphoBonus against the reference's MorphoBonus (mcl.py:107-195) -- driven through a subclass
    whose default patterns are the two glider files this package ships (the reference's own are missing from its tree) --
    plus a random extra pattern written as RLE: pattern tensors bit for bit, rewards within 1e-5 of the window's weight."""
    import contextlib
    import io
    import tempfile
    pkg = "/root/repo/carle_amd"

    class Morpho(ref_mcl.MorphoBonus):
        def add_default_patterns(self):
            self.add_rle_pattern(os.path.join(pkg, "glider_1.rle"))
            self.add_rle_pattern(os.path.join(pkg, "glider_2.rle"))

    n = int(rng.integers(1, 4))
    with contextlib.redirect_stdout(io.StringIO()):
        env = ref_env.CARLE(width=64, height=64, instances=n, device="cpu")
        w = Morpho(env)
    grids = []
    for name in ("glider_1.rle", "glider_2.rle"):
        body = env.read_rle(os.path.join(pkg, name))
        grids.append(env.rle_to_grid(body).numpy())
    if rng.random() < 0.7:
        g = np.zeros((64, 64), np.float32)
        ph, pw = int(rng.integers(1, 7)), int(rng.integers(1, 7))
        g[:ph, :pw] = (rng.random((ph, pw)) < 0.6)
        if g.sum() == 0:
            g[0, 0] = 1
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "p.rle")
            with open(path, "w") as f:
                f.write(f"x = {pw}, y = {ph}, rule = B3/S23\n" + R.encode_body(g) + "\n")
            w.add_rle_pattern(path)
            grids.append(env.rle_to_grid(env.read_rle(path)).numpy())
    pats = np.concatenate([OW.MorphoBonus.make_patterns(g) for g in grids])
    if not np.array_equal(pats, w.target_patterns.numpy()):
        raise AssertionError(f"morpho case {case}: pattern tensors differ")
    w.reward_scale = float(rng.choice([1.0, 0.5]))
    ow = OW.MorphoBonus(pats, w.reward_scale)
    w.reset()
    u = env.universe.numpy().copy()
    rule = O.rule_mask(env.birth, env.survive)
    for t in range(int(rng.integers(2, 6))):
        a = (rng.random((n, 1, 64, 64)) < float(rng.choice([0.0, 0.02, 0.3]))).astype(np.float32)
        obs, reward, done, info = w.step(torch.from_numpy(a.copy()))
        exp = ow.step(u, np.zeros((n, 1), np.float32), a)
        u, reset, _ = O.c_env_step(u, a, rule)
        assert np.array_equal(obs.detach().numpy(), u)
        got = reward.detach().numpy().reshape(n, -1)
        if not np.allclose(got, exp, rtol=1e-6, atol=2e-4):
            raise AssertionError(f"morpho case {case} step {t}: {got.ravel()} != {exp.ravel()}")


t_end = time.time() + budget
case = 0
while time.time() < t_end:
    one_case(case, np.random.default_rng(seed0 * 99991 + case))
    if case % 4 == 0:
        wrapper_case(case, np.random.default_rng(seed0 * 77773 + case))
    if case % 8 == 0:
        prediction_case(case, np.random.default_rng(seed0 * 55529 + case))
    if case % 16 == 0:
        morpho_case(case, np.random.default_rng(seed0 * 33331 + case))
    case += 1
    if case % 200 == 0:
        print(f"{case} cases ok", flush=True)
print(f"fuzz_oracle_vs_reference: {case} cases, no difference")

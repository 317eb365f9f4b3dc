#!/usr/bin/env python3
"""Generate the golden parity fixtures under tests/golden/ from the GENUINE reference.

TEST INFRASTRUCTURE -- runs only in the build container, where the reference checkout is
mounted read-only at /root/reference.  The reference never travels to the GPU box; only the
small .npz files this script writes do (inputs + expected outputs, no reference source).

How the reference is imported (no file of the reference is modified):
  * scikit-image is not installed -> empty in-memory ``skimage``/``skimage.io`` modules are put
    into ``sys.modules`` (only ``CARLE.save_frame`` uses them, carle/env.py:511).
  * torch >= 1.7 rejects the in-place kernel write at carle/env.py:113 outside ``no_grad`` ->
    ``CARLE.set_neighborhood`` is wrapped with ``torch.no_grad()`` after import.

Run (from any cwd that is NOT /root/repo, so ``import carle`` resolves to the reference):

    cd /tmp && MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference \
        python /root/repo/tools/make_golden.py --out /root/repo/tests/golden

Fixture files
  ca_cases.npz     CA-step cases: initial grids, per-step actions, expected grids (bit-packed)
  ca_predicates.npz  CA-step cases whose action values cancel / average to exactly 1.0 (env.py:191,208)
  ca_known.json    RNG-free known answers (populations of glider / R-pentomino / puffer ...)
  nets_64.npz      RND2D + AE2D(RND2D) freshly constructed under torch.manual_seed(0) at 64x64:
                   state_dicts, 16 input grids, get_bonus_only outputs (eval mode), stacked
                   wrapper rewards over a few steps, and a predictor-update trajectory
  nets_256.npz     the shipped evaluation/*.pt weights (256x256): state_dicts (as arrays),
                   4 input grids, get_bonus_only outputs (eval mode)
  nets_256_train.npz  predictor-update trajectory at the default 256x256 geometry (3 Adam updates)
  agent_net.npz    RandomNetworkAgent (carle/agents.py:45-102): weights, observations, the network's sigmoid
                   outputs and the thresholded actions (default-init weights, and the same weights with the
                   linear layer scaled up so that toggles actually occur)
  wrappers.npz     CornerBonus / SpeedDetector / PufferDetector / ParsimonyBonus and the eval.py
                   wrapper stack: inputs and per-step rewards
  morpho.npz       MorphoBonus (carle/mcl.py:107-195) on the pattern files this package ships (the reference's own defaults
                   are missing from its tree): pattern tensors, grids after the seeded reset, actions, rewards per step
  prediction.npz   PredictionBonus / SurpriseBonus (carle/mcl.py:558-728): training trajectories (rewards per step,
                   initial and final state_dicts) incl. the reference's own glider scenario at 256x256, and explicit
                   get_bonus / get_bonus_update calls
"""
import argparse
import io
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"


def import_reference():
    if os.path.realpath(os.getcwd()).startswith("/root/repo"):
        raise SystemExit("run from a cwd outside /root/repo (see module docstring)")
    if REF not in sys.path:
        sys.path.insert(0, REF)
    sk = types.ModuleType("skimage")
    skio = types.ModuleType("skimage.io")
    sk.io = skio
    sys.modules.setdefault("skimage", sk)
    sys.modules.setdefault("skimage.io", skio)
    import torch
    import carle.env as ref_env
    assert ref_env.__file__.startswith(REF), ref_env.__file__
    ref_env.CARLE.set_neighborhood = torch.no_grad()(ref_env.CARLE.set_neighborhood)
    import carle.mcl as ref_mcl
    return torch, ref_env, ref_mcl


def pack(a):
    """float/bool array of 0/1 -> (uint8 packed bits along the flattened array, shape)"""
    a = np.asarray(a)
    return np.packbits((a != 0).reshape(-1), bitorder="little"), np.array(a.shape, dtype=np.int64)


RULES = {
    "life": ([3], [2, 3]),
    "highlife": ([3, 6], [2, 3]),
    "morley": ([3, 6, 8], [2, 4, 5]),
    "daynight": ([3, 6, 7, 8], [3, 4, 6, 7, 8]),
    "lfod": ([3], [0, 2, 3]),           # "live free or die" as written in train_mcl.py:98
    "b2s0": ([2], [0]),                 # single-element lists: bool grids, eval.py:94
    "b0357s2468": ([0, 3, 5, 7], [2, 4, 6, 8]),   # B0 rule (tests/test_env.py:24-25)
    "b012345678s012345678": (list(range(9)), list(range(9))),
}


def run_case(torch, ref_env, name, rule, n, size, asize, steps, init_p, action_kind, record,
             seed, extra=None):
    """Drive the genuine reference and record everything needed to replay the case."""
    rng = np.random.default_rng(seed)
    env = ref_env.CARLE(width=size, height=size, action_width=asize, action_height=asize,
                        instances=n, device="cpu")
    env.birth, env.survive = list(RULES[rule][0]), list(RULES[rule][1])
    env.reset()
    init = (rng.random((n, 1, size, size)) < init_p).astype(np.float32)
    env.universe = torch.from_numpy(init.copy())     # what load_universe does, carle/env.py:406
    acts, snaps, stepno, ssa = [], [], [], []
    aw = env.action_width
    for t in range(steps):
        if action_kind == "zero":
            a = np.zeros((n, 1, aw, aw), np.float32)
        elif action_kind == "p10":
            a = (rng.random((n, 1, aw, aw)) <= 0.1).astype(np.float32)
        elif action_kind == "p50_first":
            a = (rng.random((n, 1, aw, aw)) < 0.5).astype(np.float32) if t == 0 else \
                np.zeros((n, 1, aw, aw), np.float32)
        elif action_kind == "oversize":      # full-grid action -> centre crop, env.py:164-167
            a = (rng.random((n, 1, size, size)) <= 0.1).astype(np.float32)
        elif action_kind == "values":        # any non-zero value toggles, env.py:182
            vals = np.array([0.0, 0.0, 0.0, 0.5, -1.0, 2.0, 1.0, -0.0, 1e-30, np.inf],
                            np.float32)
            a = vals[rng.integers(0, len(vals), (n, 1, aw, aw))]
        elif action_kind == "reset_mid":     # all-ones action at t==3: master reset, env.py:208
            if t == 3:
                a = np.ones((n, 1, aw, aw), np.float32)
            elif t == 5:                      # all ones for one instance only: no reset
                a = (rng.random((n, 1, aw, aw)) <= 0.1).astype(np.float32)
                a[0] = 1.0
            else:
                a = (rng.random((n, 1, aw, aw)) <= 0.1).astype(np.float32)
        elif action_kind == "cancel":        # values cancel: sum == 0 although cells toggle (env.py:191)
            if t % 2 == 0:
                a = np.zeros((n, 1, aw, aw), np.float32)
                a[:, 0, 5, 5] = 1.0
                a[:, 0, 9, 9] = -1.0
                a[0, 0, 20, 3:5] = (2.0, -2.0)
            elif t == 3:
                a = np.zeros((n, 1, aw, aw), np.float32)
            else:
                a = (rng.random((n, 1, aw, aw)) <= 0.1).astype(np.float32)
        elif action_kind == "mean1":         # mean == 1.0 without being all ones (env.py:208): RESET
            if t == 2:
                a = np.ones((n, 1, aw, aw), np.float32)
                a[0, 0, 3, 3] = 0.0
                a[0, 0, 4, 4] = 2.0
            elif t == 4:                      # all ones but one element 0.5: mean != 1, no reset
                a = np.ones((n, 1, aw, aw), np.float32)
                a[n - 1, 0, 7, 7] = 0.5
            else:
                a = (rng.random((n, 1, aw, aw)) <= 0.1).astype(np.float32)
        else:
            raise ValueError(action_kind)
        acts.append(a)
        obs, reward, done, info = env.step(torch.from_numpy(a.copy()))
        assert obs is env.universe
        assert tuple(reward.shape) == (n, 1) and tuple(done.shape) == (n, 1)
        assert float(reward.abs().sum()) == 0.0 and float(done.abs().sum()) == 0.0
        assert len(info) == n
        o = obs.detach().numpy()
        assert set(np.unique(o)).issubset({0.0, 1.0})
        stepno.append(env.step_number)
        ssa.append(env.steps_since_action)
        if (t + 1) in record:
            snaps.append(o.copy())
    acts = np.stack(acts)
    binary = bool(np.all((acts == 0) | (acts == 1)))
    out = {}
    meta = dict(name=name, rule=rule, birth=RULES[rule][0], survive=RULES[rule][1], n=n,
                size=size, action_size=int(aw), steps=steps, record=sorted(record),
                action_kind=action_kind, actions_binary=binary, seed=seed,
                action_shape=list(acts.shape[1:]))
    if extra:
        meta.update(extra)
    out[f"{name}/meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    out[f"{name}/init"], _ = pack(init)
    if binary:
        out[f"{name}/actions"], _ = pack(acts)
    else:
        out[f"{name}/actions"] = acts
    out[f"{name}/obs"], _ = pack(np.stack(snaps))
    out[f"{name}/step_number"] = np.array(stepno, np.int64)
    out[f"{name}/steps_since_action"] = np.array(ssa, np.int64)
    return out


def make_ca(torch, ref_env, outdir):
    cases = {}
    seed = 1000
    # (1) every rule on 64x64, random soup then zero actions; long horizon for Life
    for rule in RULES:
        seed += 1
        steps = 256 if rule == "life" else 32
        rec = {1, 2, 3, 16, 32} | ({64, 256} if rule == "life" else set())
        cases.update(run_case(torch, ref_env, f"soup64_{rule}", rule, 4, 64, 64, steps, 0.5,
                              "zero", rec, seed))
    # (2) random toggles every step (RandomAgent-like, agents.py:38-40)
    for rule, n, size, asize, steps in [("life", 8, 64, 64, 32), ("highlife", 3, 64, 64, 16),
                                        ("morley", 2, 128, 64, 16), ("life", 2, 256, 64, 16),
                                        ("daynight", 2, 256, 64, 8), ("highlife", 1, 512, 64, 8),
                                        ("life", 1, 512, 256, 6), ("life", 2, 64, 32, 16),
                                        ("lfod", 2, 128, 128, 8)]:
        seed += 1
        cases.update(run_case(torch, ref_env, f"tog{size}a{asize}_{rule}", rule, n, size, asize,
                              steps, 0.3, "p10", {1, 2, 3, steps // 2, steps}, seed))
    # (3) sizes that are not multiples of 64 (legal in the reference: square, even)
    for rule, n, size, asize in [("life", 3, 32, 32), ("highlife", 2, 96, 64),
                                 ("life", 2, 100, 64), ("morley", 2, 48, 16), ("life", 1, 8, 8),
                                 ("life", 2, 192, 64), ("b0357s2468", 1, 40, 40)]:
        seed += 1
        cases.update(run_case(torch, ref_env, f"odd{size}a{asize}_{rule}", rule, n, size, asize,
                              12, 0.4, "p10", {1, 2, 5, 12}, seed))
    # (4) oversize action (crop path), arbitrary values, master reset
    seed += 1
    cases.update(run_case(torch, ref_env, "crop256_life", "life", 2, 256, 64, 6, 0.3, "oversize",
                          {1, 3, 6}, seed))
    seed += 1
    cases.update(run_case(torch, ref_env, "crop128_highlife", "highlife", 2, 128, 64, 6, 0.3,
                          "oversize", {1, 3, 6}, seed))
    seed += 1
    cases.update(run_case(torch, ref_env, "values64_life", "life", 3, 64, 64, 6, 0.3, "values",
                          {1, 2, 6}, seed))
    seed += 1
    cases.update(run_case(torch, ref_env, "values256_morley", "morley", 2, 256, 64, 4, 0.3,
                          "values", {1, 2, 4}, seed))
    seed += 1
    cases.update(run_case(torch, ref_env, "reset64_life", "life", 3, 64, 64, 8, 0.5, "reset_mid",
                          set(range(1, 9)), seed))
    seed += 1
    cases.update(run_case(torch, ref_env, "reset256_life", "life", 2, 256, 64, 8, 0.5,
                          "reset_mid", set(range(1, 9)), seed))
    # (5) the reference's own CPU-runnable config (BASELINE.json configs[0]): N=1, 64x64, 256 steps
    seed += 1
    cases.update(run_case(torch, ref_env, "cfg1_life", "life", 1, 64, 64, 256, 0.5, "zero",
                          {1, 64, 128, 256}, seed))
    np.savez_compressed(os.path.join(outdir, "ca_cases.npz"), **cases)
    print("ca_cases.npz:", len([k for k in cases if k.endswith("/meta")]), "cases")


def make_predicates(torch, ref_env, outdir):
    """The two whole-batch action predicates of CARLE.step on actions outside {0,1}: values that
    cancel (torch.sum(action) == 0 although cells toggle, env.py:191) and a mean of exactly 1.0
    without every element being 1 (env.py:208 -> master reset).  Separate file: the default build
    evaluates 'any element non-zero' / 'every element == 1' (identical on {0,1} actions) and only
    CARLE(exact_action_predicates=True) follows the reference on these inputs."""
    cases = {}
    cases.update(run_case(torch, ref_env, "cancel64_life", "life", 2, 64, 64, 6, 0.3, "cancel",
                          set(range(1, 7)), 2001))
    cases.update(run_case(torch, ref_env, "cancel256_highlife", "highlife", 2, 256, 64, 6, 0.3, "cancel",
                          set(range(1, 7)), 2002))
    cases.update(run_case(torch, ref_env, "mean1_64_life", "life", 2, 64, 64, 6, 0.4, "mean1",
                          set(range(1, 7)), 2003))
    cases.update(run_case(torch, ref_env, "mean1_128_life", "life", 3, 128, 64, 6, 0.4, "mean1",
                          set(range(1, 7)), 2004))
    np.savez_compressed(os.path.join(outdir, "ca_predicates.npz"), **cases)
    print("ca_predicates.npz:", len([k for k in cases if k.endswith("/meta")]), "cases")


def make_known(torch, ref_env, ref_mcl, outdir):
    """RNG-free known answers + behavioural quirks of the reference, as plain JSON."""
    known = {}

    def pops(action, rule, at, size=64):
        env = ref_env.CARLE(width=size, height=size, instances=1, device="cpu")
        env.birth, env.survive = list(RULES[rule][0]), list(RULES[rule][1])
        env.reset()
        res = {}
        for t in range(1, max(at) + 1):
            a = action if t == 1 else torch.zeros_like(action)
            obs = env.step(a)[0]
            if t in at:
                res[t] = int(obs.sum().item())
        return res, env

    at = [1, 2, 3, 4, 11, 51, 101, 256]
    g, env = pops(ref_mcl.get_glider(), "life", at)
    known["glider_life_pop"] = g
    known["glider_life_step_number"] = env.step_number
    rp = torch.zeros(1, 1, 64, 64)
    rp[0, 0, 31, 32:34] = 1.0
    rp[0, 0, 32, 31:33] = 1.0
    rp[0, 0, 33, 32] = 1.0
    known["rpent_cells"] = [[31, 32], [31, 33], [32, 31], [32, 32], [33, 32]]
    known["rpent_life_pop"], _ = pops(rp, "life", at)
    known["rpent_highlife_pop"], _ = pops(rp, "highlife", at)
    known["puffer_morley_pop"], _ = pops(ref_mcl.get_morley_puffer(), "morley", at)
    known["glider_cells"] = np.argwhere(ref_mcl.get_glider()[0, 0].numpy() != 0).tolist()
    known["puffer_cells"] = np.argwhere(ref_mcl.get_morley_puffer()[0, 0].numpy() != 0).tolist()
    # torus seam
    env = ref_env.CARLE(width=64, height=64, instances=1, device="cpu")
    env.reset()
    a = torch.zeros(1, 1, 64, 64)
    for (r, c) in [(0, 63), (0, 0), (0, 1)]:
        a[0, 0, r, c] = 1.0
    obs = env.step(a)[0]
    known["seam_in"] = [[0, 63], [0, 0], [0, 1]]
    known["seam_out"] = np.argwhere(obs[0, 0].numpy() != 0).tolist()
    # corner blinker across both seams
    env.reset()
    a = torch.zeros(1, 1, 64, 64)
    for (r, c) in [(63, 63), (0, 63), (1, 63)]:
        a[0, 0, r, c] = 1.0
    obs = env.step(a)[0]
    known["seam2_in"] = [[63, 63], [0, 63], [1, 63]]
    known["seam2_out"] = np.argwhere(obs[0, 0].numpy() != 0).tolist()
    # master reset
    env = ref_env.CARLE(width=64, height=64, instances=2, device="cpu")
    env.reset()
    env.step((torch.rand(2, 1, 64, 64) < 0.3).float())
    env.step(torch.zeros(2, 1, 64, 64))
    sn_before = env.step_number
    obs = env.step(torch.ones(2, 1, 64, 64))[0]
    known["master_reset"] = dict(step_number_before=sn_before, step_number_after=env.step_number,
                                 obs_sum=float(obs.sum()),
                                 steps_since_action_after=env.steps_since_action)
    # rule parsing (tests/test_env.py:17-39)
    env = ref_env.CARLE()
    env.birth_rule_from_string("asdfasdfB0357*!@#!@$%")
    env.survive_rule_from_string("S2468")
    known["parse_b"] = env.birth
    known["parse_s"] = env.survive
    env.rules_from_string("B368/S245")
    known["parse_rules_b368s245"] = [env.birth, env.survive]
    env.rules_from_string("b3/s23")
    known["parse_rules_lower"] = [env.birth, env.survive]
    env.rules_from_string("B9a33/S")
    known["parse_rules_dup_empty"] = [env.birth, env.survive]
    # attribute surface
    env = ref_env.CARLE()
    known["defaults"] = dict(width=env.width, height=env.height, action_width=env.action_width,
                             action_height=env.action_height, instances=env.instances,
                             device=str(env.my_device), birth=env.birth, survive=env.survive,
                             state_dict_keys=list(env.state_dict().keys()),
                             weight=env.state_dict()["neighborhood.weight"].reshape(-1).tolist(),
                             weight_requires_grad=bool(env.neighborhood.weight.requires_grad))
    env.reset()
    pad = env.action_padding(torch.ones(1, 1, 64, 64))
    known["default_action_window"] = [int(np.argwhere(pad[0, 0].numpy())[:, 0].min()),
                                      int(np.argwhere(pad[0, 0].numpy())[:, 0].max()),
                                      int(np.argwhere(pad[0, 0].numpy())[:, 1].min()),
                                      int(np.argwhere(pad[0, 0].numpy())[:, 1].max())]
    # empty rule list raises TypeError (reduce of empty sequence, env.py:221-224)
    env = ref_env.CARLE(width=64, height=64)
    env.reset()
    env.birth = []
    try:
        env.step(torch.zeros(1, 1, 64, 64))
        known["empty_birth_raises"] = None
    except Exception as e:   # noqa: BLE001
        known["empty_birth_raises"] = type(e).__name__
    # action given as 2-D / 3-D / list (env.py:152-156) -- N=1
    env = ref_env.CARLE(width=64, height=64)
    env.reset()
    a2 = np.zeros((64, 64), np.float32)
    a2[10, 10:13] = 1
    # NB: step() calls torch.sum(action) first, so a raw ndarray/list works only via Tensor
    o2 = env.step(torch.from_numpy(a2))[0]
    known["action_2d_out"] = np.argwhere(o2[0, 0].numpy() != 0).tolist()
    with open(os.path.join(outdir, "ca_known.json"), "w") as f:
        json.dump(known, f, indent=1, sort_keys=True)
    print("ca_known.json written")


def sd_to_npz(prefix, sd, out):
    keys = list(sd.keys())
    out[f"{prefix}/keys"] = np.frombuffer(json.dumps(keys).encode(), dtype=np.uint8)
    for i, k in enumerate(keys):
        out[f"{prefix}/t{i}"] = sd[k].detach().cpu().numpy().copy()      # (a view would follow later in-place updates)


def make_nets(torch, ref_env, ref_mcl, outdir):
    import contextlib
    sink = io.StringIO()
    # ---- 64x64, fresh seeded construction (predictor before random net, mcl.py:248-249)
    out = {}
    n = 16
    torch.manual_seed(0)
    with contextlib.redirect_stdout(sink):
        env = ref_env.CARLE(width=64, height=64, instances=n, device="cpu")
        rnd = ref_mcl.RND2D(env)
        ae = ref_mcl.AE2D(rnd)
    rnd.eval()
    ae.eval()
    rng = np.random.default_rng(77)
    dens = np.linspace(0.02, 0.6, n).reshape(n, 1, 1, 1)
    grids = (rng.random((n, 1, 64, 64)) < dens).astype(np.float32)
    grids[0] = 0.0                      # all-dead grid
    grids[1] = 1.0                      # all-alive grid
    g = torch.from_numpy(grids)
    ae.reset()
    sd_to_npz("rnd_sd", rnd.state_dict(), out)
    sd_to_npz("ae_sd", ae.state_dict(), out)
    out["grids"], _ = pack(grids)
    out["rnd_bonus"] = rnd.get_bonus_only(g).numpy()
    out["ae_bonus"] = ae.get_bonus_only(g).numpy()
    out["rnd_target"] = rnd.random_forward(g).detach().numpy()
    out["rnd_pred"] = rnd.forward(g).detach().numpy()
    out["ae_pred"] = ae.forward(g).detach().numpy().astype(np.float32)
    # stacked wrappers, eval mode, rewards over 4 steps, no predictor update (batch_size large)
    rnd.batch_size = 10 ** 6
    ae.batch_size = 10 ** 6
    ae.reset()
    env.universe = g.clone()
    acts, rews = [], []
    for t in range(4):
        a = (rng.random((n, 1, 64, 64)) <= 0.1).astype(np.float32)
        obs, reward, done, info = ae.step(torch.from_numpy(a.copy()))
        acts.append(a)
        rews.append(reward.detach().numpy().copy())
    out["stack_actions"], _ = pack(np.stack(acts))
    out["stack_rewards"] = np.stack(rews)
    out["stack_final_obs"], _ = pack(obs.detach().numpy())
    # predictor-update trajectory ("next" row): eval mode (no dropout),
    # batch_size=2, 6 steps -> 3 Adam updates for each wrapper
    torch.manual_seed(0)
    with contextlib.redirect_stdout(sink):
        env = ref_env.CARLE(width=64, height=64, instances=4, device="cpu")
        rnd = ref_mcl.RND2D(env)
        ae = ref_mcl.AE2D(rnd)
    rnd.eval()
    ae.eval()
    rnd.batch_size = 2
    ae.batch_size = 2
    ae.reset()
    env.universe = g[4:8].clone()
    acts, rews = [], []
    for t in range(6):
        a = (rng.random((4, 1, 64, 64)) <= 0.1).astype(np.float32)
        obs, reward, done, info = ae.step(torch.from_numpy(a.copy()))
        acts.append(a)
        rews.append(reward.detach().numpy().copy())
    out["upd_init"], _ = pack(grids[4:8])
    out["upd_actions"], _ = pack(np.stack(acts))
    out["upd_rewards"] = np.stack(rews)
    out["upd_updates"] = np.array([rnd.updates, ae.updates], np.int64)
    sd_to_npz("upd_rnd_sd", rnd.state_dict(), out)
    sd_to_npz("upd_ae_sd", ae.state_dict(), out)
    np.savez_compressed(os.path.join(outdir, "nets_64.npz"), **out)
    print("nets_64.npz written")

    # ---- 256x256, the shipped evaluation weights
    out = {}
    with contextlib.redirect_stdout(sink):
        env = ref_env.CARLE(instances=4, device="cpu")
        rnd = ref_mcl.RND2D(env)
        rnd.load_state_dict(torch.load(os.path.join(REF, "evaluation/RND2D_mcl.pt"), map_location="cpu"))
        ae = ref_mcl.AE2D(rnd)
        ae.load_state_dict(torch.load(os.path.join(REF, "evaluation/AE2D_mcl.pt"), map_location="cpu"))
    rnd.eval()
    ae.eval()
    ae.reset()
    dens = np.array([0.0, 0.05, 0.2, 0.5]).reshape(4, 1, 1, 1)
    grids = (rng.random((4, 1, 256, 256)) < dens).astype(np.float32)
    g = torch.from_numpy(grids)
    sd_to_npz("rnd_sd", rnd.state_dict(), out)
    sd_to_npz("ae_sd", ae.state_dict(), out)
    out["grids"], _ = pack(grids)
    out["rnd_bonus"] = rnd.get_bonus_only(g).numpy()
    out["ae_bonus"] = ae.get_bonus_only(g).numpy()
    np.savez_compressed(os.path.join(outdir, "nets_256.npz"), **out)
    print("nets_256.npz written")


def make_train256(torch, ref_env, ref_mcl, outdir):
    """Predictor-update trajectory at the reference's DEFAULT geometry (256x256): AE2D(RND2D(CARLE)),
    seeded construction, eval mode (no dropout), batch_size=2, 6 steps -> 3 Adam updates per wrapper
    (carle/mcl.py:333-341, 368-386, 498-547)."""
    import contextlib
    sink = io.StringIO()
    out = {}
    rng = np.random.default_rng(256)
    n = 2
    torch.manual_seed(0)
    with contextlib.redirect_stdout(sink):
        env = ref_env.CARLE(instances=n, device="cpu")          # 256x256, 64x64 action window
        rnd = ref_mcl.RND2D(env)
        ae = ref_mcl.AE2D(rnd)
    rnd.eval()
    ae.eval()
    rnd.batch_size = 2
    ae.batch_size = 2
    ae.reset()
    sd_to_npz("init_rnd_sd", rnd.state_dict(), out)
    sd_to_npz("init_ae_sd", ae.state_dict(), out)
    init = (rng.random((n, 1, 256, 256)) < 0.3).astype(np.float32)
    env.universe = torch.from_numpy(init.copy())
    acts, rews = [], []
    for t in range(6):
        a = (rng.random((n, 1, 64, 64)) <= 0.1).astype(np.float32)
        obs, reward, done, info = ae.step(torch.from_numpy(a.copy()))
        acts.append(a)
        rews.append(reward.detach().numpy().copy())
    out["init"], _ = pack(init)
    out["actions"], _ = pack(np.stack(acts))
    out["rewards"] = np.stack(rews)
    out["updates"] = np.array([rnd.updates, ae.updates], np.int64)
    out["final_obs"], _ = pack(obs.detach().numpy())
    sd_to_npz("final_rnd_sd", rnd.state_dict(), out)
    sd_to_npz("final_ae_sd", ae.state_dict(), out)
    np.savez_compressed(os.path.join(outdir, "nets_256_train.npz"), **out)
    print("nets_256_train.npz written")


def make_wrappers(torch, ref_env, ref_mcl, outdir):
    """Hand-coded reward wrappers: CornerBonus, SpeedDetector, PufferDetector,
    ParsimonyBonus driven by the genuine reference; inputs + rewards per step."""
    import contextlib
    sink = io.StringIO()
    out = {}
    rng = np.random.default_rng(99)

    def run(name, build, n, size, steps, act_fn, init_p=0.2, post=None):
        torch.manual_seed(0)
        with contextlib.redirect_stdout(sink):
            env = ref_env.CARLE(width=size, height=size, instances=n, device="cpu")
            w = build(env)
        if post:
            post(w)
        w.eval()
        w.reset()
        init = (rng.random((n, 1, size, size)) < init_p).astype(np.float32)
        env.universe = torch.from_numpy(init.copy())
        acts, rews = [], []
        for t in range(steps):
            a = act_fn(t, n)
            obs, reward, done, info = w.step(torch.from_numpy(a.copy()))
            acts.append(a)
            rews.append(np.asarray(reward.detach().numpy(), dtype=np.float32).reshape(n, -1).copy())
        out[f"{name}/init"], _ = pack(init)
        out[f"{name}/actions"], _ = pack(np.stack(acts))
        out[f"{name}/rewards"] = np.stack(rews)
        out[f"{name}/final_obs"], _ = pack(obs.detach().numpy())
        out[f"{name}/meta"] = np.frombuffer(json.dumps(dict(n=n, size=size, steps=steps)).encode(), dtype=np.uint8)
        return w

    p10 = lambda t, n: (rng.random((n, 1, 64, 64)) <= 0.1).astype(np.float32)
    sparse = lambda t, n: (p10(t, n) if t < 2 else np.zeros((n, 1, 64, 64), np.float32))
    run("corner", lambda e: ref_mcl.CornerBonus(e), 2, 256, 6, p10)
    run("speed", lambda e: ref_mcl.SpeedDetector(e), 2, 256, 8, sparse)

    def small_threshold(w):
        w.growth_threshold = 6
    run("puffer", lambda e: ref_mcl.PufferDetector(e), 1, 256, 20, sparse, init_p=0.3, post=small_threshold)
    # eval.py stacking: RND2D -> AE2D -> SpeedDetector -> PufferDetector (eval.py:83-87), N=1, 64x64
    def stack(e):
        w = ref_mcl.RND2D(e)
        w = ref_mcl.AE2D(w)
        w.batch_size = 10 ** 6
        w.env.batch_size = 10 ** 6
        w = ref_mcl.SpeedDetector(w)
        w.reward_scale = 1e-2
        w = ref_mcl.PufferDetector(w)
        w.reward_scale = 1e-3
        return w
    w = run("evalstack", stack, 1, 64, 6, sparse)
    sd_to_npz("evalstack_sd", w.state_dict(), out)
    # ParsimonyBonus over RND2D, N=1 (the reference divides (N,1) by (N,): only N=1 is meaningful)
    def pars(e):
        w = ref_mcl.RND2D(e)
        w.batch_size = 10 ** 6
        return ref_mcl.ParsimonyBonus(w)
    dense = lambda t, n: (rng.random((n, 1, 64, 64)) <= (0.5 if t % 2 else 0.01)).astype(np.float32)
    w = run("parsimony", pars, 1, 64, 4, dense)
    sd_to_npz("parsimony_sd", w.state_dict(), out)
    np.savez_compressed(os.path.join(outdir, "wrappers.npz"), **out)
    print("wrappers.npz written")


def make_prediction(torch, ref_env, ref_mcl, outdir):
    """PredictionBonus / SurpriseBonus (carle/mcl.py:558-728) driven by the genuine reference in eval mode (no dropout):
    per case the initial state_dict and grids, the actions, the reward of every step, the final state_dict.  batch_size 2,
    so the frame buffer fills (prediction_steps = 5) while Adam updates happen; one instance stays empty (the
    zero-for-empty-grids branch, mcl.py:687-691); 64x64 and the default 256x256 (the reference's own test scenario:
    tests/test_mcl.py:17-53, a glider then idle steps)."""
    import contextlib
    sink = io.StringIO()
    out = {}
    rng = np.random.default_rng(558)

    def run(name, cls, n, size, steps, init_fn, act_fn, batch_size=2, scale=1.0):
        torch.manual_seed(7)
        with contextlib.redirect_stdout(sink):
            env = ref_env.CARLE(width=size, height=size, instances=n, device="cpu")
            w = cls(env)
        w.eval()
        w.batch_size = batch_size
        w.reward_scale = scale
        w.reset()
        sd_to_npz(f"{name}/init_sd", w.state_dict(), out)
        init = init_fn(n, size)
        env.universe = torch.from_numpy(init.copy())
        acts, rews, live = [], [], []
        for t in range(steps):
            a = act_fn(t, n)
            obs, reward, done, info = w.step(torch.from_numpy(a.copy()))
            acts.append(a)
            rews.append(np.asarray(reward.detach().numpy(), dtype=np.float32).reshape(n, -1).copy())
            live.append(float(w.live_cells))
        out[f"{name}/init"], _ = pack(init)
        out[f"{name}/actions"], _ = pack(np.stack(acts))
        out[f"{name}/rewards"] = np.stack(rews)
        out[f"{name}/live_cells"] = np.array(live, np.float64)
        out[f"{name}/final_obs"], _ = pack(obs.detach().numpy())
        sd_to_npz(f"{name}/final_sd", w.state_dict(), out)
        out[f"{name}/meta"] = np.frombuffer(json.dumps(dict(n=n, size=size, steps=steps, batch_size=batch_size,
                                                            scale=scale, cls=cls.__name__)).encode(), dtype=np.uint8)
        return w

    def init_one_empty(n, size):
        g = (rng.random((n, 1, size, size)) < 0.25).astype(np.float32)
        g[-1] = 0.0                                           # the last instance stays empty (zero actions below)
        return g

    def sparse_keep_last_empty(t, n):
        a = (rng.random((n, 1, 64, 64)) <= (0.05 if t % 3 == 0 else 0.0)).astype(np.float32)
        a[-1] = 0.0
        return a

    run("pred64", ref_mcl.PredictionBonus, 3, 64, 10, init_one_empty, sparse_keep_last_empty)
    run("pred64_scaled", ref_mcl.PredictionBonus, 2, 64, 7, init_one_empty, sparse_keep_last_empty, batch_size=3, scale=0.5)
    run("surprise64", ref_mcl.SurpriseBonus, 3, 64, 6, init_one_empty, sparse_keep_last_empty)
    glider = ref_mcl.get_glider().numpy().astype(np.float32)
    run("pred256_glider", ref_mcl.PredictionBonus, 1, 256, 8, lambda n, size: np.zeros((n, 1, size, size), np.float32),
        lambda t, n: glider.copy() if t == 0 else np.zeros_like(glider))
    # get_bonus (no gradient), get_bonus_update (one Adam step per call) on explicit frames
    torch.manual_seed(11)
    with contextlib.redirect_stdout(sink):
        env = ref_env.CARLE(width=64, height=64, instances=2, device="cpu")
        w = ref_mcl.PredictionBonus(env)
    w.eval()
    w.reset()
    sd_to_npz("calls/init_sd", w.state_dict(), out)
    frames = (rng.random((7, 2, 1, 64, 64)) < 0.3).astype(np.float32)
    res = []
    for t in range(7):
        f = torch.from_numpy(frames[t].copy())
        r = w.get_bonus(f) if t < 3 else w.get_bonus_update(f)
        res.append(r.detach().numpy().copy())
    out["calls/frames"], _ = pack(frames.reshape(14, 1, 64, 64))
    out["calls/bonus"] = np.stack(res)
    sd_to_npz("calls/final_sd", w.state_dict(), out)
    np.savez_compressed(os.path.join(outdir, "prediction.npz"), **out)
    print("prediction.npz written")


def make_morpho(torch, ref_env, ref_mcl, outdir):
    """MorphoBonus (carle/mcl.py:107-195).  Its constructor reads glider_1.rle / glider_2.rle beside mcl.py, which the
    reference tree does not contain: the class is driven through a two-line subclass whose ``add_default_patterns`` points
    at the two pattern files THIS package ships under those names (carle_amd/glider_*.rle) and at one more pattern written
    here; everything else -- RLE parsing, padding, cropping, normalisation, the six orientations, reset's random seed
    cells, the correlation and the reward -- is the reference's code.  Recorded: the pattern tensor, the grids after
    reset (seeded CPU generator), actions, rewards per step."""
    import contextlib
    import tempfile
    sink = io.StringIO()
    out = {}
    rng = np.random.default_rng(107)
    pkg = "/root/repo/carle_amd"
    extra_rle = "x = 5, y = 4, rule = B3/S23\nb3o$o3bo$2bo$o2bo!\n"
    # 14 live cells: 15 / 14 and (1 / 14) * 15 -- what torch's ``15. / tensor`` computes -- differ by one ulp in fp32
    # (a difference tests/tools/fuzz_oracle_vs_reference.py found in the first restatement of the normalisation)
    cells14_rle = "x = 4, y = 5, rule = B3/S23\n3o$b3o$2o$2obo$ob2o!\n"
    with tempfile.TemporaryDirectory() as tmp:
        extra = os.path.join(tmp, "extra.rle")
        with open(extra, "w") as f:
            f.write(extra_rle)

        class Morpho(ref_mcl.MorphoBonus):
            def add_default_patterns(self):
                self.add_rle_pattern(os.path.join(pkg, "glider_1.rle"))
                self.add_rle_pattern(os.path.join(pkg, "glider_2.rle"))

        extra14 = os.path.join(tmp, "cells14.rle")
        with open(extra14, "w") as f:
            f.write(cells14_rle)
        for name, n, steps, with_extra in (("default", 3, 6, False), ("extra", 2, 4, True), ("cells14", 1, 3, extra14)):
            torch.manual_seed(107)
            with contextlib.redirect_stdout(sink):
                env = ref_env.CARLE(width=64, height=64, instances=n, device="cpu")
                w = Morpho(env)
                if with_extra:
                    w.add_rle_pattern(extra if with_extra is True else with_extra)
            w.reward_scale = 0.5 if with_extra else 1.0
            obs = w.reset()
            out[f"{name}/patterns"] = w.target_patterns.detach().numpy().copy()
            out[f"{name}/reset_obs"], _ = pack(obs.detach().numpy())
            acts, rews = [], []
            for t in range(steps):
                a = (rng.random((n, 1, 64, 64)) <= (0.02 if t % 2 == 0 else 0.0)).astype(np.float32)
                if t == 0:
                    a[0, 0, 10:13, 20:23] = np.array([[0, 1, 0], [0, 0, 1], [1, 1, 1]], np.float32)   # a glider: a full match
                obs, reward, done, info = w.step(torch.from_numpy(a.copy()))
                acts.append(a)
                rews.append(np.asarray(reward.detach().numpy(), dtype=np.float32).reshape(n, -1).copy())
            out[f"{name}/actions"], _ = pack(np.stack(acts))
            out[f"{name}/rewards"] = np.stack(rews)
            out[f"{name}/final_obs"], _ = pack(obs.detach().numpy())
            out[f"{name}/meta"] = np.frombuffer(json.dumps(dict(n=n, steps=steps, scale=float(w.reward_scale),
                                                                birth=list(env.birth), survive=list(env.survive),
                                                                extra_rle=(extra_rle if with_extra is True else cells14_rle) if with_extra else "")).encode(),
                                                         dtype=np.uint8)
    np.savez_compressed(os.path.join(outdir, "morpho.npz"), **out)
    print("morpho.npz written")


def make_agent_net(torch, ref_env, outdir):
    """RandomNetworkAgent of the reference.  Its constructor ignores ``observation_width`` (the key it
    tests for is misspelt, agents.py:56-58) and always builds for width 256; ``observation_height``
    works.  Geometry here: observations (n,1,16,256) -> dense 256, action 8x8 -> a 64 KB linear layer."""
    import carle.agents as ref_agents
    assert ref_agents.__file__.startswith(REF)
    out = {}
    torch.manual_seed(5)
    agent = ref_agents.RandomNetworkAgent(observation_height=16, action_width=8, action_height=8)
    assert agent.observation_width == 256 and agent.network[7].weight.shape == (64, 256)
    rng = np.random.default_rng(9)
    n = 6
    dens = np.array([0.0, 1.0, 0.1, 0.3, 0.5, 0.8]).reshape(n, 1, 1, 1)
    obs = (rng.random((n, 1, 16, 256)) < dens).astype(np.float32)
    out["obs"], out["obs_shape"] = pack(obs)
    sd_to_npz("sd", agent.state_dict(), out)
    g = torch.from_numpy(obs)
    out["output_default"] = agent.network(g).numpy()
    out["action_default"] = agent(g).numpy().astype(np.float32)
    # the freshly initialised agent practically never toggles (sigmoid(x) <= 0.1 needs x <= -2.2): scale the
    # linear layer so that both sides of the threshold occur
    with torch.no_grad():
        agent.network[7].weight.mul_(40.0)
    out["lin_scaled"] = agent.network[7].weight.numpy().copy()
    out["output_scaled"] = agent.network(g).numpy()
    out["action_scaled"] = agent(g).numpy().astype(np.float32)
    out["toggle_rate"] = np.float32(agent.toggle_rate)
    frac = float(out["action_scaled"].mean())
    assert 0.02 < frac < 0.98, frac
    np.savez_compressed(os.path.join(outdir, "agent_net.npz"), **out)
    print("agent_net.npz: toggled fraction (scaled weights) %.3f" % frac)


def make_rle(torch, ref_env, outdir):
    """RLE codec: get_rle strings and rle_to_grid grids of the reference."""
    import base64
    rng = np.random.default_rng(5)
    cases = []
    for (size, p, rule, step, action) in [(64, 0.5, "life", 3, False), (64, 0.02, "morley", 0, True),
                                          (64, 0.0, "life", 7, False), (64, 1.0, "b2s0", 1, False),
                                          (256, 0.1, "daynight", 12, False), (32, 0.3, "life", 2, False)]:
        env = ref_env.CARLE(width=size, height=size, device="cpu")
        env.birth, env.survive = list(RULES[rule][0]), list(RULES[rule][1])
        env.reset()
        env.instance_id = "1700000000"
        env.step_number = step
        g = (rng.random((size, size)) < p).astype(np.float32)
        rle = env.get_rle(torch.from_numpy(g), action=action)
        back = env.rle_to_grid(rle[rle.index("\n", rle.index("rule")) + 1:]).numpy()
        cases.append(dict(size=size, birth=env.birth, survive=env.survive, step=step, action=action,
                          instance_id=env.instance_id,
                          grid=base64.b64encode(np.packbits(g != 0, bitorder="little").tobytes()).decode(),
                          rle=rle,
                          decoded=base64.b64encode(np.packbits(back != 0, bitorder="little").tobytes()).decode()))
    # standard (Golly-style) RLE bodies through rle_to_grid
    decodes = []
    for body, size in [("3b2o$3ob2o$5o$b3o!", 64), ("bo$2bo$3o!", 64), ("2o$2o3$4bo$5bo$3b3o!", 32),
                       ("10b3o\n2$o!", 64), ("64o$64b$o62bo!", 64)]:
        env = ref_env.CARLE(width=size, height=size, device="cpu")
        g = env.rle_to_grid(body).numpy()
        decodes.append(dict(size=size, rle=body,
                            grid=base64.b64encode(np.packbits(g != 0, bitorder="little").tobytes()).decode()))
    with open(os.path.join(outdir, "rle.json"), "w") as f:
        json.dump(dict(encode=cases, decode=decodes), f, indent=1)
    print("rle.json written")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="/root/repo/tests/golden")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    torch, ref_env, ref_mcl = import_reference()
    torch.set_num_threads(8)
    todo = args.only.split(",") if args.only else ["ca", "predicates", "known", "nets", "train256", "wrappers", "rle", "agent", "prediction", "morpho"]
    if "ca" in todo:
        make_ca(torch, ref_env, args.out)
    if "predicates" in todo:
        make_predicates(torch, ref_env, args.out)
    if "known" in todo:
        make_known(torch, ref_env, ref_mcl, args.out)
    if "nets" in todo:
        make_nets(torch, ref_env, ref_mcl, args.out)
    if "train256" in todo:
        make_train256(torch, ref_env, ref_mcl, args.out)
    if "wrappers" in todo:
        make_wrappers(torch, ref_env, ref_mcl, args.out)
    if "rle" in todo:
        make_rle(torch, ref_env, args.out)
    if "agent" in todo:
        make_agent_net(torch, ref_env, args.out)
    if "prediction" in todo:
        make_prediction(torch, ref_env, ref_mcl, args.out)
    if "morpho" in todo:
        make_morpho(torch, ref_env, ref_mcl, args.out)


if __name__ == "__main__":
    main()

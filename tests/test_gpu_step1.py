"""GPU parity tests of the ONE-LAUNCH step (CARLE_STEP_DEFERRED, include/carle_hip.h): 64x64 batches of more than 16
instances run CARLE.step (carle/env.py:188-242) as a single kernel -- master reset decided inside the step kernel, zero
reward and reward moments written by it, counters settled lazily.  Checked against
  (1) the golden cases recorded from the genuine reference, replicated along the batch axis (every replica must follow
      the recorded grids and counters, master reset included),
  (2) the C oracle and the two-launch path (step kernel + carle_ca_finish) on crafted action sequences: all ones
      everywhere, all ones but one element, all-ones grids among others, non-binary values, windows of 8 ... 64,
  (3) batches larger than the resident wavefronts (the in-kernel verification must never wait for a wavefront that
      has not started), uint8 state, mixtures with run() / multistep() / the epilogue paths.
Bit-exact everywhere."""
import numpy as np
import pytest
import torch

import golden_util as G
from oracle import ca as O

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _one_launch_form(monkeypatch):
    """These tests are about the one-launch step: an experiment knob that forces the two-launch form everywhere
    (tools/test_all_paths.sh runs the suite under CARLE_TWO_LAUNCH_STEP=1) is lifted for their duration."""
    import os
    from carle_amd import hip
    forced = os.environ.get("CARLE_TWO_LAUNCH_STEP")
    if forced is not None:
        monkeypatch.delenv("CARLE_TWO_LAUNCH_STEP")
        hip.reload_knobs()
    yield
    if forced is not None:
        monkeypatch.setenv("CARLE_TWO_LAUNCH_STEP", forced)
        hip.reload_knobs()


def _env(**kw):
    from carle_amd.env import CARLE
    return CARLE(device="cuda:0", **kw)


def _cases64():
    names = [nm for nm in G.ca_case_names() if G.ca_case(nm).size == 64 and G.ca_case(nm).meta["action_kind"] != "oversize"
             and G.ca_case(nm).steps <= 64]
    return names + ["pred:" + nm for nm in G.predicate_case_names() if G.predicate_case(nm).size == 64]


@pytest.mark.parametrize("storage", ["f32", "u8", "b1"])
@pytest.mark.parametrize("name", _cases64())
def test_replicated_golden_case_through_one_launch_step(name, storage):
    """Every recorded 64x64 case, tiled to more than 16 instances: each replica reproduces the reference's grids, the
    counters follow the recording (an all-ones action is all ones in every replica: the master reset fires)."""
    pred = name.startswith("pred:")
    c = G.predicate_case(name[5:]) if pred else G.ca_case(name)
    if pred and "cancel" not in name:
        pytest.skip("mean == 1.0 without all ones: exact predicates only (test_gpu_ca.py)")
    reps = 17 // c.n + 1
    n = reps * c.n
    if storage == "b1" and c.asize != 64:
        pytest.skip("packed state: 64-column windows")
    env = _env(width=64, height=64, action_width=c.asize, action_height=c.asize, instances=n, storage=storage)
    env.birth, env.survive = list(c.birth), list(c.survive)
    env.reset()
    env.universe = torch.from_numpy(np.tile(c.init, (reps, 1, 1, 1))).cuda()
    rec = 0
    for t in range(c.steps):
        a = torch.from_numpy(np.tile(c.actions[t], (reps, 1, 1, 1))).cuda()
        obs, reward, done, info = env.step(a)
        if (t + 1) in c.record:
            got = env.universe.float().cpu().numpy()
            assert np.array_equal(got, np.tile(c.obs[rec], (reps, 1, 1, 1))), f"grid mismatch at step {t + 1}"
            assert float(reward.abs().sum()) == 0.0
            assert env.step_number == c.step_number[t]
            if not pred:                      # cancelling values: the documented default-predicate deviation
                assert env.steps_since_action == c.steps_since_action[t]
            rec += 1
    assert rec == len(c.record)


def _oracle_loop(u, actions, rule):
    """-> list of (grid, step_number, steps_since_action, resets) after every step (kernel predicates)."""
    sn = ssa = rs = 0
    out = []
    for a in actions:
        if a is None or not (a != 0).any():
            ssa += 1
        nxt, reset, _ = O.c_env_step(u, a, rule)
        if reset:
            sn, ssa, rs = 0, 0, rs + 1
        else:
            sn += 1
        u = nxt
        out.append((u, sn, ssa, rs))
    return out


def _crafted_actions(rng, n, aw):
    ones = np.ones((n, 1, aw, aw), np.float32)
    acts = []
    acts.append((rng.random((n, 1, aw, aw)) < 0.1).astype(np.float32))
    acts.append(np.zeros((n, 1, aw, aw), np.float32))
    acts.append(ones.copy())                                             # master reset
    a = ones.copy(); a[n - 1, 0, aw - 1, aw - 1] = 0.0; acts.append(a)    # all ones but the very last element
    a = ones.copy(); a[0, 0, 0, 0] = 2.0; acts.append(a)                 # ... but the very first, and non-zero
    a = (rng.random((n, 1, aw, aw)) < 0.5).astype(np.float32); a[min(1, n - 1)] = 1.0; a[n // 2] = 1.0; a[n - 1] = 1.0
    if n == 1:
        a[0, 0, 5, 5] = 0.0
    acts.append(a)                                                       # all-ones grids among ordinary ones
    a = ones.copy(); a[n // 3, 0, aw // 2, 3] = np.float32("nan"); acts.append(a)
    acts.append(ones.copy())                                             # and a second reset
    a = (rng.random((n, 1, aw, aw)) < 0.3).astype(np.float32)
    a[a != 0] = rng.choice(np.array([1.0, 0.5, -1.0, np.inf], np.float32), int((a != 0).sum()))
    acts.append(a)
    acts.append(None)
    a = ones.copy(); a[:, :, :, 0] = -0.0; acts.append(a)
    a = ones.copy(); a[:, :, 1:, :] = (rng.random((n, 1, aw - 1, aw)) < 0.5)     # first window row all ones, the rest not
    acts.append(a.astype(np.float32))
    a = ones.copy(); a[n - 1, 0, aw - 1, aw // 2] = 0.0; acts.append(a)           # ... every row but the very last element's
    return acts


@pytest.mark.parametrize("storage", ["f32", "u8"])
@pytest.mark.parametrize("n,aw,size", [(17, 64, 64), (100, 64, 64), (1030, 32, 64), (64, 8, 64), (333, 48, 64),
                                       (1, 64, 256), (16, 64, 256), (5, 128, 256), (3, 64, 512), (2, 256, 512), (40, 64, 256)])
def test_one_launch_step_against_oracle_and_two_launch_path(n, aw, size, storage):
    """64x64: wavefront-per-grid kernel; 256 / 512: wavefront-per-strip kernel (the verdict on the master reset comes from
    a sample of the grid's action, the first window row, in every workgroup)."""
    from carle_amd import hip
    rng = np.random.default_rng(n * 100 + aw + size)
    rule = O.rule_mask([3, 6], [2, 3])
    u0 = (rng.random((n, 1, size, size)) < 0.4).astype(np.float32)
    acts = _crafted_actions(rng, n, aw)
    ref = _oracle_loop(u0, acts, rule)
    dt = torch.float32 if storage == "f32" else torch.uint8
    for deferred in (True, False):
        st = hip.new_status("cuda:0")
        mom_out = torch.zeros(3, dtype=torch.float64, device="cuda")
        mom_acc = torch.zeros(3, dtype=torch.float64, device="cuda")
        st[12:16] = torch.tensor([mom_out.data_ptr(), mom_acc.data_ptr()], dtype=torch.int64).view(torch.int32).cuda()
        cur = torch.from_numpy(u0).cuda().to(dt)
        for t, a in enumerate(acts):
            nxt = torch.full_like(cur, 7)
            reward = torch.full((n, 1), 3.0, device="cuda")
            d_a = None if a is None else torch.from_numpy(a).cuda()
            if storage == "f32":
                hip.env_step_f32(cur, d_a, nxt, reward, rule, st, t & 1, True, True, deferred=deferred)
            else:
                hip.env_step_u8(cur, d_a, nxt, reward, rule, st, t & 1, True, True, deferred=deferred)
            exp, sn, ssa, rs = ref[t]
            assert np.array_equal(nxt.float().cpu().numpy(), exp), (deferred, t)
            assert float(reward.abs().sum()) == 0.0
            raw = st.cpu()
            if deferred and (size != 64 or n > 16):
                assert raw[5].item() == (0 if a is None else 3 | (4 if t & 1 else 0)), (t, raw.tolist())
            s = hip.env_settle(st).cpu()
            assert s[5].item() == 0 and s[16:20].tolist() == [0, 0, 0, 0], s.tolist()
            assert int(s[6:8].view(torch.int64)) == sn and int(s[8:10].view(torch.int64)) == ssa, (deferred, t)
            assert int(s[10:12].view(torch.int64)) == rs and s[4].item() == int(rs > (ref[t - 1][3] if t else 0))
            assert mom_acc.cpu().tolist() == [float(n * (t + 1)), 0.0, 0.0] and mom_out.cpu().tolist() == [float(n), 0.0, 0.0]
            cur = nxt


def test_counters_settle_through_the_next_step_without_an_explicit_settle():
    """Unsettled steps chain: every launch applies its predecessor's bookkeeping (no carle_env_settle in between)."""
    n = 40
    env = _env(width=64, height=64, instances=n)
    env.reset()
    rng = np.random.default_rng(5)
    u0 = (rng.random((n, 1, 64, 64)) < 0.3).astype(np.float32)
    env.universe = torch.from_numpy(u0).cuda()
    acts = _crafted_actions(rng, n, 64) * 2
    ref = _oracle_loop(u0, acts, env.rule_mask())
    for t, a in enumerate(acts):
        obs = env.step(torch.zeros(n, 1, 64, 64, device="cuda") if a is None else torch.from_numpy(a).cuda())[0]
        assert np.array_equal(obs.cpu().numpy(), ref[t][0]), t
    exp_ssa = ref[-1][2]
    assert (env.step_number, env.master_resets) == (ref[-1][1], ref[-1][3])
    assert env.steps_since_action == exp_ssa


@pytest.mark.parametrize("storage", ["f32", "u8"])
def test_master_reset_of_a_batch_larger_than_the_resident_wavefronts(storage):
    """20000 wavefronts (the chip holds 8192 at most): an all-ones action everywhere zeroes every grid, all ones but
    the last element of the last grid steps normally -- the wavefronts that verify the batch never wait for one that
    has not been dispatched."""
    from carle_amd import hip
    n = 20000
    rule = O.rule_mask([3], [2, 3])
    g = torch.Generator(device="cuda").manual_seed(3)
    dt = torch.float32 if storage == "f32" else torch.uint8
    u = (torch.rand(n, 1, 64, 64, device="cuda", generator=g) < 0.4).to(dt)
    a = torch.ones(n, 1, 64, 64, device="cuda", dtype=dt)
    out = torch.full_like(u, 5)
    reward = torch.empty(n, 1, device="cuda")
    st = hip.new_status("cuda:0")
    step = hip.env_step_f32 if storage == "f32" else hip.env_step_u8
    step(u, a, out, reward, rule, st, 0)
    assert int(out.count_nonzero()) == 0
    s = hip.env_settle(st).cpu()
    assert s[4].item() == 1 and int(s[10:12].view(torch.int64)) == 1 and int(s[6:8].view(torch.int64)) == 0
    for where in ((n - 1, 63, 63), (0, 0, 0), (n // 2, 31, 7)):
        a2 = a.clone()
        a2[where[0], 0, where[1], where[2]] = 0
        out.fill_(5)
        step(u, a2, out, reward, rule, st, 1)
        idx = sorted({0, 1, where[0], n // 2, n - 1})
        exp, reset, _ = O.c_env_step(u[idx].float().cpu().numpy(), a2[idx].float().cpu().numpy(), rule)
        assert np.array_equal(out[idx].float().cpu().numpy(), exp)
        # toggling every cell but one, then one generation: the same as the complement stepped
        ref = torch.empty_like(u)
        hip.ca_step_f32(u, a2, ref, rule, None, 0) if storage == "f32" else hip.ca_step_u8(u, a2, ref, rule, None, 0)
        assert torch.equal(out, ref)
        s = hip.env_settle(st).cpu()
        assert s[4].item() == 0 and s[16:20].tolist() == [0, 0, 0, 0]
        st[0:4] = 0


def test_one_launch_steps_mixed_with_run_multistep_and_exact_predicates():
    n = 48
    rng = np.random.default_rng(11)
    u0 = (rng.random((n, 1, 64, 64)) < 0.35).astype(np.float32)
    a1 = (rng.random((n, 1, 64, 64)) < 0.1).astype(np.float32)
    ones = np.ones((n, 1, 64, 64), np.float32)
    env = _env(width=64, height=64, instances=n)
    env.reset()
    env.universe = torch.from_numpy(u0).cuda()
    rule = env.rule_mask()
    seq = [a1, None, None, None, a1, ones, a1, None, None, a1]
    ref = _oracle_loop(u0, seq, rule)
    env.step(torch.from_numpy(a1).cuda())                       # one launch, pending
    env.multistep(3)                                            # settles, + 3
    assert (env.step_number, env.steps_since_action) == (ref[3][1], ref[3][2])
    env.run(actions=torch.from_numpy(np.stack([a1, ones, a1])).cuda())
    assert np.array_equal(env.universe.cpu().numpy(), ref[6][0])
    env.run(steps=2)
    obs = env.step(torch.from_numpy(a1).cuda())[0]
    assert np.array_equal(obs.cpu().numpy(), ref[9][0])
    assert (env.step_number, env.steps_since_action, env.master_resets) == ref[9][1:]
    # exact predicates (flags stored before the launch): the kernel reads them instead of deciding itself
    env2 = _env(width=64, height=64, instances=n, exact_action_predicates=True)
    env2.reset()
    env2.universe = torch.from_numpy(u0).cuda()
    half = np.full((n, 1, 64, 64), 0.5, np.float32); half[:, :, ::2] = 1.5      # mean exactly 1.0, not all ones
    o = env2.step(torch.from_numpy(a1).cuda())[0]
    assert np.array_equal(o.cpu().numpy(), ref[0][0])
    o = env2.step(torch.from_numpy(half).cuda())[0]
    assert int(o.count_nonzero()) == 0 and env2.step_number == 0 and env2.master_resets == 1


def test_step_number_assignment_survives_a_pending_step():
    env = _env(width=64, height=64, instances=20)
    env.reset()
    env.step(torch.ones(20, 1, 64, 64, device="cuda") * 0.5)
    env.step_number = 100                                       # settles first: the pending step is not re-applied
    env.step(torch.zeros(20, 1, 64, 64, device="cuda"))
    assert env.step_number == 101 and env.steps_since_action == 1


# ---------------------------------------------------------------------------------------------------------------
# RND2D.step on a 64x64 fp32 CARLE (BASELINE configs[3]): CA step + bonus + reward (+ reward moments) in ONE launch
# ---------------------------------------------------------------------------------------------------------------

def _rnd_env(n, aw, stats, two_launch, monkeypatch):
    from carle_amd import hip, dist as cd
    from carle_amd.env import CARLE
    from carle_amd.mcl import RND2D
    if two_launch:
        monkeypatch.setenv("CARLE_TWO_LAUNCH_STEP", "1")
    else:
        monkeypatch.delenv("CARLE_TWO_LAUNCH_STEP", raising=False)
    hip.reload_knobs()
    torch.manual_seed(12)
    env = CARLE(width=64, height=64, action_width=aw, action_height=aw, instances=n, device="cuda:0")
    w = RND2D(env, train_predictor=False)
    w.reward_scale = 0.75
    w.eval()
    st = None
    if stats:
        st = cd.RunningRewardStats("cuda:0", sync_every=4)
        assert st.attach(w)
    w.reset()
    return env, w, st


@pytest.mark.parametrize("n,aw,stats", [(17, 64, True), (100, 64, False), (1030, 32, True), (2050, 64, True)])
def test_one_launch_rnd_step_equals_two_launch_path_and_oracle(n, aw, stats, monkeypatch):
    """carle_env_step_f32_rnd with CARLE_STEP_DEFERRED (default through RND2D.step) against the step kernel + epilogue
    pair: observations (also against the C oracle), rewards, counters bit-identical; the reward statistics -- merged
    one call later from per-workgroup partial sums -- equal to the epilogue's to fp64 rounding.  Crafted actions:
    master resets (every reward becomes scale * bonus(all-dead grid)), all ones but one element, all-ones grids."""
    from carle_amd import hip
    rng = np.random.default_rng(n + aw)
    u0 = (rng.random((n, 1, 64, 64)) < 0.4).astype(np.float32)
    acts = [a for a in _crafted_actions(rng, n, aw) if a is not None]
    res = []
    try:
        for two in (False, True):
            env, w, st = _rnd_env(n, aw, stats, two, monkeypatch)
            env.universe = torch.from_numpy(u0).cuda()
            out = []
            for t, a in enumerate(acts):
                obs, reward, done, info = w.step(torch.from_numpy(a).cuda())
                if st is not None:
                    assert w._moments_merged
                    st.after_step(w, reward)
                out.append((obs.clone(), reward.clone()))
                if not two and st is None:
                    assert env._status[5].item() != 0            # pending: no epilogue ran
            snap = None
            if st is not None:
                st.flush()
                snap = (st.state.cpu().numpy().copy(), st.last_batch.cpu().numpy().copy())
            res.append((out, env.step_number, env.steps_since_action, env.master_resets, snap))
    finally:
        monkeypatch.delenv("CARLE_TWO_LAUNCH_STEP", raising=False)
        hip.reload_knobs()
    ref = _oracle_loop(u0, acts, O.rule_mask([3], [2, 3]))
    (o1, sn1, ssa1, r1, m1), (o2, sn2, ssa2, r2, m2) = res
    dead = None
    for t, ((oa, ra), (ob, rb)) in enumerate(zip(o1, o2)):
        assert np.array_equal(oa.cpu().numpy(), ref[t][0]), t
        assert torch.equal(oa, ob) and torch.equal(ra, rb), (t, (ra - rb).abs().max())
        if int(oa.count_nonzero()) == 0:                         # a master reset: one reward value for every instance
            assert float(ra.max()) == float(ra.min()) > 0
            dead = float(ra[0])
    assert dead is not None
    assert (sn1, ssa1, r1) == (sn2, ssa2, r2) == ref[-1][1:]
    if stats:
        allr = torch.cat([r.reshape(-1) for _, r in o1]).double().cpu().numpy()
        np.testing.assert_allclose(m1[0], m2[0], rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(m1[1], m2[1], rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(m1[0], [allr.size, allr.mean(), ((allr - allr.mean()) ** 2).sum()], rtol=1e-10)


def test_pending_rnd_moments_are_merged_by_whatever_comes_next(monkeypatch):
    """The partial sums a one-launch CA + RND2D step leaves behind are merged by the next launch on the status block:
    another wrapper step, a bare step of the inner environment, run(), multistep(), a stats flush -- same totals."""
    n = 64
    totals = []
    for follow in ("wrapper", "bare", "run", "multistep", "flush"):
        env, w, st = _rnd_env(n, 64, True, False, monkeypatch)
        g = torch.Generator(device="cuda").manual_seed(1)
        env.universe = (torch.rand(n, 1, 64, 64, device="cuda", generator=g) < 0.3).float()
        a = (torch.rand(n, 1, 64, 64, device="cuda", generator=g) < 0.1).float()
        rewards = []
        for t in range(3):
            rewards.append(w.step(a)[1].reshape(-1).double())
        if follow == "wrapper":
            rewards.append(w.step(a)[1].reshape(-1).double())
        elif follow == "bare":
            env.step(a)                                     # (the statistics are attached to the wrapper: not counted)
        elif follow == "run":
            env.run(steps=2)
        elif follow == "multistep":
            env.multistep(2)
        if follow == "flush":
            st._pending_steps = 3
            st.flush()
            got = st.state.cpu().numpy()
        else:
            env.settled_status()
            got = st._pending.cpu().numpy()
        allr = torch.cat(rewards).cpu().numpy()
        assert got[0] == allr.size, (follow, got, allr.size)
        np.testing.assert_allclose(got[1:], [allr.mean(), ((allr - allr.mean()) ** 2).sum()], rtol=1e-10, err_msg=follow)
        totals.append(got)



# ---------------------------------------------------------------------------------------------------------------
# bit-packed state (BASELINE configs[4]): the whole step in one launch for no action / the 64-column window
# ---------------------------------------------------------------------------------------------------------------

def _np_pack(a):
    n, _, h, w = a.shape
    bits = np.packbits((a != 0).reshape(n, 1, h, w // 64, 64), axis=-1, bitorder="little")
    return bits.reshape(n, 1, h, w // 64, 8).copy().view("<u8").reshape(n, 1, h, w // 64).astype(np.int64)


@pytest.fixture
def _b1_walk(request, monkeypatch):
    """"single": the one-launch packed step walks single strips (CARLE_B1_DBG bit 3, LDS strip modes forced): strips
    outside the window decide the master reset from a sample of the grid's action record."""
    from carle_amd import hip
    if request.param == "single":
        monkeypatch.setenv("CARLE_B1_DBG", "8")
        monkeypatch.setenv("CARLE_B1_LDS_MIN_BYTES", "0")
        monkeypatch.setenv("CARLE_B1_NT_MIN_BYTES", str(1 << 60))
    hip.reload_knobs()
    yield request.param
    monkeypatch.delenv("CARLE_B1_DBG", raising=False)
    monkeypatch.delenv("CARLE_B1_LDS_MIN_BYTES", raising=False)
    monkeypatch.delenv("CARLE_B1_NT_MIN_BYTES", raising=False)
    hip.reload_knobs()


@pytest.mark.parametrize("n,size", [(1, 64), (9, 64), (40, 64), (3, 128), (5, 256), (2, 512), (33, 512)])
@pytest.mark.parametrize("publish", [True, False])
@pytest.mark.parametrize("_b1_walk", ["units", "single"], indirect=True)
def test_one_launch_packed_step_against_oracle_and_two_launch_path(n, size, publish, _b1_walk):
    """carle_env_step_b1 with CARLE_STEP_DEFERRED against the oracle and the packed step kernel + epilogue: packed
    64x64 action windows (flags derived from the bits by the kernel, or stored beforehand as carle_pack_f32_b1 does),
    crafted all-ones / all-but-one-bit / all-ones-grid cases, no action; grids, reward, counters, moments."""
    from carle_amd import hip
    if _b1_walk == "single" and size not in (256, 512):
        pytest.skip("single-strip walk: LDS strip modes (256 / 512 columns)")
    rng = np.random.default_rng(n * 10 + size)
    rule = O.rule_mask([3, 6], [2, 3])
    u0 = (rng.random((n, 1, size, size)) < 0.4).astype(np.float32)
    acts = [a if a is None else (a != 0).astype(np.float32) for a in _crafted_actions(rng, n, 64)]
    ref = _oracle_loop(u0, acts, rule)
    for deferred in (True, False):
        st = hip.new_status("cuda:0")
        mom_out = torch.zeros(3, dtype=torch.float64, device="cuda")
        mom_acc = torch.zeros(3, dtype=torch.float64, device="cuda")
        st[12:16] = torch.tensor([mom_out.data_ptr(), mom_acc.data_ptr()], dtype=torch.int64).view(torch.int32).cuda()
        cur = torch.from_numpy(_np_pack(u0)).cuda()
        for t, a in enumerate(acts):
            nxt = torch.full_like(cur, 7)
            reward = torch.full((n, 1), 3.0, device="cuda")
            slot = t & 1
            if a is None:
                d_a = None
            elif publish:
                d_a = torch.from_numpy(_np_pack(a)).cuda()
            else:
                d_a = hip.pack_f32_b1(torch.from_numpy(a).cuda(), st, slot)       # stores the flags of `slot`
            hip.env_step_b1(cur, d_a, nxt, reward, rule, st, slot, publish, True, deferred=deferred)
            exp, sn, ssa, rs = ref[t]
            assert np.array_equal(nxt.cpu().numpy(), _np_pack(exp)), (deferred, t)
            assert float(reward.abs().sum()) == 0.0
            s = hip.env_settle(st).cpu()
            assert s[5].item() == 0 and s[16:20].tolist() == [0, 0, 0, 0], s.tolist()
            assert int(s[6:8].view(torch.int64)) == sn and int(s[8:10].view(torch.int64)) == ssa, (deferred, t)
            assert int(s[10:12].view(torch.int64)) == rs and s[4].item() == int(rs > (ref[t - 1][3] if t else 0))
            assert mom_acc.cpu().tolist() == [float(n * (t + 1)), 0.0, 0.0]
            cur = nxt


@pytest.mark.parametrize("_b1_walk", ["units", "single"], indirect=True)
def test_packed_master_reset_of_a_batch_larger_than_the_resident_wavefronts(_b1_walk):
    from carle_amd import hip
    n, size = 9000, 256                                    # 18000 half-grid units (36000 strips), 8192 wavefronts resident
    rule = O.rule_mask([3], [2, 3])
    g = torch.Generator(device="cuda").manual_seed(3)
    u = hip.pack_f32_b1((torch.rand(n, 1, size, size, device="cuda", generator=g) < 0.4).float())
    a = torch.full((n, 1, 64, 1), -1, dtype=torch.int64, device="cuda")
    out = torch.full_like(u, 5)
    reward = torch.empty(n, 1, device="cuda")
    st = hip.new_status("cuda:0")
    hip.env_step_b1(u, a, out, reward, rule, st, 0)
    assert int(out.count_nonzero()) == 0
    s = hip.env_settle(st).cpu()
    assert s[4].item() == 1 and int(s[10:12].view(torch.int64)) == 1
    for where in ((n - 1, 63), (0, 0), (n // 2, 31)):
        a2 = a.clone()
        a2[where[0], 0, where[1], 0] = ~(1 << 17)
        ref = torch.empty_like(u)
        hip.ca_step_b1(u, a2, ref, rule, None, 0)
        out.fill_(5)
        hip.env_step_b1(u, a2, out, reward, rule, st, 1)
        assert torch.equal(out, ref) and int(out.count_nonzero()) > 0
        s = hip.env_settle(st).cpu()
        assert s[4].item() == 0 and s[16:20].tolist() == [0, 0, 0, 0]
        st[0:4] = 0


@pytest.mark.parametrize("n,asize", [(1, 64), (3, 32), (17, 64), (300, 48), (1025, 64), (33, 0)])
def test_one_launch_step_also_leaves_the_packed_copy_at_64(n, asize):
    """carle_env_step_f32_bits at 64x64 (round 5): the one-launch kernel transposes its nibbles into row words and
    stores them beside the fp32 state -- the same grids as without the copy, the copy equal to carle_pack_f32_b1 of
    them, also when the master reset fires (all-ones action) and with other rules."""
    from carle_amd import hip
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cuda").manual_seed(1000 + n)
    for rule in (hip.rule_mask([3], [2, 3]), hip.rule_mask([3, 6], [2, 3]), hip.rule_mask([1, 3, 5, 7], [0, 2, 4, 6, 8])):
        u = (torch.rand(n, 1, 64, 64, device=dev, generator=g) < 0.4).float()
        for t in range(4):
            if asize:
                act = (torch.rand(n, 1, asize, asize, device=dev, generator=g) < 0.2).float()
                if t == 2:
                    act = torch.ones_like(act)                 # master reset (env.py:208-216)
            else:
                act = None
            outs = []
            for with_bits in (False, True):
                st = hip.new_status(dev)
                out = torch.empty_like(u)
                reward = torch.full((n, 1), 7.0, device=dev)
                bits = torch.full((n, 1, 64, 1), -1, dtype=torch.int64, device=dev) if with_bits else None
                wrote = hip.env_step_f32(u, act, out, reward, rule, st, t & 1, True, False, bits_out=bits)
                assert wrote == with_bits
                assert float(reward.abs().max()) == 0.0
                outs.append((out, bits, hip.env_settle(st).clone()))
            assert torch.equal(outs[0][0], outs[1][0])
            assert torch.equal(outs[1][1], hip.pack_f32_b1(outs[1][0]))
            # last-reset flag and the three counters (the action flags of the slot are transient: N <= 16 without the copy
            # runs the small-batch kernel, which does not keep them)
            assert torch.equal(outs[0][2][4:5], outs[1][2][4:5]) and torch.equal(outs[0][2][6:12], outs[1][2][6:12])
            if asize and t == 2:
                assert float(outs[1][0].sum()) == 0.0 and int(outs[1][1].abs().sum()) == 0
            u = outs[1][0]


def test_reward_nets_on_a_64_column_environment_read_the_packed_copy(monkeypatch):
    """An AE2D directly on a CARLE(64x64), and an RND2D in training mode: the environment's step hands them the observation
    bit-packed (CARLE.packed_copy_of), the rewards equal those of the same stacks reading the fp32 cells (CARLE_PACKED_COPY=0
    at construction) bit for bit."""
    import os
    from carle_amd.env import CARLE
    from carle_amd.mcl import AE2D, RND2D

    import carle_amd.env as E

    def run(kind, packed):
        monkeypatch.setattr(E, "_PACKED_STATE", packed)      # (else the step leaves the copy for its own next step anyway)
        old = os.environ.get("CARLE_PACKED_COPY")
        os.environ["CARLE_PACKED_COPY"] = "1" if packed else "0"
        try:
            torch.manual_seed(5)
            env = CARLE(width=64, height=64, instances=40, device="cuda:0")
            w = AE2D(env) if kind == "ae" else RND2D(env)
            if kind == "ae":
                w.eval()
            w.batch_size = 3
        finally:
            if old is None:
                del os.environ["CARLE_PACKED_COPY"]
            else:
                os.environ["CARLE_PACKED_COPY"] = old
        g = torch.Generator(device="cuda").manual_seed(9)
        w.reset()
        env.universe = (torch.rand(40, 1, 64, 64, device="cuda", generator=g) < 0.3).float()
        rs, seen = [], 0
        for t in range(7):
            a = (torch.rand(40, 1, 64, 64, device="cuda", generator=g) < 0.1).float()
            obs, r, _, _ = w.step(a)
            seen += env.packed_copy_of(obs) is not None
            rs.append(r.clone())
        return torch.stack(rs), seen, obs.clone()

    for kind in ("ae", "rnd_train"):
        r1, seen1, o1 = run(kind, True)
        r0, seen0, o0 = run(kind, False)
        assert seen1 == 7 and seen0 == 0
        assert torch.equal(o1, o0)
        assert torch.equal(r1, r0), (kind, float((r1 - r0).abs().max()))
        assert float(r1.abs().max()) > 0


def test_step_reads_the_packed_copy_of_its_own_last_output_and_nothing_else(monkeypatch):
    """Round 5: at 64x64 (N > 16) CARLE.step reads its input state from the bit-packed copy the previous step left
    (carle_env_step_f32_packed) -- only while ``universe`` is the very tensor that step returned, untouched since.  An
    in-place edit, an assignment, a non-binary state, reset(): each falls back to the fp32 cells, and the trajectory equals
    that of an environment that never takes the packed path, bit for bit (master reset included)."""
    import carle_amd.env as E
    from carle_amd import hip
    calls = {"packed": 0}
    real = hip.env_step_f32_packed

    def spy(*a, **k):
        calls["packed"] += 1
        return real(*a, **k)
    monkeypatch.setattr(hip, "env_step_f32_packed", spy)

    def trajectory(packed_state):
        monkeypatch.setattr(E, "_PACKED_STATE", packed_state)
        g = torch.Generator(device="cuda").manual_seed(77)
        env = E.CARLE(width=64, height=64, instances=40, device="cuda:0")
        env.reset()
        env.universe = (torch.rand(40, 1, 64, 64, device="cuda", generator=g) < 0.35).float()
        out, taken = [], []
        for t in range(14):
            a = (torch.rand(40, 1, 64, 64, device="cuda", generator=g) < 0.1).float()
            if t == 5:
                a = torch.ones_like(a)                                   # master reset
            if t == 3:
                env.universe[7, 0, 10:20, 10:20] = 1.0 - env.universe[7, 0, 10:20, 10:20]     # in-place edit: version bump
            if t == 7:
                env.universe = env.universe.clone()                      # another tensor with the same cells
            if t == 9:
                env.universe = torch.rand(40, 1, 64, 64, device="cuda", generator=g)          # non-binary state
            if t == 11:
                env.reset()
            before = calls["packed"]
            obs = env.step(a)[0]
            taken.append(calls["packed"] - before)
            out.append(obs.clone())
        return out, taken, (env.step_number, env.steps_since_action)

    ref, taken_ref, counters_ref = trajectory(False)
    got, taken, counters = trajectory(True)
    assert taken_ref == [0] * 14
    #        t = 0 (assigned state)  1  2  3 (edited)  4  5  6  7 (clone)  8  9 (new state)  10  11 (reset)  12  13
    assert taken == [0, 1, 1, 0, 1, 1, 1, 0, 1, 0, 1, 0, 1, 1], taken
    for t, (a, b) in enumerate(zip(ref, got)):
        assert torch.equal(a, b), t
    assert counters == counters_ref
    assert float(got[5].sum()) == 0.0 and float(got[4].sum()) > 0 and float(got[6].sum()) > 0     # the master reset at t = 5


@pytest.mark.parametrize("n,asize,grids", [(17, 64, None), (300, 32, None), (4099, 64, None), (65, 0, None), (3500, 64, None),
                                           (17, 64, "2"), (301, 32, "2"), (2, 8, "2"), (3500, 48, "1")])
def test_packed_input_step_against_the_oracle(n, asize, grids, monkeypatch):
    """carle_env_step_f32_packed itself: state in as row words, fp32 state + row words out; twelve steps against the C oracle
    on the fp32 cells, three rules.  ``grids``: CARLE_PACKED_GRIDS forced (two grids per wavefront is the library's choice for
    3072 < n <= 4096 only: ca_step_64_packed2_kernel; odd batches leave the last wavefront with one grid)."""
    from carle_amd import hip
    if grids is not None:
        monkeypatch.setenv("CARLE_PACKED_GRIDS", grids)
    else:
        monkeypatch.delenv("CARLE_PACKED_GRIDS", raising=False)
    hip.reload_knobs()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cuda").manual_seed(n)
    for birth, survive in (([3], [2, 3]), ([3, 6], [2, 3]), ([2], [])):
        rule = hip.rule_mask(birth, survive)
        u = (torch.rand(n, 1, 64, 64, device=dev, generator=g) < 0.4).float()
        bits = [hip.pack_f32_b1(u), torch.empty(n, 1, 64, 1, dtype=torch.int64, device=dev)]
        st = hip.new_status(dev)
        cpu = u.cpu().numpy()
        for t in range(12):
            act = (torch.rand(n, 1, asize, asize, device=dev, generator=g) < 0.15).float() if asize else None
            out = torch.empty(n, 1, 64, 64, device=dev)
            reward = torch.empty(n, 1, device=dev)
            hip.env_step_f32_packed(bits[t & 1], act, out, reward, rule, st, t & 1, True, False, bits_out=bits[(t & 1) ^ 1])
            cpu = O.c_env_step(cpu, None if act is None else act.cpu().numpy(), rule)[0]
            assert np.array_equal(out.cpu().numpy(), cpu), (birth, t)
            assert torch.equal(bits[(t & 1) ^ 1], hip.pack_f32_b1(out))
    monkeypatch.delenv("CARLE_PACKED_GRIDS", raising=False)
    hip.reload_knobs()


def test_fused_rnd_step_reads_the_packed_state_too(monkeypatch):
    """RND2D.step (forward only, 64x64, N > 16): from its second step on the fused kernel takes the state as the row words its
    previous launch left (carle_env_step_f32_rnd_packed); observations, rewards and counters equal those of the stack that
    always reads the fp32 cells, bit for bit -- through an in-place edit, an assigned state and a master reset."""
    import carle_amd.env as E
    from carle_amd import hip
    from carle_amd.mcl import RND2D
    calls = {"packed": 0}
    lib_call = hip.env_step_f32_rnd

    def spy(*a, **k):
        calls["packed"] += k.get("bits_in") is not None
        return lib_call(*a, **k)
    monkeypatch.setattr(hip, "env_step_f32_rnd", spy)

    def trajectory(packed_state):
        monkeypatch.setattr(E, "_PACKED_STATE", packed_state)
        torch.manual_seed(11)
        g = torch.Generator(device="cuda").manual_seed(78)
        env = E.CARLE(width=64, height=64, instances=33, device="cuda:0")
        w = RND2D(env, train_predictor=False)
        w.eval()
        w.reset()
        env.universe = (torch.rand(33, 1, 64, 64, device="cuda", generator=g) < 0.35).float()
        obs_l, rew_l, taken = [], [], []
        for t in range(10):
            a = (torch.rand(33, 1, 64, 64, device="cuda", generator=g) < 0.1).float()
            if t == 4:
                a = torch.ones_like(a)
            if t == 2:
                env.universe[3, 0, :8, :8] = 1.0
            if t == 6:
                env.universe = (torch.rand(33, 1, 64, 64, device="cuda", generator=g) < 0.5).float()
            before = calls["packed"]
            obs, r, _, _ = w.step(a)
            taken.append(calls["packed"] - before)
            obs_l.append(obs.clone())
            rew_l.append(r.clone())
        return obs_l, rew_l, taken, (env.step_number, env.steps_since_action)

    o0, r0, t0, c0 = trajectory(False)
    o1, r1, t1, c1 = trajectory(True)
    assert t0 == [0] * 10 and t1 == [0, 1, 0, 1, 1, 1, 0, 1, 1, 1], (t0, t1)
    for t in range(10):
        assert torch.equal(o0[t], o1[t]) and torch.equal(r0[t], r1[t]), t
    assert c0 == c1 and float(o1[4].sum()) == 0.0


@pytest.mark.parametrize("n,size,asize", [(3, 256, 64), (17, 256, 32), (129, 256, 64), (2, 512, 64), (18, 512, 0), (33, 512, 128)])
def test_wide_steps_with_the_packed_copy_against_the_oracle(n, size, asize):
    """256 / 512-column fp32 grids (round 5): carle_env_step_f32_bits leaves the packed copy in the one-launch form AND in the
    two-launch form (129 x 256x256 = 33.8 MiB per array: beyond the one-launch step), carle_env_step_f32_packed reads it back.
    Eight steps against the C oracle, the copy equal to carle_pack_f32_b1 of every output, a master reset in the middle."""
    from carle_amd import hip
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cuda").manual_seed(n * size)
    birth, survive = [3, 6], [2, 3]
    rule = hip.rule_mask(birth, survive)
    u = (torch.rand(n, 1, size, size, device=dev, generator=g) < 0.4).float()
    bits = [torch.empty(n, 1, size, size // 64, dtype=torch.int64, device=dev) for _ in (0, 1)]
    st = hip.new_status(dev)
    cpu = u.cpu().numpy()
    for t in range(8):
        act = None
        if asize:
            act = (torch.rand(n, 1, asize, asize, device=dev, generator=g) < 0.15).float()
            if t == 4:
                act = torch.ones_like(act)
        out = torch.empty_like(u)
        reward = torch.full((n, 1), 3.0, device=dev)
        if t == 0:
            assert hip.env_step_f32(u, act, out, reward, rule, st, t & 1, True, False, bits_out=bits[1])
        else:
            hip.env_step_f32_packed(bits[t & 1], act, out, reward, rule, st, t & 1, True, False, bits_out=bits[(t & 1) ^ 1])
        cpu = O.c_env_step(cpu, None if act is None else act.cpu().numpy(), rule)[0]
        assert np.array_equal(out.cpu().numpy(), cpu), t
        assert torch.equal(bits[(t & 1) ^ 1], hip.pack_f32_b1(out)), t
        assert float(reward.abs().max()) == 0.0
        u = out
    hip.env_settle(st)
    assert int(st[6]) == (3 if asize else 8)          # CARLE_ST_STEP_NUMBER: the master reset at t = 4 restarts the count


def test_wide_environment_reads_its_packed_state(monkeypatch):
    """CARLE(256x256), N = 5 (one launch) and N = 130 (two launches): the packed path from the second step on, the
    trajectory of the environment that never takes it, an in-place edit and an assignment in between."""
    import carle_amd.env as E
    from carle_amd import hip
    calls = {"packed": 0}
    real = hip.env_step_f32_packed

    def spy(*a, **k):
        calls["packed"] += 1
        return real(*a, **k)
    monkeypatch.setattr(hip, "env_step_f32_packed", spy)
    for n in (5, 130):
        def trajectory(packed_state):
            monkeypatch.setattr(E, "_PACKED_STATE", packed_state)
            g = torch.Generator(device="cuda").manual_seed(n)
            env = E.CARLE(width=256, height=256, instances=n, device="cuda:0")
            env.reset()
            env.universe = (torch.rand(n, 1, 256, 256, device="cuda", generator=g) < 0.3).float()
            outs, taken = [], []
            for t in range(7):
                a = (torch.rand(n, 1, 64, 64, device="cuda", generator=g) < 0.1).float()
                if t == 3:
                    env.universe[0, 0, 100:110, 100:110] = 1.0
                if t == 5:
                    env.universe = env.universe.clone()
                before = calls["packed"]
                outs.append(env.step(a)[0].clone())
                taken.append(calls["packed"] - before)
            return outs, taken, (env.step_number, env.steps_since_action)
        ref, t0, c0 = trajectory(False)
        got, t1, c1 = trajectory(True)
        assert t0 == [0] * 7 and t1 == [0, 1, 1, 0, 1, 0, 1], (n, t1)
        assert c0 == c1
        for a, b in zip(ref, got):
            assert torch.equal(a, b)


def test_a_step_captured_into_a_hip_graph_reads_the_fp32_state():
    """A caller's own torch.cuda.graph around env.step: the replay runs on whatever the captured input buffer holds by then,
    so the capture must record the kernel that READS that buffer -- not the packed copy of what it held at capture time."""
    import carle_amd.env as E
    g = torch.Generator(device="cuda").manual_seed(5)
    env = E.CARLE(width=64, height=64, instances=40, device="cuda:0", packed_state=True)     # (whatever CARLE_PACKED_STATE says)
    env.reset()
    env.universe = (torch.rand(40, 1, 64, 64, device="cuda", generator=g) < 0.3).float()
    act = (torch.rand(40, 1, 64, 64, device="cuda", generator=g) < 0.1).float()
    for _ in range(3):
        env.step(act)                                  # the packed path is live now
    static_in = env.universe
    assert env.packed_copy_of(static_in) is not None
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            obs = env.step(act)[0]
    torch.cuda.current_stream().wait_stream(side)
    rule = env.rule_mask()
    for _ in range(3):
        new_state = (torch.rand(40, 1, 64, 64, device="cuda", generator=g) < 0.4).float()
        static_in.copy_(new_state)
        graph.replay()
        torch.cuda.synchronize()
        want = O.c_env_step(new_state.cpu().numpy(), act.cpu().numpy(), rule)[0]
        assert np.array_equal(obs.cpu().numpy(), want)

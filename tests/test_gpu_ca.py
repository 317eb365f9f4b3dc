"""GPU parity tests for the CA step: the HIP path (through the C ABI) against
  (1) every golden case recorded from the genuine reference (tests/golden/ca_cases.npz),
  (2) the RNG-free known answers (ca_known.json),
  (3) the CPU oracle (oracle/ca_oracle.c) on seeded random inputs at awkward batch sizes,
  (4) size-independent properties at BASELINE.json's full size (N=4096, 64x64).
Bit-exact everywhere (integer/boolean work carried in fp32)."""
import numpy as np
import pytest
import torch

import golden_util as G
from oracle import ca as O

pytestmark = pytest.mark.gpu


def _env(**kw):
    from carle_amd.env import CARLE
    return CARLE(device="cuda:0", **kw)


@pytest.mark.parametrize("name", G.ca_case_names())
def test_env_replays_golden_case(name):
    c = G.ca_case(name)
    env = _env(width=c.size, height=c.size, action_width=c.asize, action_height=c.asize,
               instances=c.n)
    env.birth, env.survive = list(c.birth), list(c.survive)
    env.reset()
    env.universe = torch.from_numpy(c.init.copy()).cuda()
    rec = 0
    for t in range(c.steps):
        a = torch.from_numpy(c.actions[t].copy())
        if t % 2:
            a = a.cuda()                      # actions arrive from the CPU or the device
        obs, reward, done, info = env.step(a)
        assert obs is env.universe and obs.is_cuda and obs.dtype == torch.float32
        assert tuple(reward.shape) == (c.n, 1) and reward.is_cuda
        assert tuple(done.shape) == (c.n, 1) and not done.is_cuda and len(info) == c.n
        if (t + 1) in c.record:
            got = obs.cpu().numpy()
            assert np.array_equal(got, c.obs[rec]), f"grid mismatch at step {t + 1}"
            assert float(reward.abs().sum()) == 0.0
            assert env.step_number == c.step_number[t]
            assert env.steps_since_action == c.steps_since_action[t]
            rec += 1
    assert rec == len(c.record)


def _replay(c, env, to_device):
    """-> list of (obs, step_number, steps_since_action) at the recorded steps."""
    env.birth, env.survive = list(c.birth), list(c.survive)
    env.reset()
    env.universe = torch.from_numpy(c.init.copy()).cuda()
    out = []
    for t in range(c.steps):
        a = torch.from_numpy(c.actions[t].copy())
        obs = env.step(a.cuda() if to_device else a)[0]
        if (t + 1) in c.record:
            o = env.universe.float() if obs.dtype != torch.float32 or obs.shape[-1] != c.size else obs
            out.append((o.cpu().numpy(), env.step_number, env.steps_since_action))
    return out


@pytest.mark.parametrize("storage", ["f32", "u8", "b1"])
@pytest.mark.parametrize("name", G.predicate_case_names())
def test_exact_action_predicates_follow_the_reference(name, storage):
    """Actions outside {0,1}: values that cancel (torch.sum(action) == 0 although cells toggle,
    env.py:191) and a mean of exactly 1.0 without all ones (env.py:208 -> master reset).
    CARLE(exact_action_predicates=True) reproduces the reference's grids AND counters."""
    c = G.predicate_case(name)
    env = _env(width=c.size, height=c.size, action_width=c.asize, action_height=c.asize, instances=c.n,
               storage=storage, exact_action_predicates=True)
    got = _replay(c, env, to_device=True)
    for i, (o, sn, ssa) in enumerate(got):
        t = c.record[i] - 1
        assert np.array_equal(o, c.obs[i]), f"grid mismatch at step {t + 1}"
        assert sn == c.step_number[t] and ssa == c.steps_since_action[t], (t, sn, ssa)


@pytest.mark.parametrize("name", G.predicate_case_names())
def test_default_action_predicates_documented_deviation(name):
    """The default (in-kernel) predicates are 'some element non-zero' / 'every element == 1.0'
    (DESIGN.md 4.4).  On the cancelling actions the grids still equal the reference's (the toggle
    is the same) and only steps_since_action differs; an action of mean 1.0 that is not all ones
    does NOT reset.  Pinned against the oracle driven with those two predicates."""
    c = G.predicate_case(name)
    env = _env(width=c.size, height=c.size, action_width=c.asize, action_height=c.asize, instances=c.n)
    got = _replay(c, env, to_device=False)
    rule = O.rule_mask(c.birth, c.survive)
    u, sn, ssa = c.init.copy(), 0, 0
    for t in range(c.steps):
        a = c.actions[t]
        if not (a != 0).any():
            ssa += 1
        if (a == 1.0).all():
            u, sn, ssa = np.zeros_like(u), 0, 0
        else:
            u = O.np_ca_update(O.np_apply_action(u, a), rule).astype(np.float32)
            sn += 1
        o, gsn, gssa = got[t]
        assert np.array_equal(o, u) and (gsn, gssa) == (sn, ssa), t
        if "cancel" in name:
            assert np.array_equal(o, c.obs[t])           # same grids as the reference
    if "cancel" in name:
        assert got[-1][2] < c.steps_since_action[-1]     # the counter is where they differ


def _pops(cells, birth, survive, at):
    env = _env(width=64, height=64)
    env.birth, env.survive = birth, survive
    env.reset()
    a = torch.zeros(1, 1, 64, 64)
    for r, c in cells:
        a[0, 0, r, c] = 1.0
    z = torch.zeros(1, 1, 64, 64, device="cuda")
    res = {}
    for t in range(1, max(at) + 1):
        obs = env.step(a if t == 1 else z)[0]
        if t in at:
            res[str(t)] = int(obs.sum().item())
    return res, env


def test_known_answers():
    k = G.known()
    at = [1, 2, 3, 4, 11, 51, 101, 256]
    res, env = _pops(k["glider_cells"], [3], [2, 3], at)
    assert res == k["glider_life_pop"] and env.step_number == k["glider_life_step_number"]
    assert _pops(k["rpent_cells"], [3], [2, 3], at)[0] == k["rpent_life_pop"]
    assert _pops(k["rpent_cells"], [3, 6], [2, 3], at)[0] == k["rpent_highlife_pop"]
    assert _pops(k["puffer_cells"], [3, 6, 8], [2, 4, 5], at)[0] == k["puffer_morley_pop"]
    for key_in, key_out in (("seam_in", "seam_out"), ("seam2_in", "seam2_out")):
        env = _env(width=64, height=64)
        env.reset()
        a = torch.zeros(1, 1, 64, 64)
        for r, c in k[key_in]:
            a[0, 0, r, c] = 1.0
        obs = env.step(a)[0]
        assert obs[0, 0].nonzero().cpu().tolist() == k[key_out]


def test_master_reset_and_counters():
    k = G.known()["master_reset"]
    env = _env(width=64, height=64, instances=2)
    env.reset()
    env.step((torch.rand(2, 1, 64, 64) < 0.3).float())
    env.step(torch.zeros(2, 1, 64, 64))
    assert env.step_number == k["step_number_before"] and env.steps_since_action == 1
    obs = env.step(torch.ones(2, 1, 64, 64))[0]
    assert float(obs.sum()) == k["obs_sum"]
    assert env.step_number == k["step_number_after"]
    assert env.steps_since_action == k["steps_since_action_after"]
    assert env.master_resets == 1
    # the reference's own test (tests/test_env.py:42-67)
    env = _env()
    reset_obs = env.reset()
    action = torch.ones(env.instances, 1, env.action_height, env.action_width)
    toggle_obs = env.step(action)[0]
    action[:, :, 0:10, 0:10] = 0.0
    normal_obs = env.step(action)[0]
    assert toggle_obs.mean().item() == 0.0 and reset_obs.mean().item() == 0.0
    assert (1.0 * (reset_obs == toggle_obs)).mean().item() == 1.0
    assert (1.0 * (toggle_obs == normal_obs)).mean().item() != 1.0


@pytest.mark.parametrize("n,size,asize,rule", [
    (1, 64, 64, ([3], [2, 3])), (5, 64, 64, ([3, 6], [2, 3])), (37, 64, 64, ([3, 6, 8], [2, 4, 5])),
    (7, 64, 32, ([3], [2, 3])), (3, 64, 48, ([1, 3, 5, 7], [0, 2, 4, 6, 8])), (2, 64, 62, ([3], [2, 3])),
    (3, 128, 64, ([3], [2, 3])), (2, 256, 64, ([3, 6], [2, 3])), (1, 512, 64, ([3], [2, 3])),
    (2, 20, 10, ([3], [2, 3])), (1, 2, 2, ([1], [1])), (3, 66, 64, ([2], [])),
])
def test_abi_against_oracle_random(n, size, asize, rule):
    """carle_ca_step_f32 called directly (no env) vs the C oracle, with and without action."""
    from carle_amd import hip
    rng = np.random.default_rng(n * 1000 + size)
    mask = O.rule_mask(*rule)
    u = (rng.random((n, 1, size, size)) < 0.4).astype(np.float32)
    u[u != 0] = rng.choice(np.array([1.0, 2.0, -3.5, np.nan, 1e-40], np.float32), int((u != 0).sum()))
    a = (rng.random((n, 1, asize, asize)) < 0.2).astype(np.float32)
    a[a != 0] = rng.choice(np.array([1.0, 0.5, -1.0, np.inf], np.float32), int((a != 0).sum()))
    a[0, 0, 0, 0] = -0.0
    d_u, d_a = torch.from_numpy(u).cuda(), torch.from_numpy(a).cuda()
    out = torch.empty_like(d_u)
    st = hip.new_status("cuda:0")
    for act_np, act_d in ((a, d_a), (None, None)):
        exp, _, flags = O.c_env_step(u, act_np, mask)
        out.fill_(7.0)
        st.zero_()
        hip.ca_step_f32(d_u, act_d, out, mask, st, 1)
        assert np.array_equal(out.cpu().numpy(), exp)
        if act_np is not None:
            s = st.cpu().tolist()
            assert s[1] == 1 and s[3] == 1 and s[0] == 0 and s[2] == 0   # slot 1 only


def test_full_size_properties():
    """BASELINE configs[1]: N=4096, 64x64, B3/S23.  Size-independent checks + sampled oracle."""
    from carle_amd import hip
    n = 4096
    g = torch.Generator(device="cuda").manual_seed(1234)
    u = (torch.rand(n, 1, 64, 64, device="cuda", generator=g) < 0.5).float()
    a = (torch.rand(n, 1, 64, 64, device="cuda", generator=g) <= 0.1).float()
    life = O.rule_mask([3], [2, 3])
    out = torch.empty_like(u)
    hip.ca_step_f32(u, a, out, life, None, 0)
    # (1) sampled instances against the oracle
    idx = [0, 1, 2047, 2048, 4094, 4095, 777]
    exp, _, _ = O.c_env_step(u[idx].cpu().numpy(), a[idx].cpu().numpy(), life)
    assert np.array_equal(out[idx].cpu().numpy(), exp)
    # (2) batch-permutation equivariance: instances are independent
    perm = torch.randperm(n, device="cuda", generator=g)
    out_p = torch.empty_like(u)
    hip.ca_step_f32(u[perm].contiguous(), a[perm].contiguous(), out_p, life, None, 0)
    assert torch.equal(out_p, out[perm])
    # (3) translation equivariance on the torus (rolls commute with the step)
    out_r = torch.empty_like(u)
    hip.ca_step_f32(torch.roll(u, (5, -9), (2, 3)).contiguous(),
                    torch.roll(a, (5, -9), (2, 3)).contiguous(), out_r, life, None, 0)
    assert torch.equal(out_r, torch.roll(out, (5, -9), (2, 3)))
    # (4) toggling twice is the identity: step(u ^ a, a) == step(u, 0)
    ua = torch.logical_xor(u, a).float()
    o1, o2 = torch.empty_like(u), torch.empty_like(u)
    hip.ca_step_f32(ua, a, o1, life, None, 0)
    hip.ca_step_f32(u, None, o2, life, None, 0)
    assert torch.equal(o1, o2)
    # (5) output alphabet
    assert set(torch.unique(out).cpu().tolist()) <= {0.0, 1.0}
    # (6) rule B012345678/S012345678 -> all alive; empty rule -> all dead
    hip.ca_step_f32(u, a, out, (1 << 18) - 1, None, 0)
    assert float(out.min()) == 1.0
    hip.ca_step_f32(u, a, out, 0, None, 0)
    assert float(out.max()) == 0.0


def test_apply_action_standalone():
    env = _env(width=256, height=256, instances=2)
    env.reset()
    u = (torch.rand(2, 1, 256, 256) < 0.5).float()
    a = (torch.rand(2, 1, 64, 64) < 0.5).float()
    env.universe = u.clone()
    env.apply_action(a)
    exp = O.np_apply_action(u.numpy(), a.numpy()).astype(np.float32)
    assert np.array_equal(env.universe.cpu().numpy(), exp)


def test_action_forms():
    """2-D / 3-D / broadcast actions (env.py:152-156) and the 2-D known answer."""
    k = G.known()
    env = _env(width=64, height=64)
    env.reset()
    a2 = torch.zeros(64, 64)
    a2[10, 10:13] = 1
    assert env.step(a2)[0][0, 0].nonzero().cpu().tolist() == k["action_2d_out"]
    env = _env(width=64, height=64, instances=3)
    env.reset()
    obs = env.step(a2.unsqueeze(0))[0]          # (1,64,64) broadcast over 3 instances
    for i in range(3):
        assert obs[i, 0].nonzero().cpu().tolist() == k["action_2d_out"]
    with pytest.raises(AssertionError):
        env.step(torch.zeros(3, 1, 32, 32))


def test_logging_and_rle_round_trip(tmp_path):
    """logging=True (env.py:194-195,466-491): every step records [action rle, universe rle];
    load_universe (env.py:390-406) restores a pattern from a file."""
    env = _env(width=64, height=64, instances=2, logging=True)
    env.reset()
    a = torch.zeros(2, 1, 64, 64)
    a[0, 0, 10, 10:13] = 1
    env.step(a)
    env.step(torch.zeros(2, 1, 64, 64))
    assert len(env.log) == 2 and "(action)" in env.log[0][0] and "(universe)" in env.log[0][1]
    assert "step=1 (universe)" in env.log[1][1] and "rule = B3/S23:T64, 64" in env.log[1][1]
    env.save_log(str(tmp_path))
    env.save_frame(str(tmp_path))
    assert any(f.startswith("carle_log") for f in __import__("os").listdir(tmp_path))
    p = tmp_path / "blinker.rle"
    p.write_text("x = 3, y = 1, rule = B3/S23\n3o!\n")
    env.load_universe(str(p), universe_index=1)
    assert env.universe[1, 0].nonzero().cpu().tolist() == [[0, 0], [0, 1], [0, 2]]
    obs = env.step(torch.zeros(2, 1, 64, 64))[0]
    assert obs[1, 0].nonzero().cpu().tolist() == [[0, 1], [1, 1], [63, 1]]


def test_unindexed_cuda_device_and_use_cuda():
    """device="cuda" / use_cuda=True (carle/env.py:28-37): counters must survive across steps."""
    from carle_amd.env import CARLE
    for kw in (dict(device="cuda"), dict(use_cuda=True), dict()):
        env = CARLE(width=64, height=64, instances=2, **kw)
        assert env.my_device.type == "cuda" and env.my_device.index is not None
        env.reset()
        for _ in range(3):
            env.step(torch.zeros(2, 1, 64, 64))
        assert env.step_number == 3 and env.steps_since_action == 3


def test_index_width_beyond_2_31_cells():
    """BASELINE configs[2] has N*H*W = 2^31 cells: offsets must be 64-bit.  N=32769 x 256x256
    (8 GiB per array) through the ABI; first / middle / last instances against the oracle."""
    from carle_amd import hip
    n = 32769
    g = torch.Generator(device="cuda").manual_seed(7)
    u = torch.empty(n, 1, 256, 256, device="cuda")
    for i in range(0, n, 4096):                       # fill in slices: bounded temporaries
        m = min(4096, n - i)
        u[i:i + m] = (torch.rand(m, 1, 256, 256, device="cuda", generator=g) < 0.4).float()
    a = (torch.rand(n, 1, 64, 64, device="cuda", generator=g) < 0.1).float()
    out = torch.empty_like(u)
    life = O.rule_mask([3], [2, 3])
    hip.ca_step_f32(u, a, out, life, None, 0)
    idx = [0, 1, 16383, 16384, 32767, 32768]
    exp, _, _ = O.c_env_step(u[idx].cpu().numpy(), a[idx].cpu().numpy(), life)
    assert np.array_equal(out[idx].cpu().numpy(), exp)
    del u, out, a
    torch.cuda.empty_cache()


@pytest.mark.parametrize("n,size", [(4099, 256), (1027, 512)])
def test_wide_full_size_properties(n, size):
    """BASELINE configs[2] shape (256x256) and 512x512 at full batch, ragged N, dense 64x64 action with arbitrary
    non-zero values: sampled instances against the C oracle, batch-permutation equivariance, toggle-twice identity,
    and the action predicates.  (Four strips per workgroup share edge rows through LDS; the window is read in a
    compact mapping: ca_step_cells.hip.)"""
    from carle_amd import hip
    g = torch.Generator(device="cuda").manual_seed(n + size)
    u = (torch.rand(n, 1, size, size, device="cuda", generator=g) < 0.45).float()
    a = (torch.rand(n, 1, 64, 64, device="cuda", generator=g) < 0.5).float()
    a[3] = 1.0                                     # one instance with an all-ones window (no reset: others differ)
    a[5] = 0.0
    a[7, 0, 0, :] = -2.5                           # first / last window rows: the strips' halo rows see them
    a[7, 0, 63, :] = float("nan")
    rule = O.rule_mask([3, 6], [2, 3])
    out = torch.empty_like(u)
    st = hip.new_status("cuda:0")
    hip.ca_step_f32(u, a, out, rule, st, 0)
    idx = [0, 1, 2, 3, 5, 7, n // 2, n - 2, n - 1]
    exp, _, _ = O.c_env_step(u[idx].cpu().numpy(), a[idx].cpu().numpy(), rule)
    assert np.array_equal(out[idx].cpu().numpy(), exp)
    s_ = st.cpu().tolist()
    assert s_[0] == 1 and s_[2] == 1
    perm = torch.randperm(n, device="cuda", generator=g)
    out_p = torch.empty_like(u)
    hip.ca_step_f32(u[perm].contiguous(), a[perm].contiguous(), out_p, rule, None, 0)
    assert torch.equal(out_p, out[perm])
    del out_p
    pad = (size - 64) // 2
    ua = torch.logical_xor(u, torch.nn.functional.pad(a != 0, (pad, pad, pad, pad))).float()
    o1 = torch.empty_like(u)
    hip.ca_step_f32(ua, a, o1, rule, None, 0)
    hip.ca_step_f32(u, None, out, rule, None, 0)
    assert torch.equal(o1, out)
    # all-ones action everywhere: the reset predicate stays clear (NOT_ALL_ONES == 0)
    st.zero_()
    hip.ca_step_f32(u, torch.ones_like(a), o1, rule, st, 1)
    s_ = st.cpu().tolist()
    assert s_[1] == 1 and s_[3] == 0
    del u, a, ua, o1, out
    torch.cuda.empty_cache()


def test_zero_instances_and_tiny_batches():
    env = _env(width=64, height=64, instances=0)
    obs = env.reset()
    assert tuple(obs.shape) == (0, 1, 64, 64)
    obs, reward, done, info = env.step(torch.zeros(0, 1, 64, 64))
    assert tuple(obs.shape) == (0, 1, 64, 64) and tuple(reward.shape) == (0, 1) and info == []
    assert env.step_number == 1


@pytest.mark.parametrize("storage", ["f32", "u8", "b1"])
def test_long_run_against_oracle(storage):
    """Soak: 600 steps with a fresh random action every step (every 97th all-zero, every 211th
    all-ones = master reset), three rules changed on the fly as train_mcl.py does; grids and counters
    equal the C oracle at every checkpoint."""
    n, steps = 6, 600
    rng = np.random.default_rng(11)
    kw = dict(storage=storage, obs="f32") if storage != "f32" else {}
    env = _env(width=64, height=64, instances=n, **kw)
    env.reset()
    cur = np.zeros((n, 1, 64, 64), np.float32)
    rules = [([3], [2, 3]), ([3, 6], [2, 3]), ([3, 6, 8], [2, 4, 5])]
    step_number = since = 0
    for t in range(steps):
        if t % 150 == 0:
            b, s = rules[(t // 150) % 3]
            env.birth, env.survive = list(b), list(s)
            mask = O.rule_mask(b, s)
        a = (rng.random((n, 1, 64, 64)) < 0.05).astype(np.float32)
        if t % 97 == 96:
            a[:] = 0.0
        if t % 211 == 210:
            a[:] = 1.0
        obs = env.step(torch.from_numpy(a).cuda())[0]
        nxt, reset, flags = O.c_env_step(cur, a, mask)
        since = since + (0 if a.any() else 1)
        if reset:
            cur, step_number, since = np.zeros_like(cur), 0, 0
        else:
            cur, step_number = nxt, step_number + 1
        if t % 50 == 49 or reset or t == steps - 1:
            assert np.array_equal(obs.float().cpu().numpy(), cur), f"step {t}"
            assert env.step_number == step_number and env.steps_since_action == since
    assert env.master_resets == 2


def test_random_geometries_and_rules_hypothesis():
    """Property test: for arbitrary even sizes, centred windows, 18-bit rule masks, densities and
    batch sizes the fp32 and uint8 step kernels (fast and generic paths) equal the C oracle."""
    from hypothesis import given, settings, strategies as st, HealthCheck
    from carle_amd import hip

    @settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck))
    @given(st.data())
    def run(data):
        half = data.draw(st.integers(1, 40))
        size = 2 * half if data.draw(st.booleans()) else data.draw(st.sampled_from([64, 64, 128, 256, 512]))
        ahalf = data.draw(st.integers(1, size // 2))
        asize = 2 * ahalf if (size - 2 * ahalf) % 2 == 0 else size
        n = data.draw(st.integers(1, 5))
        rule = data.draw(st.integers(0, (1 << 18) - 1))
        dens = data.draw(st.sampled_from([0.0, 0.05, 0.3, 0.5, 0.9, 1.0]))
        seed = data.draw(st.integers(0, 2 ** 31 - 1))
        rng = np.random.default_rng(seed)
        u = (rng.random((n, 1, size, size)) < dens).astype(np.float32)
        a = (rng.random((n, 1, asize, asize)) < 0.2).astype(np.float32)
        exp, _, _ = O.c_env_step(u, a, rule)
        d_u, d_a = torch.from_numpy(u).cuda(), torch.from_numpy(a).cuda()
        out = torch.empty_like(d_u)
        hip.ca_step_f32(d_u, d_a, out, rule, None, 0)
        assert np.array_equal(out.cpu().numpy(), exp), (size, asize, n, hex(rule))
        out8 = torch.empty(n, 1, size, size, dtype=torch.uint8, device="cuda")
        hip.ca_step_u8(d_u.to(torch.uint8), d_a.to(torch.uint8), out8, rule, None, 0)
        assert np.array_equal(out8.cpu().numpy(), exp.astype(np.uint8)), (size, asize, n, hex(rule))
        if size in (64, 128, 256, 512) and asize % 64 == 0:      # 512: always the LDS strip modes of the packed kernel
            bits, abits = hip.pack_f32_b1(d_u), hip.pack_f32_b1(d_a)
            outb = torch.empty_like(bits)
            hip.ca_step_b1(bits, abits, outb, rule, None, 0)
            assert np.array_equal(hip.unpack_b1_f32(outb).cpu().numpy(), exp), (size, asize, n, hex(rule))

    run()


@pytest.mark.parametrize("name", ["tog64a64_life", "reset64_life", "tog64a32_life", "tog256a64_daynight",
                                  "reset256_life", "values64_life", "cfg1_life", "odd32a32_life"])
def test_run_replays_golden_case(name):
    """CARLE.run (carle_env_run_f32: k steps per library call; N <= 16 at 64x64 = one launch per
    step) replays the reference's golden cases in chunks: grids and counters at every recorded step."""
    c = G.ca_case(name)
    env = _env(width=c.size, height=c.size, action_width=c.asize, action_height=c.asize, instances=c.n)
    env.birth, env.survive = list(c.birth), list(c.survive)
    env.reset()
    env.universe = torch.from_numpy(c.init.copy()).cuda()
    t, rec = 0, 0
    marks = sorted(set(c.record))
    for m in marks:
        acts = torch.from_numpy(np.ascontiguousarray(c.actions[t:m]))
        obs, reward, done, info = env.run(actions=acts)
        t = m
        assert obs is env.universe and tuple(reward.shape) == (c.n, 1) and len(info) == c.n
        assert np.array_equal(obs.cpu().numpy(), c.obs[rec]), f"grid mismatch at step {m}"
        assert env.step_number == c.step_number[m - 1]
        assert env.steps_since_action == c.steps_since_action[m - 1]
        assert float(reward.abs().sum()) == 0.0
        rec += 1


@pytest.mark.parametrize("n,size", [(1, 64), (16, 64), (40, 64), (2, 256), (3, 20)])
@pytest.mark.parametrize("source", ["actions", "agent", "none"])
def test_run_never_writes_an_observation_the_caller_holds(n, size, source):
    """The reference returns a fresh tensor per step (env.py:229): an observation the caller kept from an earlier
    step() / reset() / run() must still show ITS generation after run(k) for even and odd k (every launch form:
    one-launch small batches, one-launch action-free runs, the per-step loop)."""
    from carle_amd.agents import RandomAgent
    asz = min(64, size)
    if source == "agent" and size == 20:
        pytest.skip("in-kernel toggles need an aligned window")
    env = _env(width=size, height=size, action_width=asz, action_height=asz, instances=n)
    env.reset()
    g = torch.Generator().manual_seed(n + size)
    env.universe = (torch.rand(n, 1, size, size, generator=g) < 0.4).float().cuda()
    agent = RandomAgent(action_width=asz, action_height=asz, fused=True, seed=3)
    held = []                                     # (tensor the caller holds, what it showed when it was returned)
    obs = env.step((torch.rand(n, 1, asz, asz, generator=g) < 0.1).float())[0]
    held.append((obs, obs.clone()))
    for k in (2, 3, 1, 4):
        if source == "actions":
            obs = env.run(actions=(torch.rand(k, n, 1, asz, asz, generator=g) < 0.1).float())[0]
        elif source == "agent":
            obs = env.run(agent=agent, steps=k)[0]
        else:
            obs = env.run(steps=k)[0]
        assert obs is env.universe
        for t_, snap in held:
            assert torch.equal(t_, snap), f"run({k}) wrote into an observation returned earlier"
            assert t_.data_ptr() != obs.data_ptr()
        held.append((obs, obs.clone()))


@pytest.mark.parametrize("n,size", [(1, 64), (16, 64), (17, 64), (300, 64), (3, 256)])
def test_run_with_fused_random_agent_equals_step_loop(n, size):
    """env.run(agent=RandomAgent, steps=k) == k x env.step(agent(obs)): same Philox draws, same
    grids, counters, and the agent's call counter advances by k."""
    from carle_amd.agents import RandomAgent
    outs = []
    for mode in ("loop", "run"):
        env = _env(width=size, height=size, instances=n)
        env.reset()
        g = torch.Generator().manual_seed(7)
        env.universe = (torch.rand(n, 1, size, size, generator=g) < 0.4).float().cuda()
        agent = RandomAgent(action_width=64, action_height=64, fused=True, seed=11)
        if mode == "loop":
            for t in range(9):
                obs = env.step(agent(env.universe))[0]
        else:
            env.run(agent=agent, steps=4)
            obs = env.run(agent=agent, steps=5)[0]
            env.run(steps=0)
        assert agent.calls == 9
        outs.append((obs.cpu().numpy(), env.step_number, env.steps_since_action))
        # and then a plain step still works (flag slot parity carried over)
        env.step(torch.zeros(n, 1, 64, 64))
        assert env.steps_since_action == 1
    assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][1:] == outs[1][1:] == (9, 0)


@pytest.mark.parametrize("n", [5, 40, 1030])
def test_run_without_action_equals_multistep(n):
    """run(steps=k) without an action source: all k generations in one launch (n <= 16: the small-batch kernel, above:
    ca_run_noaction64_kernel, one wave per grid) against the step-by-step evolution; in place for even k."""
    env_a, env_b = _env(width=64, height=64, instances=n), _env(width=64, height=64, instances=n)
    u = (torch.rand(n, 1, 64, 64) < 0.5).float()
    for e in (env_a, env_b):
        e.reset()
        e.universe = u.clone().cuda()
    oa = env_a.run(steps=12)[0]
    ob = env_b.multistep(12)[0]
    assert torch.equal(oa, ob) and env_a.step_number == env_b.step_number == 12
    assert env_a.steps_since_action == env_b.steps_since_action == 12
    oa = env_a.run(steps=7)[0]
    for _ in range(7):
        ob = env_b.step(torch.zeros(n, 1, 64, 64))[0]
    assert torch.equal(oa, ob) and env_a.step_number == env_b.step_number == 19
    assert env_a.steps_since_action == env_b.steps_since_action == 19
    assert env_a.settled_status().cpu().tolist()[4:12] == env_b.settled_status().cpu().tolist()[4:12]      # counters, last-reset word
    # and an action afterwards behaves as usual (flag slots / parity carried over)
    a = (torch.rand(n, 1, 64, 64) < 0.2).float()
    assert torch.equal(env_a.step(a)[0], env_b.step(a)[0])
    assert env_a.steps_since_action == env_b.steps_since_action == 19 and env_a.step_number == env_b.step_number == 20


def test_c_consumer_steps_on_the_gpu(tmp_path):
    """The boundary used the way a non-Python host would: tests/c_abi/abi_step_gpu.c (gcc -std=c99, HIP runtime C
    API, no Python / torch in the process) steps a batch through carle_env_step_f32 and carle_ca_step_b1 and
    compares every cell with oracle/ca_oracle.c, incl. the master reset and the status-block counters."""
    import os
    import shutil
    import subprocess
    from carle_amd import _lib
    _lib.load()
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(_lib.LIB_PATH)
    odir = os.path.dirname(O.build())
    exe = tmp_path / "abi_step_gpu"
    subprocess.run([gcc, "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                    "-I", "/opt/rocm/include", os.path.join(root, "tests", "c_abi", "abi_step_gpu.c"), "-o", str(exe),
                    "-L", libdir, "-lcarle_hip", "-L", odir, "-lca_oracle", "-L", "/opt/rocm/lib", "-lamdhip64",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath," + odir, "-Wl,-rpath,/opt/rocm/lib"],
                   check=True, capture_output=True, text=True)
    # the consumer checks the default (one-launch, deferred) form of the step: no experiment knobs in its environment
    child_env = {k: v for k, v in os.environ.items() if not k.startswith("CARLE_")}
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120, env=child_env)
    assert out.returncode == 0 and "abi gpu ok" in out.stdout, out.stdout + out.stderr


def test_abi_from_two_host_threads():
    """include/carle_hip.h: "asynchronous on the passed stream ... safe from several host threads on different streams".
    Two host threads, each with its own HIP stream and its own environment (fp32 one-launch step / bit-packed step),
    step concurrently through the C ABI (ctypes releases the GIL inside every call): both trajectories bit-identical to
    the oracle, counters right, and the thread-local error text of one thread is never the other's."""
    import threading
    from carle_amd import _lib
    lib = _lib.load()
    errors, done = [], {}
    start = threading.Barrier(2)

    def run(tid):
        try:
            rng = np.random.default_rng(500 + tid)
            n, steps = 23 + 18 * tid, 48
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                env = _env(width=64, height=64, instances=n, storage="f32" if tid == 0 else "b1",
                           **({"obs": "f32"} if tid else {}))
                env.rules_from_string("B3/S23" if tid == 0 else "B36/S23")
                env.reset()
                u = (rng.random((n, 1, 64, 64)) < 0.4).astype(np.float32)
                env.universe = torch.from_numpy(u).cuda()
                mask = O.rule_mask(env.birth, env.survive)
                acts = [(rng.random((n, 1, 64, 64)) < 0.1).astype(np.float32) for _ in range(steps)]
                dev_acts = [torch.from_numpy(a).cuda() for a in acts]
                stream.synchronize()
                start.wait(timeout=120)
                got = []
                for t in range(steps):
                    got.append(env.step(dev_acts[t])[0])
                    if t % 8 == tid:
                        # an argument error in between: its text must be THIS thread's (thread-local storage)
                        if tid == 0:
                            rc = lib.carle_ca_step_f32(None, None, None, 1, 64, 64, 0, 0, 0, None, 0, None)
                            assert rc != 0 and b"NULL" in lib.carle_last_error(), lib.carle_last_error()
                        else:
                            rc = lib.carle_instance_sums_f32(got[-1].data_ptr(), -1, 1, None, None, None, got[-1].data_ptr(), None)
                            assert rc != 0 and b"negative" in lib.carle_last_error(), lib.carle_last_error()
                stream.synchronize()
                for t in range(steps):
                    u, reset, _ = O.c_env_step(u, acts[t], mask)
                    assert not reset
                    assert np.array_equal(got[t].cpu().numpy(), u), (tid, t)
                assert env.step_number == steps
            done[tid] = True
        except BaseException as e:                           # noqa: BLE001 -- reported by the main thread
            errors.append((tid, repr(e)))
            try:
                start.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(600)
    assert not errors, errors
    assert done == {0: True, 1: True}

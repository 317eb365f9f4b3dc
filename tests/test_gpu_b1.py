"""GPU parity tests of the bit-packed (CARLE_FMT_B1) path: pack / unpack / step, bit-exact
against numpy packing, the CPU oracle and the golden cases of the genuine reference."""
import numpy as np
import pytest
import torch

import golden_util as G
from oracle import ca as O

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _restore_knobs():
    yield
    from carle_amd import hip             # after monkeypatch has undone its setenv calls (fixtures unwind in reverse)
    hip.reload_knobs()


def np_pack(a):
    """(N,1,H,W) 0/1 -> (N,1,H,W/64) int64, bit j of word i = column 64*i+j."""
    n, _, h, w = a.shape
    bits = np.packbits((a != 0).reshape(n, 1, h, w // 64, 64), axis=-1, bitorder="little")
    return bits.reshape(n, 1, h, w // 64, 8).copy().view("<u8").reshape(n, 1, h, w // 64).astype(np.int64)


@pytest.mark.parametrize("n,size", [(1, 64), (3, 64), (130, 64), (2, 128), (2, 256), (1, 512)])
def test_pack_unpack(n, size):
    from carle_amd import hip
    rng = np.random.default_rng(size + n)
    a = (rng.random((n, 1, size, size)) < 0.5).astype(np.float32)
    a[a != 0] = rng.choice(np.array([1.0, -2.0, np.nan, 1e-40, 0.5], np.float32), int((a != 0).sum()))
    a[0, 0, 0, 0] = -0.0
    st = hip.new_status("cuda:0")
    packed = hip.pack_f32_b1(torch.from_numpy(a).cuda(), st, 1)
    assert np.array_equal(packed.cpu().numpy(), np_pack(a))
    assert st.cpu().tolist()[:4] == [0, 1, 0, 1]
    back = hip.unpack_b1_f32(packed).cpu().numpy()
    assert np.array_equal(back, (a != 0).astype(np.float32))
    st.zero_()
    hip.pack_f32_b1(torch.ones(n, 1, size, size, device="cuda"), st, 0)
    assert st.cpu().tolist()[:4] == [1, 0, 0, 0]
    st.zero_()
    hip.pack_f32_b1(torch.zeros(n, 1, size, size, device="cuda"), st, 0)
    assert st.cpu().tolist()[:4] == [0, 0, 1, 0]


@pytest.mark.parametrize("n,size,asize,rule", [
    (1, 64, 64, ([3], [2, 3])), (5, 64, 64, ([3, 6], [2, 3])), (3, 128, 64, ([3, 6, 8], [2, 4, 5])),
    (3, 128, 128, ([3], [2, 3])), (2, 256, 64, ([3], [2, 3])), (2, 256, 128, ([3, 6, 7, 8], [3, 4, 6, 7, 8])),
    (2, 256, 256, ([2], [0])), (1, 512, 64, ([3, 6], [2, 3])), (2, 512, 256, ([3], [2, 3])),
    (1, 512, 512, (list(range(9)), [])), (2, 512, 448, ([0, 3, 5, 7], [2, 4, 6, 8])),
])
@pytest.mark.parametrize("strip_path", ["by size", "rows", "lds", "lds+nt", "lds+rotate", "lds+single", "lds+nt+single"])
def test_step_b1_against_oracle(n, size, asize, rule, strip_path, monkeypatch):
    """The three strip modes of the kernel (ca_step_b1.hip: row-per-lane, StripLds cached, StripLds streaming) are
    chosen by array size; `rows` / `lds` / `lds+nt` force each of them onto these small cases."""
    from carle_amd import hip
    if strip_path != "by size" and size not in (256, 512):
        pytest.skip("StripLds exists for 256 and 512 columns")
    if strip_path == "rows":
        monkeypatch.setenv("CARLE_B1_LDS_MIN_BYTES", str(1 << 60))
    elif strip_path == "lds":
        monkeypatch.setenv("CARLE_B1_LDS_MIN_BYTES", "0")
        monkeypatch.setenv("CARLE_B1_NT_MIN_BYTES", str(1 << 60))
    elif strip_path == "lds+nt":
        monkeypatch.setenv("CARLE_B1_LDS_MIN_BYTES", "0")
        monkeypatch.setenv("CARLE_B1_NT_MIN_BYTES", "0")
    else:
        # experiment knob CARLE_B1_DBG (ca_step_b1.hip): bit 2 = the half-grid units of a window launch start at different
        # strips, bit 3 = single-strip walk instead of half grids
        monkeypatch.setenv("CARLE_B1_LDS_MIN_BYTES", "0")
        monkeypatch.setenv("CARLE_B1_NT_MIN_BYTES", "0" if "+nt" in strip_path else str(1 << 60))
        monkeypatch.setenv("CARLE_B1_DBG", "4" if "rotate" in strip_path else "8")
    hip.reload_knobs()                    # the library reads its knobs once; the fixture below restores them
    rng = np.random.default_rng(n * 7 + size + asize)
    mask = O.rule_mask(*rule)
    u = (rng.random((n, 1, size, size)) < 0.4).astype(np.float32)
    a = (rng.random((n, 1, asize, asize)) < 0.2).astype(np.float32)
    d_u = torch.from_numpy(np_pack(u)).cuda()
    d_a = torch.from_numpy(np_pack(a)).cuda()
    out = torch.empty_like(d_u)
    st = hip.new_status("cuda:0")
    for act_np, act_d in ((a, d_a), (None, None)):
        exp, _, _ = O.c_env_step(u, act_np, mask)
        out.fill_(-1)
        st.zero_()
        hip.ca_step_b1(d_u, act_d, out, mask, st, 0)
        assert np.array_equal(out.cpu().numpy(), np_pack(exp))
        if act_np is not None:
            assert st.cpu().tolist()[:4] == [1, 0, 1, 0]
    # several generations: packed evolution stays identical to the oracle
    cur_np, cur = u, d_u
    for t in range(5):
        nxt = torch.empty_like(cur)
        hip.ca_step_b1(cur, None, nxt, mask, None, 0)
        cur_np, _, _ = O.c_env_step(cur_np, None, mask)
        cur = nxt
    assert np.array_equal(cur.cpu().numpy(), np_pack(cur_np))


def _b1_cases():
    return [nm for nm in G.ca_case_names()
            if G.ca_case(nm).size in (64, 128, 256, 512) and G.ca_case(nm).asize % 64 == 0
            and G.ca_case(nm).meta["actions_binary"] and G.ca_case(nm).meta["action_kind"] != "oversize"]


@pytest.mark.parametrize("name", _b1_cases())
def test_b1_replays_golden_case(name):
    """pack(fp32 action, publishes the predicates) -> packed step -> finish; grids of the genuine
    reference reproduced bit for bit, master reset and counters included."""
    from carle_amd import hip
    c = G.ca_case(name)
    mask = O.rule_mask(c.birth, c.survive)
    cur = hip.pack_f32_b1(torch.from_numpy(c.init.copy()).cuda())
    st = hip.new_status("cuda:0")
    rec = 0
    for t in range(c.steps):
        slot = t & 1
        a = hip.pack_f32_b1(torch.from_numpy(c.actions[t].copy()).cuda(), st, slot)
        nxt = torch.empty_like(cur)
        hip.ca_step_b1(cur, a, nxt, mask, None, slot)
        hip.ca_finish(nxt, None, st, slot, True)
        cur = nxt
        if (t + 1) in c.record:
            got = hip.unpack_b1_f32(cur).cpu().numpy()
            assert np.array_equal(got, c.obs[rec]), f"grid mismatch at step {t + 1}"
            host = st.cpu()
            assert int(host[6:8].view(torch.int64)) == c.step_number[t]
            assert int(host[8:10].view(torch.int64)) == c.steps_since_action[t]
            rec += 1


def test_b1_full_size_round_trip():
    """BASELINE configs[4]: B36/S23, N=4096 would be 128 MiB/array; here N=512 at 512x512 and
    N=4096 at 64x64: packed step == pack(f32 step(unpack)) on the same inputs."""
    from carle_amd import hip
    high = O.rule_mask([3, 6], [2, 3])
    for n, size in ((512, 512), (4096, 64)):
        g = torch.Generator(device="cuda").manual_seed(size)
        u = (torch.rand(n, 1, size, size, device="cuda", generator=g) < 0.5).float()
        ub = hip.pack_f32_b1(u)
        o_b = torch.empty_like(ub)
        hip.ca_step_b1(ub, None, o_b, high, None, 0)
        o_f = torch.empty_like(u)
        hip.ca_step_f32(u, None, o_f, high, None, 0)
        assert torch.equal(hip.pack_f32_b1(o_f), o_b)
        assert torch.equal(hip.unpack_b1_f32(o_b), o_f)


@pytest.mark.parametrize("name", ["tog64a64_life", "tog256a64_daynight", "reset64_life", "reset256_life",
                                  "tog512a64_highlife", "crop256_life", "values64_life"])
def test_packed_env_replays_golden_case(name):
    """CARLE(storage="b1") through the gym surface: fp32 actions in, packed state inside."""
    from carle_amd.env import CARLE
    c = G.ca_case(name)
    for obs_fmt in ("b1", "f32"):
        env = CARLE(width=c.size, height=c.size, action_width=c.asize, action_height=c.asize,
                    instances=c.n, device="cuda:0", storage="b1", obs=obs_fmt)
        env.birth, env.survive = list(c.birth), list(c.survive)
        env.reset()
        env.universe = torch.from_numpy(c.init.copy())
        rec = 0
        for t in range(c.steps):
            obs, reward, done, info = env.step(torch.from_numpy(c.actions[t].copy()))
            if obs_fmt == "b1":
                assert obs.dtype == torch.int64 and tuple(obs.shape) == (c.n, 1, c.size, c.size // 64)
            if (t + 1) in c.record:
                assert np.array_equal(env.universe.cpu().numpy(), c.obs[rec])
                if obs_fmt == "f32":
                    assert np.array_equal(obs.cpu().numpy(), c.obs[rec])
                assert env.step_number == c.step_number[t]
                assert env.steps_since_action == c.steps_since_action[t]
                rec += 1


def test_packed_env_packed_actions():
    from carle_amd.env import CARLE
    from carle_amd import hip
    env = CARLE(width=64, height=64, instances=5, device="cuda:0", storage="b1")
    ref = CARLE(width=64, height=64, instances=5, device="cuda:0")
    env.reset(); ref.reset()
    for t in range(6):
        a = (torch.rand(5, 1, 64, 64, device="cuda") < 0.2).float()
        if t == 3:
            a.fill_(1.0)
        ob = env.step(hip.pack_f32_b1(a))[0]
        of = ref.step(a)[0]
        assert torch.equal(hip.unpack_b1_f32(ob), of)
        assert env.step_number == ref.step_number


@pytest.mark.parametrize("n,size,rule", [
    (3, 64, ([3], [2, 3])), (2, 64, ([3, 6], [2, 3])), (2, 64, ([3, 6, 8], [2, 4, 5])),
    (2, 128, ([3], [2, 3])), (2, 128, ([2], [0])), (2, 256, ([3, 6], [2, 3])),
    (1, 256, ([0, 3, 5, 7], [2, 4, 6, 8])), (1, 512, ([3], [2, 3])), (2, 512, ([3, 6], [2, 3])),
    (1, 512, ([3, 6, 7, 8], [3, 4, 6, 7, 8])),
])
def test_multistep_equals_single_steps(n, size, rule):
    """carle_ca_multistep_b1(k) == k generations of the C oracle (and of the single-step kernel);
    in place and out of place; counters advance by k."""
    from carle_amd import hip
    rng = np.random.default_rng(n + size)
    mask = O.rule_mask(*rule)
    u = (rng.random((n, 1, size, size)) < 0.35).astype(np.float32)
    d_u = torch.from_numpy(np_pack(u)).cuda()
    st = hip.new_status("cuda:0")
    exp = u
    done = 0
    cur = d_u.clone()
    for k in (0, 1, 2, 7, 33):
        for _ in range(k):
            exp, _, _ = O.c_env_step(exp, None, mask)
        out = torch.empty_like(cur)
        hip.ca_multistep_b1(cur, out, mask, k, st)
        done += k
        assert np.array_equal(out.cpu().numpy(), np_pack(exp)), f"after {done} generations"
        cur = out
    host = st.cpu()
    assert int(host[6:8].view(torch.int64)) == done and int(host[8:10].view(torch.int64)) == done
    # in place == out of place == repeated single steps
    a = d_u.clone()
    hip.ca_multistep_b1(a, a, mask, 5, None)
    b = d_u
    for _ in range(5):
        nxt = torch.empty_like(b)
        hip.ca_step_b1(b, None, nxt, mask, None, 0)
        b = nxt
    assert torch.equal(a, b)


def test_multistep_through_the_env():
    """CARLE.multistep(k) / CARLE(storage="b1").multistep(k): glider known answer (population 5
    after 256 generations, ca_known.json) and equality with k step(zeros) calls."""
    from carle_amd.env import CARLE
    k = G.known()
    for storage in ("f32", "b1", "u8"):
        kw = dict(storage=storage, obs="f32") if storage != "f32" else {}
        env = CARLE(width=64, height=64, instances=2, device="cuda:0", **kw)
        env.reset()
        a = torch.zeros(2, 1, 64, 64)
        for r, c in k["glider_cells"]:
            a[0, 0, r, c] = 1.0
        a[1] = (torch.rand(64, 64, generator=torch.Generator().manual_seed(5)) < 0.3).float()
        env.step(a)
        ref = CARLE(width=64, height=64, instances=2, device="cuda:0")
        ref.reset()
        ref.universe = env.universe.clone()
        obs, reward, done, info = env.multistep(255)
        assert int(obs[0].sum().item()) == 5 and env.step_number == 256 and env.steps_since_action == 255
        assert float(reward.abs().sum()) == 0.0 and tuple(reward.shape) == (2, 1) and len(info) == 2
        z = torch.zeros(2, 1, 64, 64, device="cuda")
        for _ in range(255):
            ref.step(z)
        assert torch.equal(obs, ref.universe)
    env = CARLE(width=20, height=20, action_width=10, action_height=10, instances=1, device="cuda:0")
    env.reset()
    env.step(torch.ones(1, 1, 10, 10) * (torch.rand(1, 1, 10, 10) < 0.5))
    before = env.universe.clone()
    env.multistep(3)                                        # size without a packed kernel: loop
    ref = CARLE(width=20, height=20, action_width=10, action_height=10, instances=1, device="cuda:0")
    ref.reset(); ref.universe = before
    for _ in range(3):
        ref.step(torch.zeros(1, 1, 10, 10))
    assert torch.equal(env.universe, ref.universe) and env.step_number == 4


@pytest.mark.parametrize("n,size", [(4096, 512), (4101, 512), (16389, 256)])
def test_cfg5_full_size_with_action_against_the_oracle(n, size):
    """BASELINE configs[4] at its full size (B36/S23, N=4096, 512x512 bit-packed) WITH the 64x64
    action window, against oracle/ca_oracle.c on sampled instances (incl. the first / last instance
    and a workgroup boundary), three consecutive steps through CARLE(storage='b1').  N=4101 / 16389: batches that
    are not a multiple of 8 and need more than one round of the 8192 half-grid work units."""
    from carle_amd import hip
    from carle_amd.env import CARLE
    env = CARLE(width=size, height=size, instances=n, device="cuda:0", storage="b1")
    env.rules_from_string("B36/S23")
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(99)
    bits = torch.randint(-2 ** 63, 2 ** 63 - 1, (n, 1, size, size // 64), device="cuda", generator=g,
                         dtype=torch.int64)
    env.universe_bits = bits.clone()
    idx = [0, 1, 7, 8, 2047, 2048, 4094, 4095, n - 6, n - 2, n - 1]
    u_ref = hip.unpack_b1_f32(bits[idx].contiguous()).cpu().numpy()
    high = O.rule_mask([3, 6], [2, 3])
    for t in range(3):
        a = (torch.rand(n, 1, 64, 64, device="cuda", generator=g) <= 0.1).float()
        ap = hip.pack_f32_b1(a)
        obs = env.step(ap if t % 2 else a)[0]            # packed and fp32 actions
        u_ref, reset, _ = O.c_env_step(u_ref, a[idx].cpu().numpy(), high)
        assert not reset
        got = hip.unpack_b1_f32(obs[idx].contiguous()).cpu().numpy()
        assert np.array_equal(got, u_ref), f"grids differ at step {t}"
    assert env.step_number == 3 and env.steps_since_action == 0

"""GPU parity tests of the uint8 (CARLE_FMT_U8) path: step / toggle / pack / unpack / env surface,
bit-exact against the CPU oracle and the golden cases recorded from the genuine reference."""
import numpy as np
import pytest
import torch

import golden_util as G
from oracle import ca as O

pytestmark = pytest.mark.gpu


def _flags(act):
    """[any_nonzero, not_all_ones] the reference's two whole-batch predicates would give."""
    return [int(bool((act != 0).any())), int(bool((act != 1).any()))]


@pytest.mark.parametrize("n,size,asize,rule", [
    (1, 64, 64, ([3], [2, 3])), (5, 64, 64, ([3, 6], [2, 3])), (37, 64, 64, ([3, 6, 8], [2, 4, 5])),
    (7, 64, 32, ([3], [2, 3])), (3, 64, 48, ([1, 3, 5, 7], [0, 2, 4, 6, 8])), (2, 64, 62, ([3], [2, 3])),
    (3, 128, 64, ([3], [2, 3])), (2, 256, 64, ([3, 6], [2, 3])), (2, 256, 256, ([3], [2, 3])),
    (1, 512, 64, ([3], [2, 3])), (1, 512, 448, ([3, 6, 7, 8], [3, 4, 6, 7, 8])),
    (2, 20, 10, ([3], [2, 3])), (1, 2, 2, ([1], [1])), (3, 66, 64, ([2], [])),
])
def test_step_u8_against_oracle(n, size, asize, rule):
    """carle_ca_step_u8 called directly vs the C oracle: fp32 action, uint8 action, no action."""
    from carle_amd import hip
    rng = np.random.default_rng(n * 1000 + size + asize)
    mask = O.rule_mask(*rule)
    alive = rng.random((n, 1, size, size)) < 0.4
    u8 = np.where(alive, rng.choice(np.array([1, 2, 128, 255], np.uint8), alive.shape), 0).astype(np.uint8)
    tog = rng.random((n, 1, asize, asize)) < 0.2
    a32 = np.where(tog, rng.choice(np.array([1.0, 0.5, -1.0, np.inf], np.float32), tog.shape), 0).astype(np.float32)
    a32[0, 0, 0, 0] = -0.0
    a8 = np.where(tog, rng.choice(np.array([1, 3, 0x80], np.uint8), tog.shape), 0).astype(np.uint8)
    d_u = torch.from_numpy(u8).cuda()
    out = torch.empty_like(d_u)
    st = hip.new_status("cuda:0")
    u_f = alive.astype(np.float32)
    for act_np, act_d in ((a32, torch.from_numpy(a32).cuda()), (a8, torch.from_numpy(a8).cuda()), (None, None)):
        exp, _, _ = O.c_env_step(u_f, None if act_np is None else (act_np != 0).astype(np.float32), mask)
        out.fill_(7)
        st.zero_()
        hip.ca_step_u8(d_u, act_d, out, mask, st, 1)
        assert np.array_equal(out.cpu().numpy(), exp.astype(np.uint8))
        if act_np is not None:
            s = st.cpu().tolist()
            assert [s[1], s[3]] == _flags(act_np) and s[0] == 0 and s[2] == 0   # slot 1 only
    # exact predicates: all-ones and all-zero uint8 / fp32 actions
    for act, exp_flags in ((torch.ones(n, 1, asize, asize, dtype=torch.uint8), [1, 0]),
                           (torch.zeros(n, 1, asize, asize, dtype=torch.uint8), [0, 1]),
                           (torch.ones(n, 1, asize, asize), [1, 0]),
                           (torch.full((n, 1, asize, asize), 2, dtype=torch.uint8), [1, 1])):
        st.zero_()
        hip.ca_step_u8(d_u, act.cuda(), out, mask, st, 0)
        s = st.cpu().tolist()
        assert [s[0], s[2]] == exp_flags
    # several generations stay identical to the oracle
    cur_np, cur = u_f, d_u
    for t in range(5):
        nxt = torch.empty_like(cur)
        hip.ca_step_u8(cur, None, nxt, mask, None, 0)
        cur_np, _, _ = O.c_env_step(cur_np, None, mask)
        cur = nxt
    assert np.array_equal(cur.cpu().numpy(), cur_np.astype(np.uint8))


@pytest.mark.parametrize("n,size", [(1, 64), (3, 64), (130, 64), (2, 128), (2, 256), (1, 512)])
def test_pack_unpack_u8(n, size):
    from carle_amd import hip
    rng = np.random.default_rng(size + n)
    a = np.where(rng.random((n, 1, size, size)) < 0.5,
                 rng.choice(np.array([1, 7, 0x80, 0xFF], np.uint8), (n, 1, size, size)), 0).astype(np.uint8)
    st = hip.new_status("cuda:0")
    d = torch.from_numpy(a).cuda()
    packed = hip.pack_u8_b1(d, st, 1)
    assert torch.equal(packed, hip.pack_f32_b1((d != 0).float()))
    assert st.cpu().tolist()[:4] == [0, 1, 0, 1]
    back = hip.unpack_b1_u8(packed).cpu().numpy()
    assert np.array_equal(back, (a != 0).astype(np.uint8))
    st.zero_()
    hip.pack_u8_b1(torch.ones(n, 1, size, size, dtype=torch.uint8, device="cuda"), st, 0)
    assert st.cpu().tolist()[:4] == [1, 0, 0, 0]
    st.zero_()
    hip.pack_u8_b1(torch.zeros(n, 1, size, size, dtype=torch.uint8, device="cuda"), st, 0)
    assert st.cpu().tolist()[:4] == [0, 0, 1, 0]


def test_action_flags_entry_point():
    from carle_amd import hip
    st = hip.new_status("cuda:0")
    for t, exp in ((torch.zeros(3, 1, 70, 70), [0, 1]), (torch.ones(3, 1, 70, 70), [1, 0]),
                   (torch.full((1, 1, 5, 5), 0.5), [1, 1]), (torch.full((2, 1, 64, 64), -0.0), [0, 1])):
        for slot in (0, 1):
            st.zero_()
            hip.action_flags_f32(t.cuda(), st, slot)
            s = st.cpu().tolist()
            assert [s[slot], s[2 + slot]] == exp and s[1 - slot] == 0 and s[3 - slot] == 0
    one_off = torch.ones(5, 1, 64, 64)
    one_off[4, 0, 63, 63] = 0.0
    st.zero_()
    hip.action_flags_f32(one_off.cuda(), st, 0)
    assert st.cpu().tolist()[:4] == [1, 0, 1, 0]


@pytest.mark.parametrize("name", G.ca_case_names())
def test_u8_env_replays_golden_case(name):
    """CARLE(storage="u8") through the gym surface replays every golden case of the genuine
    reference: fp32 actions (CPU or device) and -- where the actions are 0/1 -- uint8 actions."""
    from carle_amd.env import CARLE
    c = G.ca_case(name)
    kinds = ["f32"] + (["u8"] if c.meta["actions_binary"] and c.meta["action_kind"] != "oversize" else [])
    for kind in kinds:
        env = CARLE(width=c.size, height=c.size, action_width=c.asize, action_height=c.asize,
                    instances=c.n, device="cuda:0", storage="u8")
        env.birth, env.survive = list(c.birth), list(c.survive)
        env.reset()
        env.universe = torch.from_numpy(c.init.copy()).cuda()      # fp32 in: converted by truthiness
        rec = 0
        for t in range(c.steps):
            a = torch.from_numpy(c.actions[t].copy())
            if kind == "u8":
                a = a.to(torch.uint8).cuda()
            elif t % 2:
                a = a.cuda()
            obs, reward, done, info = env.step(a)
            assert obs is env.universe and obs.is_cuda and obs.dtype == torch.uint8
            assert tuple(reward.shape) == (c.n, 1) and reward.is_cuda
            assert tuple(done.shape) == (c.n, 1) and not done.is_cuda and len(info) == c.n
            if (t + 1) in c.record:
                assert np.array_equal(obs.cpu().numpy(), c.obs[rec].astype(np.uint8)), \
                    f"grid mismatch at step {t + 1} ({kind} actions)"
                assert float(reward.abs().sum()) == 0.0
                assert env.step_number == c.step_number[t]
                assert env.steps_since_action == c.steps_since_action[t]
                rec += 1
        assert rec == len(c.record)


def test_u8_full_size_matches_f32_path():
    """BASELINE sizes: the uint8 step equals the fp32 step on the same cells (N=4096 64x64 with a
    full-window action; N=256 at 256x256 and 512x512 with the centred 64x64 window)."""
    from carle_amd import hip
    life = O.rule_mask([3], [2, 3])
    for n, size in ((4096, 64), (256, 256), (64, 512)):
        g = torch.Generator(device="cuda").manual_seed(size)
        u = (torch.rand(n, 1, size, size, device="cuda", generator=g) < 0.5)
        a = (torch.rand(n, 1, 64, 64, device="cuda", generator=g) < 0.1)
        o_f = torch.empty(n, 1, size, size, device="cuda")
        hip.ca_step_f32(u.float(), a.float(), o_f, life, None, 0)
        for act in (a.float(), a.to(torch.uint8)):
            o_8 = torch.empty(n, 1, size, size, dtype=torch.uint8, device="cuda")
            hip.ca_step_u8(u.to(torch.uint8), act, o_8, life, None, 0)
            assert torch.equal(o_8, o_f.to(torch.uint8))


def test_u8_env_surface_and_wrappers():
    """obs="f32" view, apply_action, RLE logging, master reset, and RND2D/AE2D bonuses on uint8
    observations equal to the fp32 environment's."""
    from carle_amd.env import CARLE
    from carle_amd.mcl import RND2D, AE2D
    torch.manual_seed(3)
    acts = [(torch.rand(6, 1, 64, 64) < 0.15).float() for _ in range(4)]
    rewards = {}
    for storage in ("f32", "u8"):
        torch.manual_seed(11)
        env = AE2D(RND2D(CARLE(width=64, height=64, instances=6, device="cuda:0", storage=storage)))
        env.eval()
        env.reset()
        rs = []
        for a in acts:
            obs, r, d, i = env.step(a)
            rs.append(r.detach().cpu().clone())
        rewards[storage] = (torch.stack(rs), obs.float().cpu())
    assert torch.equal(rewards["f32"][1], rewards["u8"][1])
    assert torch.allclose(rewards["f32"][0], rewards["u8"][0], atol=1e-6, rtol=0)
    assert float(rewards["f32"][0].abs().sum()) > 0

    env = CARLE(width=64, height=64, instances=2, device="cuda:0", storage="u8", obs="f32")
    obs = env.reset()
    assert obs.dtype == torch.float32 and env.universe.dtype == torch.uint8
    env.apply_action(acts[0][:2])
    assert torch.equal(env.universe.float().cpu(), acts[0][:2])
    obs = env.step(torch.zeros(2, 1, 64, 64))[0]
    assert obs.dtype == torch.float32 and torch.equal(obs, env.universe.float())
    rle = env.get_rle(env.universe[0, 0])
    assert rle.split("\n", 2)[2] == env.get_rle(env.universe[0, 0].float()).split("\n", 2)[2]
    obs = env.step(torch.ones(2, 1, 64, 64, dtype=torch.uint8, device="cuda"))[0]
    assert float(obs.sum()) == 0.0 and env.step_number == 0 and env.master_resets == 1

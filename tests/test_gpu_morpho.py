"""GPU parity tests of MorphoBonus (carle/mcl.py:107-195; carle_morpho_match_f32): against the trajectory recorded from
the genuine reference on the pattern files this package ships (tests/golden/morpho.npz; the reference's own default
files are missing from its tree), and against oracle/wrappers.py on random grids, patterns and sizes.  Tolerance of the
correlation: 1e-5 relative to the window's weight (sums of up to 64 fp32 products in another order than ATen's)."""
import os

import numpy as np
import pytest
import torch

import golden_util as G
from oracle import ca as O
from oracle import wrappers as OW

pytestmark = pytest.mark.gpu


def _build(n, extra_rle="", tmp_path=None):
    from carle_amd.env import CARLE
    from carle_amd.mcl import MorphoBonus
    torch.manual_seed(107)
    env = CARLE(width=64, height=64, instances=n, device="cuda:0")
    w = MorphoBonus(env)
    if extra_rle:
        p = os.path.join(str(tmp_path), "extra.rle")
        with open(p, "w") as f:
            f.write(extra_rle)
        w.add_rle_pattern(p)
    return env, w


@pytest.mark.parametrize("name", ["default", "extra", "cells14"])
def test_reference_trajectory(name, tmp_path):
    c = G.MorphoCase(name)
    env, w = _build(c.n, c.extra_rle, tmp_path)
    w.reward_scale = c.scale
    assert np.array_equal(w.target_patterns.cpu().numpy(), c.patterns)       # RLE parsing, padding, crop, weights, orientations
    assert (env.birth, env.survive) == (c.birth, c.survive)                  # read_rle sets the rule from the file, as there
    obs = w.reset()
    assert obs is env.universe
    assert np.array_equal(obs.cpu().numpy(), c.reset_obs)                    # the seed cells: same CPU generator, same draw
    for t in range(c.steps):
        obs, reward, done, info = w.step(torch.from_numpy(c.actions[t]))
        np.testing.assert_allclose(reward.cpu().numpy(), c.rewards[t], rtol=1e-6, atol=1e-5, err_msg=f"{name} step {t}")
    assert np.array_equal(obs.cpu().numpy(), c.final_obs)
    if name == "default":
        assert c.rewards[0][0, 0] >= 6.0                                     # the planted glider matched fully (max 15)


@pytest.mark.parametrize("h,w,n,k,binary", [(8, 8, 3, 1, True), (24, 40, 2, 12, True), (64, 64, 5, 12, True),
                                            (64, 64, 2, 32, False), (200, 72, 1, 6, False), (256, 256, 2, 12, True),
                                            (512, 512, 1, 3, True), (40, 512, 1, 2, False)])
def test_match_kernel_against_the_oracle(h, w, n, k, binary):
    from carle_amd import hip
    rng = np.random.default_rng(h * 7 + w + k)
    u = (rng.random((n, 1, h, w)) < 0.3).astype(np.float32) if binary else rng.standard_normal((n, 1, h, w)).astype(np.float32)
    a = (rng.random((n, 1, h, w)) < 0.1).astype(np.float32) if binary else rng.standard_normal((n, 1, h, w)).astype(np.float32)
    pats = rng.standard_normal((k, 1, 8, 8)).astype(np.float32)
    pats[0, 0][pats[0, 0] < 0] = -1.0
    reward = rng.standard_normal((n, 1)).astype(np.float32)
    r = torch.from_numpy(reward.copy()).cuda()
    out, _ = hip.morpho_match_f32(torch.from_numpy(u).cuda(), torch.from_numpy(a).cuda(), torch.from_numpy(pats).cuda(),
                                  reward=r, reward_scale=0.25)
    mx, mn = OW.MorphoBonus(pats).match(u, a)
    scale = np.abs(pats).sum(axis=(1, 2, 3)).max() * max(1.0, float(np.abs(u - a).max()))
    got = out.cpu().numpy()
    np.testing.assert_allclose(got[:, 0], mx, rtol=0, atol=1e-5 * scale)
    np.testing.assert_allclose(got[:, 1], mn, rtol=0, atol=1e-5 * scale)
    exp = reward[:, 0] + np.float32(0.25) * (got[:, 0] + got[:, 1])           # the kernel's own max / min: bit for bit
    assert np.array_equal(r.cpu().numpy()[:, 0], exp.astype(np.float32))


@pytest.mark.parametrize("storage", ["u8", "b1"])
def test_other_storages_step_like_fp32(storage):
    """The wrapper reads the universe as fp32 whatever the state's storage is."""
    from carle_amd.env import CARLE
    from carle_amd.mcl import MorphoBonus
    rng = np.random.default_rng(3)
    u0 = (rng.random((2, 1, 64, 64)) < 0.2).astype(np.float32)
    acts = [(rng.random((2, 1, 64, 64)) < 0.05).astype(np.float32) for _ in range(4)]
    runs = []
    for st in ("f32", storage):
        env = CARLE(width=64, height=64, instances=2, device="cuda:0", **({} if st == "f32" else {"storage": st, "obs": "f32"}))
        w = MorphoBonus(env)
        w.reset()
        env.universe = torch.from_numpy(u0).cuda()
        runs.append([w.step(torch.from_numpy(a).cuda())[1].cpu().numpy().copy() for a in acts])
    for x, y in zip(*runs):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("storage,obs_format", [("u8", "u8"), ("u8", "f32"), ("b1", "b1"), ("b1", "f32")])
def test_reset_seeds_the_state_the_kernels_read(storage, obs_format):
    """MorphoBonus.reset() on uint8 / bit-packed environments, in either observation format: the seed (mcl.py:
    189-190, the same CPU generator draw as on fp32 storage) must land in the STATE -- universe / universe_bits --, be what
    reset() returns, and evolve: one zero-action step equals the fp32 environment's."""
    from carle_amd.env import CARLE
    from carle_amd.mcl import MorphoBonus
    runs = []
    for st in ("f32", storage):
        kw = {} if st == "f32" else {"storage": st, "obs": obs_format}
        env = CARLE(width=64, height=64, instances=3, device="cuda:0", **kw)
        w = MorphoBonus(env)
        torch.manual_seed(77)
        obs = w.reset()
        state = env.universe.float().cpu().numpy().copy()                    # (b1: unpacked from universe_bits)
        assert state.sum() > 0 and set(np.unique(state)) <= {0.0, 1.0}
        if st == "b1":
            from carle_amd import hip
            assert np.array_equal(hip.unpack_b1_f32(env.universe_bits).cpu().numpy(), state)
        seen = obs if obs.dtype == torch.float32 else (env.universe if obs.dtype == torch.int64 else obs.float())
        assert np.array_equal(seen.float().cpu().numpy(), state)             # what reset() returned is that state
        o1, r1, _, _ = w.step(torch.zeros(3, 1, 64, 64, device="cuda"))
        runs.append((state, env.universe.float().cpu().numpy().copy(), r1.cpu().numpy().copy()))
    assert np.array_equal(runs[0][0], runs[1][0])                            # the same seed draw ...
    assert np.array_equal(runs[0][1], runs[1][1])                            # ... evolved by the CA on every storage
    assert np.array_equal(runs[0][2], runs[1][2])
    exp, _, _ = O.c_env_step(runs[0][0], np.zeros((3, 1, 64, 64), np.float32), O.rule_mask(env.birth, env.survive))
    assert np.array_equal(runs[1][1], exp)


def test_shape_and_argument_errors():
    from carle_amd import hip, _lib
    from carle_amd.env import CARLE
    from carle_amd.mcl import MorphoBonus
    w = MorphoBonus(CARLE(device="cuda:0"))                                  # 256 x 256 grid, 64 x 64 window
    w.reset()
    with pytest.raises(RuntimeError):                                        # mcl.py:172: universe - action does not broadcast
        w.step(torch.zeros(1, 1, 64, 64))
    x = torch.zeros(1, 1, 8, 1024, device="cuda")
    p = torch.zeros(1, 1, 8, 8, device="cuda")
    with pytest.raises(_lib.CarleHipError):
        hip.morpho_match_f32(x, x, p)                                        # wider than the kernel's LDS rows
    with pytest.raises(ValueError):
        hip.morpho_match_f32(x[..., :64].contiguous(), x[..., :32].contiguous(), p)
    assert _lib.load().carle_morpho_match_f32(None, None, None, 1, 1, 8, 8, None, None, None, 1.0, None) > 0


def test_full_size_action_at_the_default_geometry():
    """With an action of the universe's size the wrapper also works at 256 x 256 (the crop of env.py:164-167 centres it)."""
    from carle_amd.env import CARLE
    from carle_amd.mcl import MorphoBonus
    rng = np.random.default_rng(5)
    env = CARLE(device="cuda:0", instances=2)
    w = MorphoBonus(env)
    w.reset()
    u = (rng.random((2, 1, 256, 256)) < 0.2).astype(np.float32)
    env.universe = torch.from_numpy(u).cuda()
    a = (rng.random((2, 1, 256, 256)) < 0.05).astype(np.float32)
    obs, reward, done, info = w.step(torch.from_numpy(a).cuda())
    exp = OW.MorphoBonus(w.target_patterns.cpu().numpy()).step(u, np.zeros((2, 1), np.float32), a)
    np.testing.assert_allclose(reward.cpu().numpy(), exp, rtol=1e-6, atol=1e-4)
    off = (256 - 64) // 2
    un, _, _ = O.c_env_step(u, a[:, :, off:off + 64, off:off + 64], O.rule_mask(env.birth, env.survive))
    assert np.array_equal(obs.cpu().numpy(), un)

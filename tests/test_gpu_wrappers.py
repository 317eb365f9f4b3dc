"""GPU parity tests of the hand-coded reward wrappers and the harness loops against fixtures
recorded from the genuine reference (tests/golden/wrappers.npz)."""
import json
import os

import numpy as np
import pytest
import torch

import golden_util as G

pytestmark = pytest.mark.gpu


def _z():
    return np.load(os.path.join(G.GOLDEN, "wrappers.npz"))


def _replay(name, build, atol, post=None):
    from carle_amd.env import CARLE
    z = _z()
    m = json.loads(bytes(z[f"{name}/meta"]).decode())
    n, size, steps = m["n"], m["size"], m["steps"]
    torch.manual_seed(0)
    env = CARLE(width=size, height=size, instances=n, device="cuda:0")
    w = build(env)
    if post:
        post(w)
    w.eval()
    w.reset()
    env.universe = torch.from_numpy(G.unpack(z[f"{name}/init"], (n, 1, size, size))).cuda()
    acts = G.unpack(z[f"{name}/actions"], (steps, n, 1, 64, 64))
    for t in range(steps):
        obs, reward, done, info = w.step(torch.from_numpy(acts[t]))
        got = reward.detach().cpu().numpy().reshape(n, -1)
        np.testing.assert_allclose(got, z[f"{name}/rewards"][t], rtol=1e-6, atol=atol, err_msg=f"{name} step {t}")
    assert np.array_equal(obs.cpu().numpy(), G.unpack(z[f"{name}/final_obs"], (n, 1, size, size)))
    return w


def test_corner_bonus():
    from carle_amd.mcl import CornerBonus
    _replay("corner", lambda e: CornerBonus(e), atol=0)


def test_speed_detector():
    from carle_amd.mcl import SpeedDetector
    _replay("speed", lambda e: SpeedDetector(e), atol=1e-4)


def test_puffer_detector():
    from carle_amd.mcl import PufferDetector

    def small(w):
        w.growth_threshold = 6
    _replay("puffer", lambda e: PufferDetector(e), atol=0, post=small)


def test_parsimony_bonus():
    from carle_amd.mcl import RND2D, ParsimonyBonus
    z = _z()

    def build(e):
        w = RND2D(e)
        w.batch_size = 10 ** 6
        return ParsimonyBonus(w)
    from carle_amd.env import CARLE
    torch.manual_seed(0)
    env = CARLE(width=64, height=64, instances=1, device="cuda:0")
    w = build(env)
    w.load_state_dict({k: torch.from_numpy(v) for k, v in G.load_sd(z, "parsimony_sd").items()}, strict=True)
    w.eval(); w.reset()
    env.universe = torch.from_numpy(G.unpack(z["parsimony/init"], (1, 1, 64, 64))).cuda()
    acts = G.unpack(z["parsimony/actions"], (4, 1, 1, 64, 64))
    for t in range(4):
        obs, reward, done, info = w.step(torch.from_numpy(acts[t]))
        np.testing.assert_allclose(reward.cpu().numpy().reshape(1, -1), z["parsimony/rewards"][t], rtol=1e-5, atol=1e-6)


def test_eval_wrapper_stack():
    """RND2D -> AE2D -> SpeedDetector -> PufferDetector as stacked by evaluation/eval.py:83-87,
    with the reference's weights loaded strictly through the OUTERMOST wrapper."""
    from carle_amd.mcl import RND2D, AE2D, SpeedDetector, PufferDetector
    z = _z()

    def stack(e):
        w = AE2D(RND2D(e))
        w.batch_size = 10 ** 6
        w.env.batch_size = 10 ** 6
        w = SpeedDetector(w)
        w.reward_scale = 1e-2
        w = PufferDetector(w)
        w.reward_scale = 1e-3
        w.load_state_dict({k: torch.from_numpy(v) for k, v in G.load_sd(z, "evalstack_sd").items()}, strict=True)
        return w
    _replay("evalstack", stack, atol=2e-5)


def test_instance_sums_kernel():
    from carle_amd import hip
    x = (torch.rand(7, 1, 96, 96, device="cuda") < 0.3).float()
    w0 = torch.rand(1, 1, 96, 96, device="cuda")
    w1 = torch.arange(96 * 96, device="cuda", dtype=torch.float32).reshape(1, 1, 96, 96)
    got = hip.instance_sums_f32(x, w0, None, w1).cpu().numpy()
    xd = x.double()
    exp = torch.stack([xd.sum(dim=[1, 2, 3]), (xd * w0.double()).sum(dim=[1, 2, 3]),
                       torch.zeros(7, dtype=torch.float64, device="cuda"),
                       (xd * w1.double()).sum(dim=[1, 2, 3])], dim=1).cpu().numpy()
    np.testing.assert_allclose(got, exp, rtol=1e-6)


def test_train_and_evaluate_loops(tmp_path):
    """The harness loops of carle/train_mcl.py and evaluation/eval.py through the alias packages."""
    from carle.train_mcl import train
    from carle.agents import RandomAgent
    from carle.mcl import RND2D, AE2D, SpeedDetector, PufferDetector
    from evaluation.eval import evaluate
    from evaluation.submission import SubmissionAgent
    rewards = train(RandomAgent, instances=4, steps=[2, 6], rules=[[[3], [2, 3]], [[3, 6, 8], [2, 4, 5]]],
                    out_dir=str(tmp_path), verbose=False, width=64, height=64)
    assert len(rewards) == 2 * 2 * 6 and all(np.isfinite(r) for r in rewards) and max(rewards) > 0
    saved = sorted(os.listdir(tmp_path / "models"))
    assert any(s.startswith("AE2D_") for s in saved) and any(s.startswith("RND2D_") for s in saved)
    wrappers = [[RND2D, 1.0, str(tmp_path / "models" / [s for s in saved if s.startswith("RND2D_")][0])],
                [AE2D, 1.0, str(tmp_path / "models" / [s for s in saved if s.startswith("AE2D_")][0])],
                [SpeedDetector, 1e-2, None], [PufferDetector, 1e-3, None]]
    score, trace = evaluate(SubmissionAgent, [[[3], [2, 3]], [[2], [0]]], wrappers, steps=5, verbose=False,
                            width=64, height=64)
    assert len(trace) == 10 and np.isfinite(score) and score > 0

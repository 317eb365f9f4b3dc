"""GPU tests of the reward-statistics kernels and the single-process RunningRewardStats path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 7, 256, 4096, 100003])
def test_batch_moments(n):
    from carle_amd import hip
    x = torch.randn(n, device="cuda") * 3 + 1.5
    got = hip.batch_moments_f32(x).cpu().numpy()
    xd = x.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(got, [n, xd.mean(), ((xd - xd.mean()) ** 2).sum()], rtol=1e-9, atol=1e-9)
    acc = torch.zeros(3, dtype=torch.float64, device="cuda")
    x2 = torch.randn(n + 3, device="cuda") - 4.0
    hip.batch_moments_f32(x, acc=acc)
    hip.batch_moments_f32(x2, acc=acc)
    both = np.concatenate([xd, x2.cpu().numpy().astype(np.float64)])
    np.testing.assert_allclose(acc.cpu().numpy(), [both.size, both.mean(), ((both - both.mean()) ** 2).sum()],
                               rtol=1e-9)


def test_running_stats_and_normalize():
    from carle_amd import dist as cd
    from carle_amd import hip
    stats = cd.RunningRewardStats("cuda:0", sync_every=3)
    chunks = [torch.randn(512, 1, device="cuda") * (i + 1) for i in range(7)]
    for c in chunks:
        stats.update(c)
    stats.flush()
    allx = torch.cat(chunks).cpu().numpy().astype(np.float64).reshape(-1)
    np.testing.assert_allclose([stats.count, stats.mean, stats.var],
                               [allx.size, allx.mean(), allx.var()], rtol=1e-10)
    y = stats.normalize(chunks[0], eps=1e-8).cpu().numpy().astype(np.float64)
    exp = (chunks[0].cpu().numpy().astype(np.float64) - allx.mean()) / np.sqrt(allx.var() + 1e-8)
    np.testing.assert_allclose(y, exp, rtol=1e-5, atol=1e-6)
    assert y.shape == (512, 1)


@pytest.mark.parametrize("kind", ["f32", "u8", "b1", "rnd", "rnd_window32"])
def test_step_epilogue_merges_reward_moments(kind):
    """RunningRewardStats.attach: the step epilogue reduces the rewards itself (no moments launch);
    the statistics equal those of explicit update() calls on the same rewards, across reset()."""
    from carle_amd import dist as cd
    from carle_amd.env import CARLE
    from carle_amd.mcl import RND2D, AE2D
    torch.manual_seed(4)
    n, asize = 37, 32 if kind == "rnd_window32" else 64
    kw = dict(width=64, height=64, action_width=asize, action_height=asize, instances=n, device="cuda:0")
    env = CARLE(storage=kind, **kw) if kind in ("u8", "b1") else CARLE(**kw)
    wenv = RND2D(env, train_predictor=False) if kind.startswith("rnd") else env
    if wenv is not env:
        wenv.eval()
    attached = cd.RunningRewardStats("cuda:0", sync_every=3)
    explicit = cd.RunningRewardStats("cuda:0", sync_every=3)
    assert attached.attach(wenv)
    wenv.reset()
    g = torch.Generator().manual_seed(2)
    rewards = []
    for t in range(8):
        if t == 5:
            wenv.reset()                                  # the pointer slots survive a reset
        a = (torch.rand(n, 1, asize, asize, generator=g) < 0.2).float()
        if t == 3:
            a.fill_(1.0)                                  # master reset: dead-grid bonus for everyone
        obs, reward, done, info = wenv.step(a)
        assert wenv._moments_merged
        attached.after_step(wenv, reward)
        explicit.update(reward)
        rewards.append(reward.detach().clone())
    attached.flush(); explicit.flush()
    allr = torch.cat(rewards).double().cpu().numpy().reshape(-1)
    np.testing.assert_allclose([attached.count, attached.mean, attached.var],
                               [allr.size, allr.mean(), allr.var()], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(attached.state.cpu().numpy(), explicit.state.cpu().numpy(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(attached.last_batch.cpu().numpy(), explicit.last_batch.cpu().numpy(),
                               rtol=1e-9, atol=1e-12)
    if kind.startswith("rnd"):
        assert float(allr.sum()) > 0
    # a stacked wrapper adds its bonus after the epilogue: attach refuses, after_step falls back
    stack = AE2D(RND2D(CARLE(**kw), train_predictor=False), train_predictor=False)
    other = cd.RunningRewardStats("cuda:0")
    assert not other.attach(stack)
    stack.eval(); stack.reset()
    obs, reward, done, info = stack.step(torch.zeros(n, 1, asize, asize))
    other.after_step(stack, reward)
    np.testing.assert_allclose(other.mean, float(reward.double().mean()), rtol=1e-9)


def test_attached_stats_with_fused_random_agent():
    """The in-kernel random-toggle step merges the moments too; a wrapper the epilogue cannot cover
    (here: logging, which falls back to the materialised action) still counts every step once."""
    from carle_amd import dist as cd
    from carle_amd.env import CARLE
    from carle_amd.agents import RandomAgent
    env = CARLE(width=64, height=64, instances=5, device="cuda:0")
    stats = cd.RunningRewardStats("cuda:0")
    assert stats.attach(env)
    env.reset()
    agent = RandomAgent(action_width=64, action_height=64, fused=True, seed=3)
    for t in range(4):
        obs, reward, done, info = env.step(agent(env.universe))
        assert env._moments_merged
        stats.after_step(env, reward)
    stats.flush()
    assert stats.count == 20.0 and stats.mean == 0.0

"""GPU tests of the device-side agents: the Philox Bernoulli kernel bit-exact against
oracle/philox.py (pinned by the Random123 known answers), the fused random-toggle step against
the materialised action, and the toggle rate of the reference's RandomAgent (agents.py:33,38-40)."""
import numpy as np
import pytest
import torch

from oracle import ca as O
from oracle.philox import bernoulli_f32 as np_bernoulli

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("count,p,seed,offset", [(4, 0.1, 0, 0), (4096, 0.1, 1, 7), (65536 + 8, 0.5, 0xDEADBEEFCAFE, 2 ** 40 + 3),
                                                  (1024, 0.0, 5, 5), (1024, 1.0, 5, 5)])
def test_bernoulli_kernel_matches_oracle(count, p, seed, offset):
    from carle_amd import hip
    got = hip.bernoulli_f32((count,), p, seed, offset, "cuda:0").cpu().numpy()
    assert np.array_equal(got, np_bernoulli(count, p, seed, offset))


def test_random_agent_surface_and_rate():
    from carle_amd.agents import RandomAgent, SubmissionAgent, DeviceRandomAction
    agent = RandomAgent(action_width=64, action_height=64, seed=3)
    obs = torch.zeros(128, 1, 64, 64, device="cuda")
    a0, a1 = agent(obs), agent(obs)
    assert tuple(a0.shape) == (128, 1, 64, 64) and a0.is_cuda and a0.dtype == torch.float32
    assert set(torch.unique(a0).cpu().tolist()) <= {0.0, 1.0}
    assert not torch.equal(a0, a1)                         # fresh draw per call
    rate = float(torch.cat([a0, a1]).mean())
    assert abs(rate - 0.1) < 0.002                         # toggle_rate = 0.1 (agents.py:33)
    tok = RandomAgent(fused=True, seed=3)(obs)
    assert isinstance(tok, DeviceRandomAction) and torch.equal(tok.materialize(), a0)
    assert tuple(SubmissionAgent()(obs).shape) == (128, 1, 64, 64)


@pytest.mark.parametrize("n,asize", [(1, 64), (37, 64), (5, 32)])
def test_fused_random_step_equals_materialised(n, asize):
    """step(DeviceRandomAction) == step(token.materialize()) bit for bit, and == the oracle."""
    from carle_amd.agents import RandomAgent
    from carle_amd.env import CARLE
    envs = [CARLE(width=64, height=64, action_width=asize, action_height=asize, instances=n, device="cuda:0")
            for _ in range(2)]
    u0 = (torch.rand(n, 1, 64, 64) < 0.4).float()
    for e in envs:
        e.rules_from_string("B36/S23")
        e.reset()
        e.universe = u0.clone().cuda()
    fused = RandomAgent(action_width=asize, action_height=asize, fused=True, seed=11)
    plain = RandomAgent(action_width=asize, action_height=asize, fused=False, seed=11)
    u_np = u0.numpy()
    mask = O.rule_mask([3, 6], [2, 3])
    for t in range(5):
        o0 = envs[0].step(fused(envs[0].universe))[0]
        a = plain(envs[1].universe)
        o1 = envs[1].step(a)[0]
        u_np, _, _ = O.c_env_step(u_np, a.cpu().numpy(), mask)
        assert torch.equal(o0, o1)
        assert np.array_equal(o0.cpu().numpy(), u_np)
    assert envs[0].step_number == envs[1].step_number == 5
    assert envs[0].steps_since_action == envs[1].steps_since_action


@pytest.mark.parametrize("size,asize", [(256, 64), (512, 64), (256, 128), (128, 64), (66, 64)])
def test_fused_token_on_other_geometries(size, asize):
    """256 / 512 grids draw the toggles inside the wide step kernel, every other geometry falls back
    to the materialised action: either way the step equals the oracle on the Philox stream of
    oracle/philox.py, for several steps (the call counter advances the offset)."""
    from carle_amd.agents import RandomAgent
    from carle_amd.env import CARLE
    n = 3
    env = CARLE(width=size, height=size, action_width=asize, action_height=asize, instances=n, device="cuda:0")
    env.reset()
    agent = RandomAgent(action_width=asize, action_height=asize, fused=True, seed=1)
    cur = np.zeros((n, 1, size, size), np.float32)
    mask = O.rule_mask([3], [2, 3])
    for t in range(4):
        obs = env.step(agent(env.universe))[0]
        cur, _, _ = O.c_env_step(cur, np_bernoulli(n * asize * asize, 0.1, 1, t).reshape(n, 1, asize, asize), mask)
        assert np.array_equal(obs.cpu().numpy(), cur), f"step {t}"
    assert env.step_number == 4


def test_random_network_agent():
    from carle_amd.agents import RandomNetworkAgent
    torch.manual_seed(0)
    agent = RandomNetworkAgent(observation_width=64, observation_height=64)
    a = agent((torch.rand(3, 1, 64, 64, device="cuda") < 0.3).float())
    assert tuple(a.shape) == (3, 1, 64, 64) and a.is_cuda
    assert set(torch.unique(a).cpu().tolist()) <= {0.0, 1.0}


@pytest.mark.parametrize("storage,n,size", [("u8", 3, 64), ("u8", 37, 64), ("u8", 2, 256), ("u8", 1, 512),
                                             ("b1", 3, 64), ("b1", 70, 64), ("b1", 2, 128), ("b1", 3, 256),
                                             ("b1", 2, 512)])
def test_fused_random_step_on_u8_and_packed_state(storage, n, size, monkeypatch):
    """RandomAgent(fused=True) on uint8 / bit-packed state: the toggles are drawn inside the step
    kernel (carle_env_step_u8_rand / carle_env_step_b1_rand), bit-identical to the materialised
    action through the same storage and to the oracle; counters and flags included."""
    from carle_amd.agents import RandomAgent
    from carle_amd.env import CARLE
    from carle_amd import hip
    calls = []
    for name in ("env_step_u8_rand", "env_step_b1_rand"):
        orig = getattr(hip, name)
        monkeypatch.setattr(hip, name, (lambda o, nm: lambda *a, **k: (calls.append(nm), o(*a, **k))[1])(orig, name))
    if True:
        envs = [CARLE(width=size, height=size, instances=n, device="cuda:0", storage=storage, obs="f32")
                for _ in range(2)]
        u0 = (torch.rand(n, 1, size, size) < 0.4).float()
        for e in envs:
            e.rules_from_string("B36/S23")
            e.reset()
            e.universe = u0.clone().cuda()
        fused = RandomAgent(action_width=64, action_height=64, fused=True, seed=5)
        plain = RandomAgent(action_width=64, action_height=64, fused=False, seed=5)
        u_np = u0.numpy()
        mask = O.rule_mask([3, 6], [2, 3])
        for t in range(4):
            o0 = envs[0].step(fused(envs[0].universe))[0]
            a = plain(envs[1].universe)
            o1 = envs[1].step(a)[0]
            u_np, _, _ = O.c_env_step(u_np, a.cpu().numpy(), mask)
            assert torch.equal(o0, o1)
            assert np.array_equal(o0.cpu().numpy(), u_np)
        assert envs[0].step_number == envs[1].step_number == 4
        assert envs[0].steps_since_action == envs[1].steps_since_action == 0
        assert len(calls) == 4 and set(calls) == {"env_step_%s_rand" % storage}     # the fused entry really ran

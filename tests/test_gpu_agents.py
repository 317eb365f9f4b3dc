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


def _agent_check(agent, obs, scale=None):
    """HIP forward of a RandomNetworkAgent against the oracle (oracle/nets.py, pinned to the reference's
    fixture): sigmoid outputs within 1e-5, actions identical wherever the output is not within that tolerance
    of the threshold (the comparison itself is exact: fp32 output <= fp32(0.1), agents.py:98).  With the
    linear layer scaled by 40 the logits -- and the fp32 summation-order noise of their up to 4096-term dot
    products -- are 40 times larger: 1e-4 there."""
    from oracle import nets as ON
    if scale is not None:
        with torch.no_grad():
            agent.network[7].weight.mul_(scale)
    action, output = agent(obs, want_output=True)
    sd = {"network." + k: v.detach().cpu().numpy() for k, v in agent.network.state_dict().items()}
    with torch.no_grad():
        ref_out = ON.agent_output(obs.cpu().numpy(), sd).numpy()
    ref_act = (ref_out <= np.float32(agent.toggle_rate)).astype(np.float32)
    n = obs.shape[0]
    assert tuple(action.shape) == (n, 1, agent.action_width, agent.action_height) and action.is_cuda
    got_out = output.cpu().numpy()
    tol = 1e-5 if scale is None else 1e-4
    np.testing.assert_allclose(got_out, ref_out, rtol=0, atol=tol)
    got_act = action.reshape(n, -1).cpu().numpy()
    clear = np.abs(ref_out - np.float32(agent.toggle_rate)) > tol
    assert np.array_equal(got_act[clear], ref_act[clear])
    assert set(np.unique(got_act).tolist()) <= {0.0, 1.0}
    assert np.array_equal(agent(obs).cpu().numpy(), action.cpu().numpy())        # without the outputs
    return got_act.mean()


@pytest.mark.parametrize("n,h,w,aw", [(3, 64, 64, 64), (70, 16, 256, 8), (1, 8, 8, 8), (5, 128, 64, 16),
                                      (16, 256, 256, 64), (130, 64, 64, 64), (1029, 64, 64, 64)])
def test_random_network_agent_matches_oracle(n, h, w, aw):
    """carle_agent_rn_forward_f32 (tiled convolutions + the linear layer on the fp32 matrix cores) vs the
    oracle; square, non-square, one-tile, partial-tile and multi-tile geometries; default-init weights and
    a scaled linear layer (so that both sides of the threshold occur).  N=1029: the 128x128-tile kernel of large
    batches (ragged last tile); the others the 64x64-tile kernel with and without the K split."""
    from carle_amd.agents import RandomNetworkAgent
    torch.manual_seed(100 + n)
    agent = RandomNetworkAgent(observation_width=w, observation_height=h, action_width=aw, action_height=aw)
    dens = torch.linspace(0.02, 0.8, n, device="cuda").reshape(n, 1, 1, 1)
    obs = (torch.rand(n, 1, h, w, device="cuda") < dens).float()
    _agent_check(agent, obs)
    rate = _agent_check(agent, obs, scale=40.0)
    if rate < 0.01:                                         # few inputs per output: scale further
        rate = _agent_check(agent, obs, scale=8.0)
    if n * aw * aw >= 4096:
        assert 0.001 < rate < 0.999
    # non-binary observations go through the same arithmetic (the reference feeds the tensor itself)
    _agent_check(agent, torch.randn(n, 1, h, w, device="cuda"))


def test_random_network_agent_reference_fixture():
    """The fixture recorded from the reference's RandomNetworkAgent (tools/make_golden.py: observations
    (6,1,16,256), action 8x8, default-init and scaled linear layer)."""
    import os
    import golden_util as G
    from carle_amd.agents import RandomNetworkAgent
    z = np.load(os.path.join(G.GOLDEN, "agent_net.npz"))
    obs = torch.from_numpy(G.unpack(z["obs"], tuple(z["obs_shape"]))).cuda()
    agent = RandomNetworkAgent(observation_width=256, observation_height=16, action_width=8, action_height=8)
    sd = {k: torch.from_numpy(v) for k, v in G.load_sd(z, "sd").items()}
    agent.load_state_dict(sd, strict=True)
    for lin, tag in ((None, "default"), (z["lin_scaled"], "scaled")):
        if lin is not None:
            with torch.no_grad():
                agent.network[7].weight.copy_(torch.from_numpy(lin))
        action, output = agent(obs, want_output=True)
        ref_out, ref_act = z[f"output_{tag}"], z[f"action_{tag}"]
        np.testing.assert_allclose(output.cpu().numpy(), ref_out, rtol=0, atol=1e-5)
        clear = (np.abs(ref_out - z["toggle_rate"]) > 1e-5).reshape(ref_act.shape)
        assert np.array_equal(action.cpu().numpy()[clear], ref_act[clear])
        assert clear.mean() > 0.99


def test_random_network_agent_drives_the_environment():
    """agents.py:91-102 in the loop of train_mcl.py:64-69: the agent's action steps a 256x256 environment."""
    from carle_amd.agents import RandomNetworkAgent
    from carle_amd.env import CARLE
    torch.manual_seed(1)
    env = CARLE(width=256, height=256, instances=4, device="cuda:0")
    agent = RandomNetworkAgent()
    with torch.no_grad():
        agent.network[7].weight.mul_(40.0)
    obs = env.reset()
    env.universe = (torch.rand(4, 1, 256, 256, device="cuda") < 0.3).float()
    obs = env.universe
    for _ in range(3):
        action = agent(obs)
        assert tuple(action.shape) == (4, 1, 64, 64)
        obs, reward, done, info = env.step(action)
    assert env.step_number == 3


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


def test_random_network_agent_one_threshold_and_geometry_domain():
    """The same observation gives the same action with and without the sigmoid values requested (one threshold form in
    every kernel path: split-K, 64- and 128-tile), and a geometry the reference's module accepts but the kernels do not
    (100x100 -> 625 inputs) raises: there is no torch evaluation behind the agent."""
    from carle_amd.agents import RandomNetworkAgent
    for n, hw in ((3, 64), (130, 64), (1029, 64), (2, 256)):
        torch.manual_seed(n)
        agent = RandomNetworkAgent(observation_width=hw, observation_height=hw, action_width=16, action_height=16)
        with torch.no_grad():
            agent.network[7].weight.mul_(30.0)
        obs = (torch.rand(n, 1, hw, hw, device="cuda") < 0.3).float()
        a0 = agent(obs)
        a1, out = agent(obs, want_output=True)
        assert torch.equal(a0, a1)
        far = (out - agent.toggle_rate).abs() > 1e-5
        assert torch.equal(a1.reshape(n, -1)[far], (out <= agent.toggle_rate).float()[far])
    torch.manual_seed(5)
    agent = RandomNetworkAgent(observation_width=100, observation_height=100, action_width=8, action_height=8)
    obs = (torch.rand(2, 1, 100, 100, device="cuda") < 0.3).float()
    with pytest.raises(NotImplementedError, match="outside the HIP kernels' domain"):
        agent(obs)                                           # no torch evaluation in the kernels' place

"""oracle/nets.py (fp32 CPU restatement of RND2D/AE2D forward) against the golden fixtures."""
import numpy as np
import pytest
import torch

import golden_util as G
from oracle import nets as ON


@pytest.mark.parametrize("size,n", [(64, 16), (256, 4)])
def test_nets_oracle_matches_reference(size, n):
    z = G.nets(size)
    grids = G.unpack(z["grids"], (n, 1, size, size))
    rnd_sd = G.load_sd(z, "rnd_sd")
    ae_sd = G.load_sd(z, "ae_sd")
    with torch.no_grad():
        rb = ON.rnd_bonus(grids, rnd_sd).numpy()
        ab = ON.ae_bonus(grids, ae_sd).numpy()
    np.testing.assert_allclose(rb, z["rnd_bonus"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(ab, z["ae_bonus"], rtol=0, atol=1e-6)


def test_nets_oracle_intermediates_64():
    z = G.nets(64)
    grids = G.unpack(z["grids"], (16, 1, 64, 64))
    rnd_sd = G.load_sd(z, "rnd_sd")
    ae_sd = G.load_sd(z, "ae_sd")
    with torch.no_grad():
        np.testing.assert_allclose(ON.rnd_random(grids, rnd_sd).numpy(), z["rnd_target"], atol=1e-6)
        np.testing.assert_allclose(ON.rnd_predictor(grids, rnd_sd).numpy(), z["rnd_pred"], atol=1e-6)
        np.testing.assert_allclose(ON.ae_predictor(grids, ae_sd).numpy(), z["ae_pred"], atol=1e-6)


def test_state_dict_key_layout():
    """The key layout the build's modules must load strictly."""
    z = G.nets(256)
    rk = list(G.load_sd(z, "rnd_sd").keys())
    assert rk[:2] == ["inner_env.neighborhood.weight", "env.neighborhood.weight"]
    assert [k for k in rk if k.startswith("predictor.")] == [
        "predictor.0.weight", "predictor.0.bias", "predictor.5.weight", "predictor.5.bias",
        "predictor.11.weight", "predictor.11.bias"]
    assert [k for k in rk if k.startswith("random_network.")] == [
        "random_network.0.weight", "random_network.0.bias", "random_network.4.weight",
        "random_network.4.bias", "random_network.8.weight", "random_network.8.bias"]
    ak = list(G.load_sd(z, "ae_sd").keys())
    assert len(ak) == 23 and "env.random_network.8.bias" in ak and "predictor.11.bias" in ak


def test_agent_oracle_matches_reference():
    """oracle.nets.agent_output / agent_action reproduce the RandomNetworkAgent fixture recorded from the
    reference (carle/agents.py:45-102), default-init and scaled linear layer."""
    import os
    z = np.load(os.path.join(G.GOLDEN, "agent_net.npz"))
    obs = G.unpack(z["obs"], tuple(z["obs_shape"]))
    sd = G.load_sd(z, "sd")
    assert list(sd.keys()) == ["network.0.weight", "network.3.weight", "network.7.weight"]
    with torch.no_grad():
        np.testing.assert_allclose(ON.agent_output(obs, sd).numpy(), z["output_default"], rtol=0, atol=1e-6)
        np.testing.assert_array_equal(ON.agent_action(obs, sd, float(z["toggle_rate"])).numpy(), z["action_default"].reshape(6, -1))
        sd["network.7.weight"] = z["lin_scaled"]
        np.testing.assert_allclose(ON.agent_output(obs, sd).numpy(), z["output_scaled"], rtol=0, atol=1e-6)
        np.testing.assert_array_equal(ON.agent_action(obs, sd, float(z["toggle_rate"])).numpy(), z["action_scaled"].reshape(6, -1))


# ---- PredictionBonus / SurpriseBonus (carle/mcl.py:558-728): oracle/prediction.py against the reference's trajectories
@pytest.mark.parametrize("name", G.PREDICTION_CASES)
def test_prediction_oracle_reproduces_the_reference_trajectories(name):
    from oracle import ca as O
    from oracle import prediction as OP
    c = G.prediction_case(name)
    w = getattr(OP, c.cls)(c.init_sd, batch_size=c.batch_size, reward_scale=c.scale)
    life = O.rule_mask([3], [2, 3])
    u = c.init
    for t in range(c.steps):
        u, reset, _ = O.c_env_step(u, c.actions[t], life)
        reward = w.step(u, np.zeros((c.n, 1), np.float32))
        np.testing.assert_allclose(reward, c.rewards[t], rtol=2e-5, atol=2e-6, err_msg=f"{name} step {t}")
        assert w.live_cells == c.live_cells[t]
    assert np.array_equal(u, c.final_obs)
    got = w.state()
    for k, v in c.final_sd.items():
        if k.startswith("predictor."):
            np.testing.assert_allclose(got[k], v, rtol=2e-4, atol=2e-5, err_msg=f"{name} {k}")


def test_prediction_oracle_explicit_calls():
    """get_bonus (frames 0-2) then get_bonus_update (frames 3-6: one Adam step each), mcl.py:581-605, 644-656."""
    from oracle import prediction as OP
    z = G.prediction_file()
    frames = G.unpack(z["calls/frames"], (7, 2, 1, 64, 64))
    w = OP.PredictionBonus(G.load_sd(z, "calls/init_sd"))
    for t in range(7):
        r = w.get_bonus(frames[t]) if t < 3 else w.get_bonus_update(frames[t])
        np.testing.assert_allclose(r.detach().numpy(), z["calls/bonus"][t], rtol=2e-5, atol=2e-6, err_msg=f"call {t}")
    got, want = w.state(), G.load_sd(z, "calls/final_sd")
    for k, v in want.items():
        if k.startswith("predictor."):
            np.testing.assert_allclose(got[k], v, rtol=2e-4, atol=2e-5, err_msg=k)

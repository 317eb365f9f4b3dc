"""GPU parity tests of PredictionBonus / SurpriseBonus (carle/mcl.py:558-728): the HIP path (the per-layer AE2D kernels
with a target frame: carle_ae2d_predict_bonus_f32 / carle_ae2d_predict_train_f32; AE2D's own kernels for SurpriseBonus;
Adam through carle_adam_update_f32) against trajectories recorded from the genuine reference
(tests/golden/prediction.npz), against oracle/prediction.py on random cases, and the reference's own two wrapper tests
(tests/test_mcl.py:17-53, 63-100) restated."""
import numpy as np
import pytest
import torch

import golden_util as G
from oracle import ca as O_ca
from oracle import nets as ON
from oracle import prediction as OP

pytestmark = pytest.mark.gpu
# measured against the reference-recorded trajectories (fp32 autograd + torch Adam on the CPU there;
# profiles/r04_trajectory_errors.txt): rewards <= 3e-8 at 64x64 and <= 6.3e-7 in the 256x256 glider scenario, weights <= 9e-7
# but for one bias at 5.2e-6 (4 Adam steps at lr 6e-2 on an almost empty grid)
REWARD_TOL = 2e-6
WEIGHT_TOL = 2e-5


def _build(cls_name, size, n, sd=None, batch_size=64, scale=1.0, **kw):
    from carle_amd.env import CARLE
    import carle_amd.mcl as M
    if size < 64:
        kw = dict(kw, action_width=size, action_height=size)
    env = CARLE(width=size, height=size, instances=n, device="cuda:0", **kw)
    w = getattr(M, cls_name)(env)
    if sd is not None:
        w.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    w.eval()
    w.batch_size = batch_size
    w.reward_scale = scale
    w.reset()
    return env, w


@pytest.mark.parametrize("name", G.PREDICTION_CASES)
def test_reference_trajectories(name):
    c = G.prediction_case(name)
    env, w = _build(c.cls, c.size, c.n, c.init_sd, c.batch_size, c.scale)
    env.universe = torch.from_numpy(c.init).cuda()
    worst = 0.0
    for t in range(c.steps):
        obs, reward, done, info = w.step(torch.from_numpy(c.actions[t]))
        got = reward.cpu().numpy()
        worst = max(worst, float(np.abs(got - c.rewards[t]).max()))
        np.testing.assert_allclose(got, c.rewards[t], rtol=0, atol=REWARD_TOL, err_msg=f"{name} step {t}")
        assert float(w.live_cells) == c.live_cells[t]
    assert np.array_equal(obs.cpu().numpy(), c.final_obs)
    for k, v in w.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), c.final_sd[k], rtol=0, atol=WEIGHT_TOL, err_msg=f"{name} {k}")
    if c.cls == "PredictionBonus":
        assert len(w.grid_buffer) == min(c.steps, 5)
    # the empty instance of the 64x64 cases was rewarded exactly zero at every step (mcl.py:687-691)
    if c.n > 1:
        assert (c.rewards[:, -1] == 0).all()


def test_reference_explicit_calls():
    """get_bonus x 3 (no gradient), then get_bonus_update x 4 (one Adam step each) on recorded frames."""
    z = G.prediction_file()
    frames = G.unpack(z["calls/frames"], (7, 2, 1, 64, 64))
    env, w = _build("PredictionBonus", 64, 2, G.load_sd(z, "calls/init_sd"))
    for t in range(7):
        f = torch.from_numpy(frames[t]).cuda()
        r = w.get_bonus(f) if t < 3 else w.get_bonus_update(f)
        np.testing.assert_allclose(r.cpu().numpy(), z["calls/bonus"][t], rtol=0, atol=REWARD_TOL, err_msg=f"call {t}")
    want = G.load_sd(z, "calls/final_sd")
    for k, v in w.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), want[k], rtol=0, atol=WEIGHT_TOL, err_msg=k)


@pytest.mark.parametrize("size,n,mode", [(8, 5, "binary"), (24, 3, "real"), (72, 2, "binary"), (136, 1, "binary"),
                                         (256, 2, "binary"), (64, 5, "packed"), (64, 300, "packed"), (128, 3, "packed"),
                                         (256, 2, "packed"), (256, 40, "packed"), (512, 1, "packed")])
def test_predict_kernels_against_the_oracle(size, n, mode):
    """carle_ae2d_predict_bonus_* / _train_* on random frame pairs -- fp32 frames through the per-layer kernels, bit-packed
    frames through AE2D's fused 64x64 / tiled kernels with a target: bonus within 1e-5 of oracle/nets.py, the gradient of
    mean_n(loss) within 3e-6 of the tensor's largest component of fp64 autograd (the bound of the AE2D gradient kernels,
    DESIGN section 6)."""
    from carle_amd import hip
    rng = np.random.default_rng(size * 31 + n)
    torch.manual_seed(size)
    env, w = _build("PredictionBonus", size, min(n, 4))
    sd = {k: v.detach().cpu().numpy() for k, v in w.state_dict().items()}
    mk = (lambda: rng.standard_normal((n, 1, size, size)).astype(np.float32)) if mode == "real" else \
        (lambda: (rng.random((n, 1, size, size)) < 0.3).astype(np.float32))
    obs, target = mk(), mk()
    pb = w._pred_blob.get(torch.device("cuda:0"))
    o, t = torch.from_numpy(obs).cuda(), torch.from_numpy(target).cuda()
    if mode == "packed":
        o, t = hip.pack_f32_b1(o), hip.pack_f32_b1(t)
    bonus, recon, ws, _ = hip.ae2d_predict(o, t, pb, want_recon=True)
    with torch.no_grad():
        pred = ON.ae_predictor(torch.from_numpy(obs), sd)
        exp = torch.mean((torch.from_numpy(target) - pred) ** 2, dim=[1, 2, 3]).numpy()
    np.testing.assert_allclose(bonus.cpu().numpy(), exp, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(recon.cpu().numpy(), pred.numpy(), rtol=0, atol=1e-5)
    grad = torch.zeros(pb.numel(), device="cuda")
    gs = 2.0 / (size * size * n)
    bonus2, _, ws, _ = hip.ae2d_predict(o, t, pb, ws, None, grad, gs)
    np.testing.assert_allclose(bonus2.cpu().numpy(), bonus.cpu().numpy(), rtol=0, atol=0 if mode != "packed" else 2e-7)
    if mode == "packed":                                         # and once more on the tables the first call left
        g2 = torch.zeros_like(grad)
        bonus3, _, _, _ = hip.ae2d_predict(o, t, pb, ws, None, g2, gs, tables_valid=True)
        assert torch.equal(bonus3, bonus2) and torch.equal(g2, grad)
    import copy
    ref_net = copy.deepcopy(w.predictor).cpu().double().eval()           # fp64 autograd through the torch modules
    p64 = ref_net(torch.from_numpy(obs).double()).reshape(obs.shape)
    loss = torch.mean(torch.mean((torch.from_numpy(target).double() - p64) ** 2, dim=[1, 2, 3]))
    loss.backward()
    off, g = 0, grad.cpu().numpy().astype(np.float64)
    for k, prm in ref_net.named_parameters():
        ref = prm.grad.numpy().reshape(-1)
        mine = g[off:off + ref.size]
        off += ref.size
        assert np.abs(mine - ref).max() <= 3e-6 * max(np.abs(ref).max(), 1e-30) + 1e-12, k
    assert off == g.size


@pytest.mark.parametrize("cls,size,n,storage", [("PredictionBonus", 64, 4, "f32"), ("PredictionBonus", 128, 2, "u8"),
                                                ("PredictionBonus", 64, 3, "b1"), ("SurpriseBonus", 64, 5, "f32"),
                                                ("SurpriseBonus", 256, 2, "f32"), ("SurpriseBonus", 128, 2, "b1"),
                                                ("PredictionBonus", 40, 2, "f32"), ("SurpriseBonus", 24, 3, "f32")])
def test_wrappers_against_the_oracle(cls, size, n, storage):
    """Stepping on the environment with predictor updates (batch_size 3, 11 steps, one instance empty throughout,
    a master reset in between): rewards against oracle/prediction.py + the CA oracle, all storages."""
    rng = np.random.default_rng(size + 7 * n)
    torch.manual_seed(size + n)
    kw = {} if storage == "f32" else {"storage": storage, "obs": "f32"}
    env, w = _build(cls, size, n, None, 3, 0.5, **kw)
    asz = min(size, 64)
    sd = {k: v.detach().cpu().numpy().copy() for k, v in w.state_dict().items()}
    ow = getattr(OP, cls)(sd, batch_size=3, reward_scale=0.5)
    u = (rng.random((n, 1, size, size)) < 0.3).astype(np.float32)
    u[-1] = 0
    env.universe = torch.from_numpy(u).cuda()
    life = O_ca.rule_mask([3], [2, 3])
    worst = 0.0
    for t in range(11):
        a = (rng.random((n, 1, asz, asz)) < (0.1 if t % 2 else 0.0)).astype(np.float32)
        a[-1] = 0
        if t == 6:
            a[:] = 1.0                                         # master reset: every grid empty, every reward zero
        obs, reward, done, info = w.step(torch.from_numpy(a))
        u, reset, _ = O_ca.c_env_step(u, a, life)
        assert np.array_equal(obs.float().cpu().numpy(), u)
        exp = ow.step(u, np.zeros((n, 1), np.float32))
        np.testing.assert_allclose(reward.cpu().numpy(), exp, rtol=0, atol=5e-6, err_msg=f"step {t}")
        worst = max(worst, float(np.abs(reward.cpu().numpy() - exp).max()))
        if t == 6:
            assert reset and not reward.any()
    got = {k: v.detach().cpu().numpy() for k, v in w.state_dict().items()}
    wworst = 0.0
    for k, v in ow.state().items():
        np.testing.assert_allclose(got[k], v, rtol=0, atol=5e-5, err_msg=k)
        wworst = max(wworst, float(np.abs(got[k] - v).max()))
    print(f"largest differences: reward {worst:.2e}, weights {wworst:.2e}")


def test_reference_test_prediction_bonus_learns():
    """tests/test_mcl.py:17-53 of the reference, on the GPU: default 256x256 CARLE, PredictionBonus, batch_size 2; a
    glider, then idle steps -- after learning to predict the pattern the reward is higher than at the start."""
    from carle_amd.env import CARLE
    from carle_amd.mcl import PredictionBonus, get_glider
    np.random.seed(42)
    torch.random.manual_seed(42)
    env = PredictionBonus(CARLE(device="cuda:0"))
    env.batch_size = 2
    env.reset()
    action = get_glider()
    obs, initial_reward, done, info = env.step(action)
    initial_reward = initial_reward.clone()
    for ii in range(2):
        obs, reward, done, info = env.step(action * (ii % 2))
    obs, second_reward, done, info = env.step(action * 0)
    second_reward = second_reward.clone()
    action = action * 0.
    for step in range(16):
        obs, reward, done, info = env.step(action)
    assert float(reward) > float(initial_reward)
    assert float(reward) > float(second_reward)


def test_reference_test_parsimony_over_prediction_bonus():
    """tests/test_mcl.py:63-100: ParsimonyBonus(PredictionBonus(CARLE)); the reward with ~2048 toggles is less than a
    tenth of the reward with none (observed ratio in the reference: 20.36)."""
    from carle_amd.env import CARLE
    from carle_amd.mcl import PredictionBonus, ParsimonyBonus, get_glider
    np.random.seed(42)
    torch.random.manual_seed(42)
    env = PredictionBonus(CARLE(device="cuda:0"))
    env.batch_size = 2
    env = ParsimonyBonus(env)
    action = get_glider()
    env.reset()
    obs, initial_reward, done, info = env.step(action)
    action = torch.zeros(1, 1, env.action_height, env.action_width)
    for step in range(16):
        obs, reward, done, info = env.step(action)
    r0 = float(reward.mean())
    action[:, :, :env.action_height // 2, :] = 1.0
    obs, reward, done, info = env.step(action)
    r1 = float(reward.mean())
    assert abs(r1) < abs(r0) / 10
    assert 15 < r0 / r1 < 25


def test_unsupported_geometry_and_arguments_fail_loudly():
    from carle_amd import hip, _lib
    from carle_amd.env import CARLE
    from carle_amd.mcl import PredictionBonus
    w = PredictionBonus(CARLE(width=12, height=12, action_width=12, action_height=12, device="cuda:0"))
    w.reset()
    with pytest.raises(NotImplementedError):
        w.step(torch.zeros(1, 1, 12, 12))
    lib = _lib.load()
    x = torch.zeros(1, 1, 16, 16, device="cuda")
    assert lib.carle_ae2d_predict_bonus_f32(x.data_ptr(), None, 1, 16, 16, x.data_ptr(), x.data_ptr(), x.data_ptr(), None,
                                            None) > 0
    assert lib.carle_ae2d_predict_train_f32(x.data_ptr(), x.data_ptr(), 1, 12, 12, x.data_ptr(), x.data_ptr(), x.data_ptr(),
                                            x.data_ptr(), x.data_ptr(), 1.0, None) > 0
    with pytest.raises(ValueError):
        hip.ae2d_predict(x, torch.zeros(1, 1, 8, 8, device="cuda"), x.reshape(-1))


def test_state_dict_and_names_follow_the_reference():
    """PredictionBonus / SurpriseBonus are AE2D subclasses: same state_dict keys as an AE2D on the same environment
    (the reference's recorded state_dict loads strictly), my_name / prediction_steps / ca_steps as mcl.py:565-567, 706."""
    from carle_amd.env import CARLE
    from carle_amd.mcl import AE2D, PredictionBonus, SurpriseBonus
    import carle.mcl as alias
    env = CARLE(width=64, height=64, device="cuda:0")
    p, s, a = PredictionBonus(env), SurpriseBonus(env), AE2D(env)
    assert list(p.state_dict()) == list(a.state_dict()) == list(s.state_dict())
    assert list(p.state_dict()) == list(G.prediction_case("pred64").init_sd)
    assert (p.my_name, p.prediction_steps, p.learning_rate, p.optimizer.param_groups[0]["lr"]) == ("PredictionBonus", 5, 3e-4, 6e-2)
    assert (s.my_name, s.ca_steps) == ("AE2D", 3)
    assert alias.PredictionBonus is PredictionBonus and alias.SurpriseBonus is SurpriseBonus


def test_step_graph_refuses_the_frame_buffer():
    """A replay would keep reading the frames of the captured step: refused, not silently wrong."""
    from carle_amd.env import CARLE
    from carle_amd.mcl import PredictionBonus
    from carle_amd.graphs import StepGraph
    env = CARLE(width=64, height=64, instances=2, device="cuda:0")
    w = PredictionBonus(env, train_predictor=False)
    w.reset()
    a = torch.zeros(2, 1, 64, 64, device="cuda")
    for backend in ("graph", "calls"):
        with pytest.raises(NotImplementedError):
            StepGraph(w, a, backend=backend)


@pytest.mark.parametrize("n", [1, 63, 64, 257, 5000])
def test_bonus_epilogue_is_bit_exact(n):
    """carle_bonus_nonempty_f32 against the fp32 arithmetic of mcl.py:686-694 / 715-723 (two roundings in
    reward + scale * bonus, zero for empty grids), bit for bit, and the live-cell total."""
    from carle_amd import hip
    rng = np.random.default_rng(n)
    err = rng.random(n).astype(np.float32)
    reward = rng.standard_normal((n, 1)).astype(np.float32)
    sums = np.zeros((n, 4), np.float32)
    sums[:, 0] = rng.integers(0, 4, n) * rng.integers(0, 3000, n)       # about a third of the instances empty
    sums[:, 1:] = rng.standard_normal((n, 3))
    for offset, sign, scale in ((0.1, -1.0, 1.0), (0.0, 1.0, 0.5), (0.1, -1.0, 1.7)):
        r = torch.from_numpy(reward.copy()).cuda()
        total = torch.empty((), device="cuda")
        hip.bonus_nonempty_f32(r, torch.from_numpy(err).cuda(), torch.from_numpy(sums).cuda(), offset, sign, scale, total)
        b = (np.float32(offset) + np.float32(sign) * err).astype(np.float32)
        b = np.where(sums[:, 0] == 0, b * np.float32(0.0), b).astype(np.float32)
        exp = (reward[:, 0] + (np.float32(scale) * b).astype(np.float32)).astype(np.float32)
        assert np.array_equal(r.cpu().numpy()[:, 0], exp)
        assert float(total) == pytest.approx(float(sums[:, 0].astype(np.float64).sum()), rel=1e-6)

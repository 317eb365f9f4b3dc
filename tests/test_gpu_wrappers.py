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


def _eval_stack(n, size, graphable=True):
    """RND2D -> AE2D -> SpeedDetector -> PufferDetector as evaluation/eval.py:83-87 stacks them."""
    from carle_amd.env import CARLE
    from carle_amd.mcl import RND2D, AE2D, SpeedDetector, PufferDetector
    torch.manual_seed(3)
    env = CARLE(width=size, height=size, instances=n, device="cuda:0")
    kw = {"train_predictor": False} if graphable else {}
    w = PufferDetector(SpeedDetector(AE2D(RND2D(env, **kw), **kw)))
    w.env.reward_scale, w.reward_scale = 1e-2, 1e-3
    for node in (w.env.env, w.env.env.env):
        node.batch_size = 10 ** 6                           # eval.py:45: no update inside the run
    w.growth_threshold = 5
    w.eval()
    w.reset()
    return env, w


@pytest.mark.parametrize("n,size", [(16, 64), (16, 256), (40, 64)])
def test_eval_stack_steps_without_host_synchronisation(n, size):
    """the evaluation stack (eval.py:83-87) issues no host synchronisation per step -- the reference's
    PufferDetector alone makes two (mcl.py:832,835).  torch's sync debug mode raises on any synchronising call."""
    env, w = _eval_stack(n, size, graphable=False)
    g = torch.Generator(device="cuda").manual_seed(1)
    env.universe = (torch.rand(n, 1, size, size, device="cuda", generator=g) < 0.3).float()
    acts = [(torch.rand(n, 1, 64, 64, device="cuda", generator=g) < 0.05).float() if t % 3 == 0 else
            torch.zeros(n, 1, 64, 64, device="cuda") for t in range(12)]
    sums = []
    for a in acts[:4]:                                      # lazy set-up: workspaces, tables, weight planes, rings
        sums.append(w.step(a)[1].detach().sum())
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        for a in acts[4:]:
            obs, reward, done, info = w.step(a)
            sums.append(reward.detach().sum())              # the harness's per-step reduction (eval.py:69-70), deferred
    finally:
        torch.cuda.set_sync_debug_mode("default")
    trace = torch.stack(sums).cpu()
    assert torch.isfinite(trace).all() and len(w.cells) <= w.growth_threshold


@pytest.mark.parametrize("backend", ["graph", "calls"])
@pytest.mark.parametrize("n,size", [(3, 64), (16, 256)])
def test_step_graph_replays_the_evaluation_stack(n, size, backend):
    """StepGraph accepts SpeedDetector / PufferDetector: graph replay == eager loop for the whole eval.py stack,
    rewards, centre of mass, speed, growth history and counters."""
    from carle_amd.graphs import StepGraph
    g = torch.Generator(device="cuda").manual_seed(2)
    u0 = (torch.rand(n, 1, size, size, device="cuda", generator=g) < 0.3).float()
    acts = [(torch.rand(n, 1, 64, 64, device="cuda", generator=g) < 0.05).float() if t in (0, 1, 9) else
            torch.zeros(n, 1, 64, 64, device="cuda") for t in range(14)]
    res = []
    for mode in ("eager", "graph"):
        env, w = _eval_stack(n, size)
        env.universe = u0.clone()
        static = torch.zeros_like(acts[0])
        graph = StepGraph(w, static, backend=backend) if mode == "graph" else None
        out = []
        for a in acts:
            if graph is None:
                o, r, d, i = w.step(a)
            elif backend == "calls":
                o, r, d, i = graph(a)                       # the recorded calls take the action tensor as it is
            else:
                static.copy_(a)
                o, r, d, i = graph()
            out.append((o.clone(), r.clone()))
        res.append((out, w.cells, w.live_cells, w.env.center_of_mass.clone(), w.env.speed.clone(), env.step_number))
    (oe, ce, le, come, spe, sne), (og, cg, lg, comg, spg, sng) = res
    for t, ((o0, r0), (o1, r1)) in enumerate(zip(oe, og)):
        assert torch.equal(o0, o1), t
        assert torch.allclose(r0, r1, atol=1e-6, rtol=1e-6), (t, (r0 - r1).abs().max())
    assert ce == cg and le == lg and torch.equal(come, comg) and torch.equal(spe, spg) and sne == sng
    assert any(float(r.max()) > 0.9 for _, r in og[8:]) or size == 64     # the growth bonus fired somewhere (soups grow or not)


def test_puffer_history_matches_a_host_restatement():
    """The device ring of PufferDetector against a host restatement of mcl.py:832-851 (append / pop / clear / slope),
    with the threshold changed mid-run and a history assigned by the caller."""
    from carle_amd.env import CARLE
    from carle_amd.mcl import PufferDetector
    env = CARLE(width=64, height=64, instances=3, device="cuda:0")
    env.rules_from_string("B3/S012345678")                  # growing rule: slopes > 0
    w = PufferDetector(env)
    w.growth_threshold = 4
    w.reset()
    g = torch.Generator(device="cuda").manual_seed(4)
    env.universe = (torch.rand(3, 1, 64, 64, device="cuda", generator=g) < 0.05).float()
    cells, bonus = [], []
    for t in range(20):
        a = torch.zeros(3, 1, 64, 64, device="cuda")
        if t in (7, 8):
            a[1, 0, 3, 3] = 1.0
        if t == 12:
            w.growth_threshold = 6
        obs, reward, done, info = w.step(a)
        live = float(obs.sum())
        exp = 0.0
        if not bool(a.sum()):
            cells.append(live)
            if len(cells) > w.growth_threshold:
                slope = cells[-1] - cells[0]
                cells.pop(0)
                exp = 1.0 if slope > 0.01 else 0.0
        else:
            cells = []
        assert w.cells == cells and w.live_cells == live, t
        assert reward.cpu().reshape(-1).tolist() == [exp] * 3, t
        bonus.append(exp)
    assert sum(bonus) > 0
    w.cells = [1.0, 2.0]
    assert w.cells == [1.0, 2.0]
    w.step(torch.zeros(3, 1, 64, 64, device="cuda"))
    assert w.cells[:2] == [1.0, 2.0] and len(w.cells) == 3


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


@pytest.mark.parametrize("storage,n,size,stack", [
    ("f32", 3, 64, False), ("f32", 1, 64, False), ("f32", 16, 256, True), ("f32", 5, 64, True),
    ("u8", 3, 64, False), ("b1", 2, 128, False), ("u8", 2, 256, True),
])
@pytest.mark.parametrize("backend", ["graph", "calls"])
def test_step_graph_replays_eager_steps(storage, n, size, stack, backend):
    """carle_amd.graphs.StepGraph: a HIP graph of env.step(action) gives the eager loop's observations, rewards and
    counters -- bare CARLE in every storage, and the RND2D + AE2D + CornerBonus stack (forward only); survives
    reset(), an assigned universe, an eager step in between, a master reset and a rule change."""
    from carle_amd.env import CARLE
    from carle_amd.mcl import RND2D, AE2D, CornerBonus
    from carle_amd.graphs import StepGraph

    def make():
        torch.manual_seed(5)
        env = CARLE(instances=n, width=size, height=size, device="cuda:0", storage=storage,
                    **({"obs": "f32"} if storage != "f32" else {}))
        if stack:
            env = CornerBonus(AE2D(RND2D(env, train_predictor=False), train_predictor=False))
        env.reset()
        return env

    g = torch.Generator(device="cuda").manual_seed(11)
    acts = [(torch.rand(n, 1, 64, 64, device="cuda", generator=g) < 0.12).float() for _ in range(9)]
    acts[4] = torch.ones_like(acts[0])                      # master reset
    pattern = (torch.rand(n, 1, size, size, device="cuda", generator=g) < 0.3).float()

    def drive(env, stepper, copy_in):
        out = []
        for t, a in enumerate(acts):
            if t == 2:
                env.reset()
            if t == 3:
                (env if env.inner_env is None else env.inner_env).universe = pattern.clone()
            if t == 6:
                env.rules_from_string("B36/S23")
            if t == 7:
                o, r, d, i = env.step(a)                     # an eager step between replays
            else:
                if copy_in is not None:
                    copy_in.copy_(a)
                o, r, d, i = stepper(a)
            base = env if env.inner_env is None else env.inner_env
            out.append((o.float().clone(), r.clone(), base.step_number, base.steps_since_action))
        return out

    ref_env = make()
    ref = drive(ref_env, ref_env.step, None)
    env = make()
    static = torch.zeros_like(acts[0])
    if backend == "calls" and storage == "u8":
        # uint8 state with fp32 actions / observations: the conversions are torch kernels, not library calls
        with pytest.raises(NotImplementedError):
            StepGraph(env if not stack else env.env, static, backend="calls")
        return
    if backend == "calls" and stack:
        # CornerBonus adds its terms with torch operations: not a step made of library calls alone -- refused at
        # construction (the replayed step would not reproduce the eager one); without it the stack replays
        with pytest.raises(NotImplementedError):
            StepGraph(env, static, backend="calls")
        env = env.env
    graph = StepGraph(env, static, backend=backend)
    if backend == "calls" and stack:
        ref_env = make().env
        ref = drive(ref_env, ref_env.step, None)
    got = drive(env, (lambda a: graph(a)) if backend == "calls" else (lambda a: graph()), None if backend == "calls" else static)
    for t, ((o0, r0, s0, a0), (o1, r1, s1, a1)) in enumerate(zip(ref, got)):
        assert torch.equal(o0, o1), t
        assert torch.allclose(r0, r1, atol=1e-6, rtol=0), (t, (r0 - r1).abs().max())
        assert (s0, a0) == (s1, a1), t


def test_step_graph_refuses_what_it_cannot_replay():
    from carle_amd.env import CARLE
    from carle_amd.mcl import RND2D, SpeedDetector
    from carle_amd.agents import RandomAgent
    from carle_amd.graphs import StepGraph
    env = CARLE(instances=2, width=64, height=64, device="cuda:0")
    a = torch.zeros(2, 1, 64, 64, device="cuda")
    with pytest.raises(RuntimeError):
        StepGraph(env, a)                                   # before reset()
    env.reset()
    with pytest.raises(TypeError):
        StepGraph(env, a.cpu())
    with pytest.raises(TypeError):
        StepGraph(env, RandomAgent(instances=2, action_width=64, action_height=64, device="cuda:0", fused=True)(env.universe))
    with pytest.raises(NotImplementedError):
        StepGraph(RND2D(env), a)                            # trains its predictor
    g = StepGraph(SpeedDetector(env), a)                    # device-side history: replayable since round 3
    assert g()[1].shape == (2, 1)


@pytest.mark.parametrize("backend", ["graph", "calls"])
def test_step_graph_built_behind_a_pending_one_launch_step(backend):
    """A StepGraph constructed right after an eager one-launch CA + RND2D step with attached reward statistics: that
    step's reward moments are still per-workgroup partial sums registered in the status block, its counters PENDING.
    The capture settles them first -- the statistics and counters of eager step + replays equal the all-eager run."""
    from carle_amd import dist as cd
    from carle_amd.env import CARLE
    from carle_amd.mcl import RND2D
    from carle_amd.graphs import StepGraph
    n = 40                                                   # > 16 grids: the one-launch fused CA + RND2D step
    g = torch.Generator(device="cuda").manual_seed(3)
    u0 = (torch.rand(n, 1, 64, 64, device="cuda", generator=g) < 0.4).float()
    acts = [(torch.rand(n, 1, 64, 64, device="cuda", generator=g) < 0.1).float() for _ in range(6)]
    res = []
    for mode in ("eager", "graph"):
        torch.manual_seed(2)
        env = CARLE(width=64, height=64, instances=n, device="cuda:0")
        w = RND2D(env, train_predictor=False)
        stats = cd.RunningRewardStats("cuda:0", sync_every=1000)
        assert stats.attach(w)                               # the statistics of the wrapper's (final) reward
        w.reset()
        env.universe = u0.clone()
        rewards = [w.step(acts[0])[1].clone()]
        if mode == "graph":
            assert env.__dict__.get("_moments_pending")      # the situation under test
            graph = StepGraph(w, torch.zeros_like(acts[0]), backend=backend)
        for a in acts[1:]:
            if mode == "eager":
                rewards.append(w.step(a)[1].clone())
            elif backend == "calls":
                rewards.append(graph(a)[1].clone())
            else:
                graph.action.copy_(a)
                rewards.append(graph()[1].clone())
        status = env.settled_status().cpu().tolist()[4:12]
        res.append((torch.stack(rewards).cpu().numpy(), stats._pending.cpu().numpy().copy(), status,
                    env.universe.cpu().numpy()))
    a, b = res
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[0], b[0])
    assert a[1][0] == len(acts) * n
    np.testing.assert_allclose(a[1], b[1], rtol=1e-12)
    assert a[2] == b[2], (a[2], b[2])


def test_step_graph_calls_refuses_a_foreign_stream():
    """backend='calls' replays on the stream that was current at capture; under another stream context the state
    copy and the caller's action producer would race with the recorded launches: refused."""
    from carle_amd.env import CARLE
    from carle_amd.graphs import StepGraph
    env = CARLE(instances=3, width=64, height=64, device="cuda:0")
    env.reset()
    a = torch.zeros(3, 1, 64, 64, device="cuda")
    graph = StepGraph(env, a, backend="calls")
    graph(a)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with pytest.raises(RuntimeError, match="stream"):
            graph(a)
    torch.cuda.synchronize()
    assert graph(a)[0].shape == (3, 1, 64, 64)


@pytest.mark.parametrize("n,shape", [(3, (1, 64, 64)), (16, (1, 256, 256)), (5, (1, 7, 9)), (2, (1, 130, 130)), (1, (4099,)),
                                     (4, (1, 6, 6))])
def test_instance_sums_every_plane_combination(n, shape):
    """carle_instance_sums_f32 (the reduction behind CornerBonus / SpeedDetector / ParsimonyBonus): every combination
    of present weight planes, vector (elems % 4 == 0) and scalar paths, small and large block sizes, unaligned
    plane views -- against fp64 torch sums."""
    from carle_amd import hip
    g = torch.Generator(device="cuda").manual_seed(n * 31 + len(shape))
    x = torch.randn(n, *shape, device="cuda", generator=g)
    elems = x[0].numel()
    planes = [torch.randn(elems + 1, device="cuda", generator=g) for _ in range(3)]
    for mask in range(8):
        for shift in (0, 1):                               # shift 1: planes start 4 bytes off a 16-byte boundary
            ws = [planes[j][shift:shift + elems] if (mask >> j) & 1 else None for j in range(3)]
            got = hip.instance_sums_f32(x, *ws).double().cpu()
            xd = x.double().reshape(n, -1)
            exp = torch.zeros(n, 4, dtype=torch.float64)
            exp[:, 0] = xd.sum(1).cpu()
            for j in range(3):
                if ws[j] is not None:
                    exp[:, 1 + j] = (xd * ws[j].double()).sum(1).cpu()
            scale = xd.abs().sum(1).cpu().clamp_min(1.0).unsqueeze(1)
            assert torch.all((got - exp).abs() <= 2e-6 * scale * 4), (mask, shift, (got - exp).abs().max())

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
    # a stacked wrapper that adds its bonus after the last kernel that can reduce the rewards: attach refuses, after_step
    # falls back to the stand-alone moments launch.  (AE2D(RND2D(env)) forward-only IS accepted since round 5: AE2D's
    # bonus kernel leaves the moments, test_pair_finish_kernel_leaves_the_reward_moments.)
    from carle_amd.mcl import CornerBonus
    assert cd.RunningRewardStats("cuda:0").attach(AE2D(RND2D(CARLE(**kw), train_predictor=False), train_predictor=False))
    for stack in (CornerBonus(RND2D(CARLE(**kw), train_predictor=False)), AE2D(RND2D(CARLE(**kw), train_predictor=False))):
        other = cd.RunningRewardStats("cuda:0")
        assert not other.attach(stack)
        stack.eval(); stack.reset()
        obs, reward, done, info = stack.step(torch.zeros(n, 1, asize, asize))
        other.after_step(stack, reward)
        np.testing.assert_allclose(other.mean, float(reward.double().mean()), rtol=1e-9)


@pytest.mark.parametrize("train,storage,size,n", [(False, "f32", 256, 16), (True, "f32", 256, 16), (False, "b1", 128, 37),
                                                  (True, "f32", 128, 1030), (False, "f32", 64, 40), (False, "f32", 64, 2050)])
def test_pair_finish_kernel_leaves_the_reward_moments(train, storage, size, n):
    """AE2D(RND2D(CARLE)): the rewards are final in the pair's finish kernel (tiled sizes: carle_rnd_ae_bonus /
    carle_rnd_ae_train with moment_partials) or in AE2D's bonus kernel (64x64 forward passes, round 5:
    carle_ae2d_bonus_b1_stats on the bit-packed observation RND2D's one-launch step hands over), which leaves their
    partial sums for the next environment step to merge -- the statistics equal those of explicit update() calls, across
    reset() and a change of mode."""
    import os
    from carle_amd import dist as cd
    from carle_amd.env import CARLE
    from carle_amd.mcl import RND2D, AE2D
    if size == 64 and os.environ.get("CARLE_TWO_LAUNCH_STEP", "0") not in ("", "0"):
        pytest.skip("the 64x64 hand-off comes from RND2D's ONE-launch step (tools/test_all_paths.sh forces two launches)")
    torch.manual_seed(5)
    env = CARLE(width=size, height=size, instances=n, device="cuda:0", storage=storage)
    w = AE2D(RND2D(env, train_predictor=train), train_predictor=train)
    w.env.batch_size = w.batch_size = 3
    w.eval(); w.reset()
    attached = cd.RunningRewardStats("cuda:0", sync_every=3)
    explicit = cd.RunningRewardStats("cuda:0", sync_every=3)
    assert attached.attach(w)
    g = torch.Generator(device="cuda").manual_seed(3)
    env.universe = (torch.rand(n, 1, size, size, device="cuda", generator=g) < 0.3).float()
    rewards = []
    for t in range(8):
        if t == 5:
            w.reset()
        a = (torch.rand(n, 1, 64, 64, device="cuda", generator=g) < 0.2).float()
        obs, reward, done, info = w.step(a)
        assert w._moments_merged
        attached.after_step(w, reward)
        explicit.update(reward)
        rewards.append(reward.detach().clone())
    attached.flush(); explicit.flush()
    allr = torch.cat(rewards).double().cpu().numpy().reshape(-1)
    np.testing.assert_allclose([attached.count, attached.mean, attached.var],
                               [allr.size, allr.mean(), allr.var()], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(attached.state.cpu().numpy(), explicit.state.cpu().numpy(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(attached.last_batch.cpu().numpy(), explicit.last_batch.cpu().numpy(), rtol=1e-9, atol=1e-12)
    # a step the pair cannot take (one wrapper training, the other not): the stand-alone moments kernel counts it
    w.train_predictor = not train
    obs, reward, done, info = w.step(torch.zeros(n, 1, 64, 64, device="cuda"))
    assert not w._moments_merged
    before = attached.count
    attached.after_step(w, reward); attached.flush()
    assert attached.count == before + n


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


@pytest.mark.parametrize("world", [1, 2, 8])
def test_merge_moments_kernel_matches_numpy(world):
    """carle_merge_moments on a fabricated all-gather result [world, 3] (what RCCL delivers at
    N GPUs), including an empty rank: Chan merge in rank order vs numpy on the pooled samples."""
    from carle_amd import hip
    rng = np.random.default_rng(world)
    parts = [rng.normal(2.0 * r, 1.0 + r, size=(0 if (world > 2 and r == 3) else 50 + 7 * r)) for r in range(world)]
    rows = [[p.size, p.mean() if p.size else 0.0, ((p - p.mean()) ** 2).sum() if p.size else 0.0] for p in parts]
    gathered = torch.tensor(rows, dtype=torch.float64, device="cuda")
    prev = rng.normal(-1.0, 0.5, size=33)
    state = torch.tensor([prev.size, prev.mean(), ((prev - prev.mean()) ** 2).sum()], dtype=torch.float64,
                         device="cuda")
    last = torch.zeros(3, dtype=torch.float64, device="cuda")
    pending = torch.ones(3, dtype=torch.float64, device="cuda")
    hip.merge_moments(gathered, world, state, last, pending)
    pooled = np.concatenate(parts)
    both = np.concatenate([prev, pooled])
    np.testing.assert_allclose(last.cpu().numpy(), [pooled.size, pooled.mean(), ((pooled - pooled.mean()) ** 2).sum()],
                               rtol=1e-12)
    np.testing.assert_allclose(state.cpu().numpy(), [both.size, both.mean(), ((both - both.mean()) ** 2).sum()],
                               rtol=1e-12)
    assert pending.abs().sum().item() == 0.0


def _flush_worker(rank, world, port, q):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from carle_amd import dist as cd
    # two ranks on the ONE device of the test box: gloo moves the device tensors (RCCL refuses two
    # ranks per device); flush() takes its device branch: all_gather_into_tensor + carle_merge_moments
    r, w, dev = cd.init_from_env("gloo", share_devices=True)
    try:
        assert w == world and dev.type == "cuda"
        stats = cd.RunningRewardStats(dev, sync_every=2)
        g = torch.Generator().manual_seed(100 + rank)
        seen = []
        for t in range(5):
            x = (torch.randn(64 + 3 * rank, 1, generator=g) * (t + 1) + rank).to(dev)
            seen.append(x.cpu())
            stats.update(x)
        stats.flush()
        torch.cuda.synchronize()
        q.put((rank, stats.state.cpu().tolist(), stats.last_batch.cpu().tolist(),
               torch.cat(seen).reshape(-1).tolist()))
    finally:
        dist.destroy_process_group()


def test_running_stats_flush_two_ranks_on_device():
    """World-size-2 run of RunningRewardStats.flush()'s DEVICE branch (all-gather of the 24-byte
    triples into a device tensor + carle_merge_moments with world = 2)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_flush_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    allx = np.asarray(res[0][3] + res[1][3], np.float64)
    for r in res:
        np.testing.assert_allclose(r[1], [allx.size, allx.mean(), ((allx - allx.mean()) ** 2).sum()], rtol=1e-10)
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2]      # bit-identical on both ranks


def _dp_inputs(size, total, steps):
    g = torch.Generator().manual_seed(size + total)
    u0 = (torch.rand(total, 1, size, size, generator=g) < 0.35).float()
    acts = (torch.rand(steps, total, 1, 64, 64, generator=g) < 0.1).float()
    return u0, acts


def _dp_trajectory(size, start, count, total, steps, batch_size, sync, midwindow=False, stack="rnd+ae"):
    """`steps` training steps of AE2D(RND2D(CARLE)) on instances [start, start + count) of the seeded batch; returns
    (rewards (steps, count), state_dict as numpy).  ``midwindow``: after step 5 (inside the second update window) both
    wrappers also take ``update_predictor(mean loss of the current observation)`` -- torch's own optimiser step on a
    caller's loss while the kernel gradient of the window is in flight."""
    from carle_amd.env import CARLE
    import carle_amd.mcl as M
    u0, acts = _dp_inputs(size, total, steps)
    torch.manual_seed(0)
    env = CARLE(width=size, height=size, instances=count, device="cuda:0")
    # (stack: "rnd+ae" = AE2D(RND2D(CARLE)); otherwise one wrapper class by name: PredictionBonus / SurpriseBonus)
    ae = M.AE2D(M.RND2D(env)) if stack == "rnd+ae" else getattr(M, stack)(env)
    wrappers = [ae, ae.env] if stack == "rnd+ae" else [ae]
    ae.reset()
    for w in wrappers:
        w.batch_size = batch_size
    if sync:
        from carle_amd import dist as cd
        assert cd.data_parallel(ae, broadcast=True) == wrappers
    env.universe = u0[start:start + count].cuda()
    rewards = []
    for t in range(steps):
        obs, reward = ae.step(acts[t, start:start + count].cuda())[:2]
        rewards.append(reward.reshape(-1).cpu().numpy().copy())
        if midwindow and t == 5:
            for w in (ae.env, ae):
                w.eval()                                     # Dropout = identity, as in the kernels
                w.update_predictor(torch.mean(w._torch_loss(obs)))
    torch.cuda.synchronize()
    return np.stack(rewards), {k: v.cpu().numpy().copy() for k, v in ae.state_dict().items()}


def _dp_worker(rank, world, port, size, total, steps, batch_size, q, midwindow=False, stack="rnd+ae"):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from carle_amd import dist as cd
    cd.init_from_env("gloo", share_devices=True)             # two ranks on the one device of the test box
    try:
        start, count = cd.shard(total)
        q.put((rank, start, count) + _dp_trajectory(size, start, count, total, steps, batch_size, True, midwindow, stack))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("size,total,midwindow,stack", [(64, 11, False, "rnd+ae"), (256, 6, False, "rnd+ae"), (64, 11, True, "rnd+ae"),
                                                        (64, 7, False, "PredictionBonus"), (128, 5, False, "SurpriseBonus")])
def test_data_parallel_predictors_take_the_unsharded_updates(size, total, midwindow, stack):
    """carle_amd.dist.data_parallel: two ranks, each stepping its shard of the batch through AE2D(RND2D(CARLE)) in
    training mode (HIP gradient kernels; 64x64: the fused kernels with unequal shards 6 + 5, 256x256: the paired tiled
    call), all-reduce their flat gradients before every Adam step.  Rewards and weights must be those of ONE process
    stepping the whole batch -- the reference's loss is the mean over the whole batch (carle/mcl.py:362,376)."""
    import socket
    import torch.multiprocessing as mp
    steps, batch_size = 9, 4                                 # two Adam steps inside the trajectory
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, size, total, steps, batch_size, q, midwindow, stack)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    whole_rewards, whole_sd = _dp_trajectory(size, 0, total, total, steps, batch_size, False, midwindow, stack)
    for k in whole_sd:
        assert np.array_equal(res[0][4][k], res[1][4][k]), k                      # the ranks never drift apart
        np.testing.assert_allclose(res[0][4][k], whole_sd[k], rtol=0, atol=2e-5, err_msg=k)
    sharded = np.concatenate([res[0][3], res[1][3]], axis=1)
    assert sharded.shape == whole_rewards.shape and res[0][2] + res[1][2] == total
    np.testing.assert_allclose(sharded[:batch_size], whole_rewards[:batch_size], rtol=0, atol=1e-6)   # before any update
    np.testing.assert_allclose(sharded, whole_rewards, rtol=0, atol=2e-5)
    assert np.abs(whole_rewards[-1] - whole_rewards[0]).max() > 1e-3              # the updates did change the rewards


@pytest.mark.parametrize("n", [1, 7, 16, 40])
def test_run_merges_reward_moments_like_the_step_loop(n):
    """CARLE.run with attached statistics: the k steps of a small batch run in one launch (ca_env_run_small64_kernel)
    and merge their reward moments per step in the kernel; the pending triple, the counters and the status block must
    equal those of k separate steps (n = 40: the two-launch path).  A master reset sits in the middle of the run."""
    from carle_amd import dist as cd
    from carle_amd.env import CARLE
    g = torch.Generator().manual_seed(n)
    u0 = (torch.rand(n, 1, 64, 64, generator=g) < 0.4).float()
    acts = (torch.rand(9, n, 1, 64, 64, generator=g) < 0.1).float()
    acts[4] = 1.0                                            # master reset at step 5
    acts[6] = 0.0                                            # a step without any toggle
    res = []
    for mode in ("loop", "run"):
        env = CARLE(width=64, height=64, instances=n, device="cuda:0")
        stats = cd.RunningRewardStats("cuda:0", sync_every=100)
        assert stats.attach(env)
        env.reset()
        env.universe = u0.clone().cuda()
        if mode == "loop":
            for t in range(9):
                obs, reward, done, info = env.step(acts[t].cuda())
                assert env._moments_merged
        else:
            env.run(actions=acts[:4].cuda())
            obs, reward, done, info = env.run(actions=acts[4:].cuda())
            assert env._moments_merged
        # three action-free steps on top (n > 16: all generations in one launch of ca_run_noaction64_kernel)
        if mode == "loop":
            for t in range(3):
                obs, reward, done, info = env.step(torch.zeros(n, 1, 64, 64, device="cuda"))
        else:
            obs, reward, done, info = env.run(steps=3)
        # (words 0..3 are the transient action-flag slots: an explicit all-zero action leaves NOT_ALL_ONES behind, no action nothing)
        res.append((obs.cpu().numpy(), reward.cpu().numpy(), env.settled_status().cpu().tolist()[4:12], stats._pending.cpu().numpy(),
                    env.step_number, env.steps_since_action, env.master_resets))
    a, b = res
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert a[2] == b[2], (a[2], b[2])
    assert np.array_equal(a[3], b[3]) and a[3][0] == 12 * n
    assert a[4:] == b[4:] == (4, 0, 1) or a[4:] == b[4:]


@pytest.mark.parametrize("kind", ["bare", "rnd", "pair256", "rnd_ae64", "stack"])
def test_trace_ring_keeps_every_steps_reward_total(kind):
    """RunningRewardStats.start_trace / drain_trace (the harness's per-step reward totals, train_mcl.py:68-69 / eval.py:69-70):
    the kernel that writes a step's reward moments takes the next row of a device ring (CARLE_ST_TRACE_*) -- bare CARLE, the
    fused RND2D step (moments settled one launch later), the tiled pair, AE2D's stats kernel at 64x64 -- or, for a stack the
    epilogues cannot serve, the stand-alone moments kernel writes the row.  Totals equal reward.sum() per step, across
    reset(), a drain in the middle, run() (one row per step) and a ring shorter than the loop (the last rows survive)."""
    from carle_amd import dist as cd
    from carle_amd.env import CARLE
    from carle_amd.mcl import RND2D, AE2D, CornerBonus
    torch.manual_seed(6)
    n = 40
    size = 256 if kind == "pair256" else 64
    env = CARLE(width=size, height=size, instances=n, device="cuda:0")
    if kind == "bare":
        w = env
    elif kind == "rnd":
        w = RND2D(env, train_predictor=False)
    elif kind in ("pair256", "rnd_ae64"):
        w = AE2D(RND2D(env, train_predictor=False), train_predictor=False)
    else:
        w = CornerBonus(RND2D(env, train_predictor=False))
    if w is not env:
        w.eval()
    w.reset()
    stats = cd.RunningRewardStats("cuda:0", sync_every=1 << 30)
    attached = stats.start_trace(w, 5)
    assert attached is (kind != "stack")
    g = torch.Generator(device="cuda").manual_seed(9)
    env.universe = (torch.rand(n, 1, size, size, device="cuda", generator=g) < 0.3).float()
    expected, got = [], []

    def step():
        a = (torch.rand(n, 1, 64, 64, device="cuda", generator=g) < 0.1).float()
        obs, reward, done, info = w.step(a)
        stats.after_step(w, reward)
        expected.append(float(reward.double().sum()))

    for t in range(3):
        step()
    got += stats.drain_trace().tolist()                      # a drain in the middle: the ring starts over
    step()
    w.reset()                                                # ... and a reset between two steps of a loop
    env.universe = (torch.rand(n, 1, size, size, device="cuda", generator=g) < 0.3).float()
    for t in range(3):
        step()
    got += stats.drain_trace().tolist()
    np.testing.assert_allclose(got, expected, rtol=1e-6, atol=1e-9)
    assert len(got) == 7 and (kind == "bare" or min(abs(x) for x in expected) > 0)
    for t in range(8):                                       # more steps than rows: the last five survive, oldest first
        step()
    np.testing.assert_allclose(stats.drain_trace().tolist(), expected[-5:], rtol=1e-6, atol=1e-9)
    assert stats.count == n * len(expected)
    if kind == "bare":                                       # k steps in ONE launch: k rows
        obs, reward, done, info = env.run(steps=3)
        assert env._moments_merged
        stats.after_step(env, reward, steps=3)
        rows = stats.drain_trace().tolist()
        assert rows == [0.0, 0.0, 0.0] and stats.count == n * (len(expected) + 3)

"""world_size-2 gloo tests (CPU) of the multi-GPU host logic: sharding, the running-stat
all-gather + Chan merge, and the global master-reset predicate."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _moments(x):
    x = np.asarray(x, np.float64)
    return torch.tensor([x.size, x.mean() if x.size else 0.0, ((x - x.mean()) ** 2).sum() if x.size else 0.0],
                        dtype=torch.float64)


def _worker(rank, world, port, total, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from carle_amd import dist as cd
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(5)
        rewards = rng.normal(size=(3, total)).astype(np.float32)      # 3 steps, whole batch
        start, count = cd.shard(total)
        stats = cd.RunningRewardStats("cpu")
        totals = []
        for t in range(3):
            local = rewards[t, start:start + count]
            stats.update_from_local(_moments(local))
            totals.append(stats.last_total)
        flag = torch.tensor([1 if rank == 1 else 0], dtype=torch.int32)
        g1 = int(cd.global_all_ones(flag).item())
        g0 = int(cd.global_all_ones(torch.zeros(1, dtype=torch.int32)).item())
        q.put((rank, start, count, stats.state.tolist(), totals, g1, g0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [4096, 7])
def test_running_stats_two_ranks_gloo(total):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    rng = np.random.default_rng(5)
    rewards = rng.normal(size=(3, total)).astype(np.float32).astype(np.float64)
    flat = rewards.reshape(-1)
    exp = [flat.size, flat.mean(), ((flat - flat.mean()) ** 2).sum()]
    # shards tile the batch exactly
    assert res[0][1] == 0 and res[0][1] + res[0][2] == res[1][1] and res[1][1] + res[1][2] == total
    for r in res:
        np.testing.assert_allclose(r[3], exp, rtol=1e-12)
        np.testing.assert_allclose(r[4], rewards.sum(axis=1), rtol=1e-10, atol=1e-9)
        assert r[5] == 1 and r[6] == 0
    assert res[0][3] == res[1][3]              # bit-identical statistics on every rank


class _Plain:
    """Stands in for an environment / a wrapper on the CPU: only the attributes the host logic looks at."""
    _moments_merged = False


def _host_logic_worker(rank, world, port, total, q):
    """The batch-global exchanges added in round 5, host side on CPU tensors: the trace ring of per-step reward moments
    drained with ONE all-reduce, the two action flags OR-ed over the ranks, which wrappers exchange their scalars."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from carle_amd import dist as cd
    assert not cd.exchanging() and cd.batch_scalar_group(_Plain()) is None      # no group yet: nothing to exchange
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(11)
        rewards = rng.normal(size=(7, total)).astype(np.float32)
        start, count = cd.shard(total)
        stats = cd.RunningRewardStats("cpu", sync_every=1 << 30)
        env = _Plain()
        assert stats.start_trace(env, 5) is False                               # nothing to attach to: the moments path
        drained = []
        for t in range(7):                                                      # 7 steps through a ring of 5 rows, drained at 3
            stats.after_step(env, torch.from_numpy(rewards[t, start:start + count]).reshape(-1, 1))
            if t == 2:
                drained.append(stats.drain_trace())
        drained.append(stats.drain_trace())
        # flags: [ANY_NONZERO(0), ANY_NONZERO(1), NOT_ALL_ONES(0), NOT_ALL_ONES(1)]
        status = torch.zeros(32, dtype=torch.int32)
        status[1] = 1 if rank == 1 else 0                                       # slot 1: rank 1 saw a non-zero element
        status[3] = 1 if rank == 0 else 0                                       # slot 1: rank 0 saw an element != 1
        status[0] = 0
        status[2] = 0                                                           # slot 0: nobody saw anything
        cd.all_reduce_flags(status, 1)
        cd.all_reduce_flags(status, 0)
        w = _Plain()
        grp = cd.batch_scalar_group(w)
        w.__dict__["batch_scalars"] = "per_rank"
        q.put((rank, [d.tolist() for d in drained], stats.state.tolist(), status[:4].tolist(), grp,
               cd.batch_scalar_group(w), cd.exchanging()))
    finally:
        dist.destroy_process_group()


def test_trace_flags_and_scalar_groups_two_ranks_gloo():
    world, total = 2, 9
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_host_logic_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    rewards = np.random.default_rng(11).normal(size=(7, total)).astype(np.float32).astype(np.float64)
    sums = rewards.sum(axis=1)
    flat = rewards.reshape(-1)
    for r in res:
        first, second = r[1]
        np.testing.assert_allclose(first, sums[:3], rtol=1e-10, atol=1e-9)      # the WHOLE batch's totals, on every rank
        np.testing.assert_allclose(second, sums[3:], rtol=1e-10, atol=1e-9)     # the ring started over after the drain
        np.testing.assert_allclose(r[2], [flat.size, flat.mean(), ((flat - flat.mean()) ** 2).sum()], rtol=1e-10)
        assert r[3] == [0, 1, 0, 1]                                             # OR over the ranks, slot by slot
        assert r[4] == (None,) and r[5] is None and r[6] is True
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2]                    # bit-identical on both ranks


def test_trace_ring_wraps_to_the_last_rows():
    from carle_amd import dist as cd
    stats = cd.RunningRewardStats("cpu", sync_every=1 << 30)
    env = _Plain()
    stats.start_trace(env, 4)
    for t in range(6):
        stats.after_step(env, torch.full((3, 1), float(t)))
    assert stats.drain_trace().tolist() == [6.0, 9.0, 12.0, 15.0]               # the last four of six steps, oldest first
    stats.after_step(env, torch.full((3, 1), 7.0))
    assert stats.drain_trace().tolist() == [21.0]
    assert stats.count == 21


def test_shard_partition():
    from carle_amd import dist as cd
    for total in (0, 1, 7, 8, 4096, 32768):
        for w in (1, 2, 3, 8):
            got = [cd.shard(total, r, w) for r in range(w)]
            assert got[0][0] == 0 and sum(c for _, c in got) == total
            for (s0, c0), (s1, c1) in zip(got, got[1:]):
                assert s0 + c0 == s1 and 0 <= c0 - c1 <= 1


def test_chan_merge_matches_numpy():
    from carle_amd import dist as cd
    rng = np.random.default_rng(0)
    parts = [rng.normal(3, 2, size=k) for k in (5, 0, 17, 1)]
    merged = cd.chan_merge(torch.stack([_moments(p) for p in parts]))
    allx = np.concatenate(parts)
    np.testing.assert_allclose(merged.numpy(), [allx.size, allx.mean(), ((allx - allx.mean()) ** 2).sum()],
                               rtol=1e-12)


# ---------------------------------------------------------------------------------------------
# bench.py's own launcher (python bench.py --gpus N outside torch.distributed.run)
# ---------------------------------------------------------------------------------------------
import json
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags, env=None, timeout=240):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True,
                          text=True, timeout=timeout, cwd=ROOT, env=e)


@pytest.mark.parametrize("world", [2, 3])
def test_bench_launcher_spawns_one_rank_per_gpu(world):
    """`bench.py --gpus N` starts N fresh rank processes with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set, they rendezvous (gloo here), and rank 0 prints ONE JSON line with n_gpus == N."""
    out = _bench("--gpus", str(world), "--backend", "gloo", "--launch-check")
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["master"].startswith("127.0.0.1:")
    assert [(r["rank"], r["local_rank"], r["world_size"]) for r in d["ranks"]] == \
        [(r, r, world) for r in range(world)]


def test_bench_under_torch_distributed_run():
    """The driver's launch form: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` -- every process is one rank (RANK / LOCAL_RANK / WORLD_SIZE from the env),
    nobody spawns anything, rank 0 prints the one JSON line."""
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--launch-check"],
                         capture_output=True, text=True, timeout=240, cwd=ROOT, env=e)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and [(r["rank"], r["local_rank"], r["world_size"]) for r in d["ranks"]] == [(0, 0, 2), (1, 1, 2)]


def test_bench_refuses_mismatched_world():
    # --gpus disagrees with WORLD_SIZE: fail loudly before the rendezvous (no hang, no 1-GPU number)
    out = _bench("--gpus", "4", "--launch-check", env={"WORLD_SIZE": "2", "RANK": "0"}, timeout=60)
    assert out.returncode == 2 and "WORLD_SIZE=2" in out.stderr
    # more GPUs asked for than the node has (none here): refuse instead of measuring fewer
    if torch.cuda.device_count() < 2:
        out = _bench("--gpus", "2", timeout=60)
        assert out.returncode == 2 and "refusing" in out.stderr


def test_bench_launcher_propagates_rank_failure():
    # one rank dies (bad backend name): the launcher stops the others and exits non-zero
    out = _bench("--gpus", "2", "--backend", "no_such_backend", "--launch-check", timeout=120)
    assert out.returncode != 0


def _dp_worker(rank, world, port, q):
    """Host logic of the data-parallel predictor on CPU: broadcast of rank 0's networks, the instance-weighted
    all-reduce of the gradients in front of the optimiser step (here the caller's own autograd loss through
    ``update_predictor``; the HIP gradient kernels take the same hook, tests/test_gpu_stats.py)."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from carle_amd import dist as cd
    from carle_amd.env import CARLE
    from carle_amd.mcl import RND2D, AE2D
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        total = 8
        start, count = (0, 3) if rank == 0 else (3, 5)                # unequal shards: weights 3/8 and 5/8
        torch.manual_seed(100 + rank)                                 # the ranks start from DIFFERENT networks
        ae = AE2D(RND2D(CARLE(width=64, height=64, instances=count, device="cpu")))
        rnd = ae.env
        ae.eval()                                                     # Dropout = identity, as in the HIP kernels
        enabled = cd.data_parallel(ae)
        assert enabled == [ae, rnd] and rnd._grad_sync[1] == count / total
        obs = (torch.rand(total, 1, 64, 64, generator=torch.Generator().manual_seed(7)) < 0.3).float()
        # a kernel gradient of a window in flight (what the HIP training kernels accumulate in w._grad): update_predictor
        # must leave it alone -- it is weighted and all-reduced ONCE, by the _apply_update that consumes it
        inflight = {}
        for w in (rnd, ae):
            w._grad = torch.full((7,), float(rank + 1))
            w._grad_used = True
            inflight[w.my_name] = w._grad
        for w in (rnd, ae):
            for _ in range(2):
                w.update_predictor(torch.mean(w._torch_loss(obs[start:start + count])))
        for w in (rnd, ae):
            assert w._grad is inflight[w.my_name] and torch.equal(w._grad, torch.full((7,), float(rank + 1)))
            w._sync_gradients(list(w.predictor.parameters()))          # as _apply_update does
            assert torch.allclose(w._grad, torch.full((7,), 1 * 3 / 8 + 2 * 5 / 8)), w._grad
            w._grad, w._grad_used = None, False
        q.put((rank, {k: v.numpy().copy() for k, v in ae.state_dict().items()}))
    finally:
        dist.destroy_process_group()


def test_data_parallel_predictor_two_ranks_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k]), k                # the ranks stay bit-identical
    # the unsharded run from rank 0's networks
    from carle_amd.env import CARLE
    from carle_amd.mcl import RND2D, AE2D
    torch.manual_seed(100)
    ae = AE2D(RND2D(CARLE(width=64, height=64, instances=8, device="cpu")))
    ae.eval()
    obs = (torch.rand(8, 1, 64, 64, generator=torch.Generator().manual_seed(7)) < 0.3).float()
    for w in (ae.env, ae):
        for _ in range(2):
            w.update_predictor(torch.mean(w._torch_loss(obs)))
    for k, v in ae.state_dict().items():
        # Adam's first steps are +-lr whatever the gradient's size: a sign error or a missing weight shows at 1e-1
        np.testing.assert_allclose(res[0][k], v.numpy(), rtol=0, atol=2e-5, err_msg=k)

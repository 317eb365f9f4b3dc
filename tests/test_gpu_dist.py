"""Batch sharding on the device: the reference's batch-global scalars -- SpeedDetector's speed over ALL
instances (carle/mcl.py:789), PufferDetector's live-cell total and action truthiness (mcl.py:832,835), the harness's
per-step reward total (train_mcl.py:68-69, eval.py:69-70), the master reset on the whole batch's action (env.py:208) --
under two ranks reproduce ONE process stepping the whole batch; and RCCL itself executes every exchange once, with one rank.

Two ranks share the ONE device of the test box through gloo (RCCL refuses two ranks per device), as in test_gpu_stats.py."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs(size, total, steps):
    g = torch.Generator().manual_seed(size * 131 + total)
    u0 = (torch.rand(total, 1, size, size, generator=g) < 0.08).float()
    acts = torch.zeros(steps, total, 1, 64, 64)
    acts[0] = (torch.rand(total, 1, 64, 64, generator=g) < 0.05).float()
    if steps > 7:
        acts[7, total - 1, 0, 5, 9] = 1.0      # ONE toggle, in the LAST instance (rank 1's shard): the whole batch "had an action"
    return u0, acts


def _make_stack(variant, size, count):
    from carle_amd.env import CARLE
    from carle_amd.mcl import RND2D, AE2D, SpeedDetector, PufferDetector
    torch.manual_seed(3)
    kw = {}
    if variant == "u8":
        kw = dict(storage="u8", obs="f32")       # the observation is NOT the inner universe: live cells from live4
    env = CARLE(width=size, height=size, instances=count, device="cuda:0", **kw)
    nets = AE2D(RND2D(env, train_predictor=False), train_predictor=False)
    if variant in ("pair", "u8"):
        w = PufferDetector(SpeedDetector(nets))              # eval.py:83-87: both exchange in ONE all-reduce
        w.env.reward_scale = 1e-2
    elif variant == "speed":
        w = SpeedDetector(nets)
    elif variant == "puffer":
        w = PufferDetector(nets)
    else:                                                    # "apart": a wrapper between the two -> two all-reduces
        from carle_amd.mcl import CornerBonus
        w = PufferDetector(CornerBonus(SpeedDetector(nets)))
    if isinstance(w, PufferDetector):
        w.growth_threshold = 4
    w.eval()
    w.reset()
    return env, w


def _trajectory(variant, size, start, count, total, steps):
    from carle_amd.mcl import PufferDetector, SpeedDetector
    u0, acts = _inputs(size, total, steps)
    env, w = _make_stack(variant, size, count)
    env.rules_from_string("B3/S012345678")                   # growing rule: the growth bonus fires
    env.universe = u0[start:start + count].cuda()
    rewards = []
    for t in range(steps):
        obs, reward, done, info = w.step(acts[t, start:start + count].cuda())
        rewards.append(reward.reshape(-1).cpu().numpy().copy())
    torch.cuda.synchronize()
    extra = {}
    node = w
    while getattr(node, "env", None) is not None:
        if type(node) is PufferDetector:
            extra["cells"], extra["live"] = node.cells, node.live_cells
        if type(node) is SpeedDetector:
            extra["speed"] = float(node.speed)
        node = node.env
    return np.stack(rewards), extra


def _traj_worker(rank, world, port, variant, size, total, steps, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from carle_amd import dist as cd
    cd.init_from_env("gloo", share_devices=True)
    try:
        start, count = cd.shard(total)
        q.put((rank, start, count) + _trajectory(variant, size, start, count, total, steps))
    finally:
        dist.destroy_process_group()


def _two_ranks(target, args, timeout=600):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, 2, port) + tuple(args) + (q,)) for r in range(2)]
    for p in procs:
        p.start()
    import queue
    import time
    res, t0 = [], time.time()
    while len(res) < 2:                                      # a rank that died must fail the test at once, not after a long wait
        try:
            res.append(q.get(timeout=2))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() - t0 > timeout:
                for p in procs:
                    if p.is_alive():
                        p.terminate()                        # exactly the children started above
                raise AssertionError(f"rank process failed (exit codes {[p.exitcode for p in procs]}) or timed out")
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return sorted(res, key=lambda r: r[0])


# (64 x 64 grids under the 64 x 64 window have no cell outside it: their centre of mass and speed are zero by construction)
@pytest.mark.parametrize("variant,size,total", [("pair", 128, 11), ("pair", 256, 5), ("pair", 64, 9), ("speed", 128, 7),
                                                ("puffer", 64, 7), ("apart", 128, 6), ("u8", 128, 5)])
def test_eval_stack_sharded_two_ways_reproduces_the_unsharded_rewards(variant, size, total):
    steps = 14
    res = _two_ranks(_traj_worker, (variant, size, total, steps))
    whole, extra = _trajectory(variant, size, 0, total, total, steps)
    sharded = np.concatenate([res[0][3], res[1][3]], axis=1)
    assert sharded.shape == whole.shape and res[0][2] + res[1][2] == total
    np.testing.assert_allclose(sharded, whole, rtol=1e-5, atol=1e-5)
    for r in res:                                            # every rank holds the WHOLE batch's scalars
        for k, v in extra.items():
            if k == "cells":
                assert r[4][k] == v, (k, r[4][k], v)         # live-cell totals are integers: exact
            else:
                np.testing.assert_allclose(r[4][k], v, rtol=1e-6)
    if "speed" in extra and size > 64:
        assert extra["speed"] > 1e-3                         # a real, batch-coupled speed went through the exchange
    if variant == "puffer":
        # (no SpeedDetector below: the reward is the nets' bonus, well under 0.9, plus the growth bonus of 1)
        assert whole[5].min() > 0.9 and whole[6].min() > 0.9     # the growth bonus did fire (history of 4 full at step 5) ...
        assert whole[8:12].max() < 0.9                           # ... and the lone toggle at step 7 (rank 1's shard) cleared it:
        assert sharded[8:12].max() < 0.9                         # on EVERY rank
        assert whole[12].min() > 0.9


def test_per_rank_scalars_opt_out_keeps_the_shards_apart():
    """dist.per_rank_scalars: no exchange -- a rank's SpeedDetector sees its own instances only."""
    res = _two_ranks(_per_rank_worker, ())
    assert res[0][1] != res[1][1]                            # different shards, different speeds
    for r in res:
        assert r[2] is True


def _per_rank_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from carle_amd import dist as cd
    from carle_amd.mcl import _scalar_group
    cd.init_from_env("gloo", share_devices=True)
    try:
        u0, acts = _inputs(128, 6, 4)
        start, count = cd.shard(6)
        env, w = _make_stack("speed", 128, count)
        assert _scalar_group(w) is not None
        cd.per_rank_scalars(w)
        opted_out = _scalar_group(w) is None
        env.universe = u0[start:start + count].cuda()
        for t in range(4):
            w.step(acts[t, start:start + count].cuda())
        q.put((rank, float(w.speed), opted_out))
    finally:
        dist.destroy_process_group()


class _SlicedAgent:
    """An agent of pre-drawn actions for the WHOLE batch that hands every rank the rows of its shard."""
    total, steps, start, count = 9, 6, 0, 9

    def __init__(self):
        g = torch.Generator().manual_seed(77)
        self.actions = (torch.rand(2 * self.steps, self.total, 1, 64, 64, generator=g) < 0.03).float()
        self.actions[1::2] = 0.0                             # every other step without an action
        self.t = 0

    def load_state_dict(self, sd):
        pass

    def __call__(self, obs):
        a = self.actions[self.t % len(self.actions), self.start:self.start + self.count].cuda()
        self.t += 1
        return a


def _evaluate(start, count):
    from carle_amd import harness
    from carle_amd.mcl import RND2D, AE2D, SpeedDetector, PufferDetector
    _SlicedAgent.start, _SlicedAgent.count = start, count
    torch.manual_seed(21)                                    # the wrappers' networks: the same on every rank
    wrappers = [[RND2D, 1.0, None], [AE2D, 1.0, None], [SpeedDetector, 1e-2, None], [PufferDetector, 1e-3, None]]
    rules = [[[3], [2, 3]], [[3, 6], [2, 3]]]
    score, trace = harness.evaluate(_SlicedAgent, rules, wrappers, steps=_SlicedAgent.steps, verbose=False,
                                    width=64, height=64, instances=count, device="cuda:0")
    return float(score), [float(x) for x in trace]


def _evaluate_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from carle_amd import dist as cd
    cd.init_from_env("gloo", share_devices=True)
    try:
        start, count = cd.shard(_SlicedAgent.total)
        q.put((rank,) + _evaluate(start, count))
    finally:
        dist.destroy_process_group()


def test_harness_evaluate_sharded_gives_the_unsharded_score():
    """evaluation/eval.py's loop on two shards: score and per-step trace of the WHOLE batch on every rank (the device
    ring of per-step moments, one all-reduce per rule set), equal to one process evaluating all instances."""
    res = _two_ranks(_evaluate_worker, ())
    score, trace = _evaluate(0, _SlicedAgent.total)
    assert len(trace) == 2 * _SlicedAgent.steps and score > 0
    for r in res:
        np.testing.assert_allclose(r[2], trace, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(r[1], score, rtol=1e-5)
    assert res[0][2] == res[1][2]                            # the same numbers on both ranks


def _reset_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from carle_amd import dist as cd
    from carle_amd.env import CARLE
    cd.init_from_env("gloo", share_devices=True)
    try:
        out = []
        for storage in ("f32", "u8", "b1"):
            for glob in (False, True):
                env = CARLE(width=64, height=64, instances=20, device="cuda:0", storage=storage,
                            global_action_predicates=glob, **({"obs": "f32"} if storage != "f32" else {}))
                env.reset()
                env.universe = torch.ones(20, 1, 64, 64, device="cuda")
                ones = torch.ones(20, 1, 64, 64, device="cuda")
                mixed = ones.clone()
                mixed[3, 0, 2, 2] = 0.0
                # step 1: rank 0 all ones, rank 1 not -> the WHOLE batch's action is not all ones
                obs, _, _, _ = env.step(ones if rank == 0 else mixed)
                pop1, n1 = float(obs.sum()), env.step_number
                # step 2: every rank all ones -> master reset everywhere
                obs, _, _, _ = env.step(ones)
                # step 3: an action on rank 1 only
                zero = torch.zeros(20, 1, 64, 64, device="cuda")
                env.step(mixed if rank == 1 else zero)
                out.append((storage, glob, pop1 > 0 or n1 == 1, n1, float(obs.sum()), env.master_resets,
                            env.steps_since_action))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_global_action_predicates_two_ranks():
    """CARLE(global_action_predicates=True): the master reset (env.py:208) fires iff EVERY rank's action is all ones and
    steps_since_action (env.py:191) sees an action on ANY rank; the default decides per rank."""
    res = _two_ranks(_reset_worker, ())
    for rank, rows in res:
        for storage, glob, _, n1, pop2, resets, since in rows:
            if glob:
                # step 1 was no reset anywhere (rank 1's action had a zero); step 2 was; step 3 had an action somewhere
                assert (n1, pop2, resets, since) == (1, 0.0, 1, 0), (rank, storage, rows)
            else:
                # per rank: rank 0 reset at step 1 AND step 2, rank 1 at step 2 only; rank 0 saw no action at step 3
                assert resets == (2 if rank == 0 else 1) and since == (1 if rank == 0 else 0), (rank, storage, rows)


def _child(mode):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dist_child.py"), "--mode", mode],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("DIST_CHILD ")]
    assert len(line) == 1, out.stdout[-2000:]
    return json.loads(line[0][len("DIST_CHILD "):])


def test_rccl_executes_with_one_rank():
    """The first (and, on a one-GPU box, only possible) execution of RCCL by this code: a fresh child process joins a
    process group of ONE rank with backend "nccl" and runs every exchange of carle_amd.dist through it -- the all-gather of
    the reward moments (all_gather_into_tensor, fp64 [W,3]), the wrappers' scalar all-reduce, the trace all-reduce, the flag
    all-reduce (MAX, int32), the parameter broadcast and the flat-gradient all-reduce in front of two Adam steps.  With one
    rank they are identities: every number must equal the same program run without a process group, bit for bit."""
    nccl = _child("nccl")
    plain = _child("plain")
    info = nccl.pop("info")
    plain.pop("info")
    assert info["collective_backend_initialised"] is True and info["backend"] == "nccl" and info["world_size"] == 1
    assert info["rank_device_map"] == [{"rank": 0, "local_rank": 0, "device": 0}]
    assert nccl.pop("mode") == "nccl" and plain.pop("mode") == "plain"
    assert nccl["dp_wrappers"] == 2 and nccl["train_updates"] == 2 and nccl["resets"] == 1
    assert nccl["reset_population"] == 0.0 and nccl["reset_step_number"] == 0
    assert set(nccl) == set(plain)
    for k in nccl:
        assert nccl[k] == plain[k], k
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "rccl_one_rank.json"), "w") as f:
        json.dump({"info": info, "result": nccl}, f, indent=1)

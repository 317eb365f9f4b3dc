"""Child process of tests/test_gpu_dist.py::test_rccl_executes_with_one_rank -- NOT collected by pytest.

Runs the multi-GPU exchanges of carle_amd.dist on ONE rank, either through a real process group of one rank
(``--mode nccl``: backend "nccl" = RCCL; every exchange is the identity there, but it is RCCL that executes it) or
without any group (``--mode plain``), and prints ONE JSON line with everything the exchanges touched.  The parent
compares the two lines: a one-rank all-gather / all-reduce / broadcast must not change a bit.

A fresh process on purpose: the parent never initialises a process group, and nothing here exec()s.
"""
import argparse
import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _digest(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["nccl", "plain"], required=True)
    args = ap.parse_args()
    from carle_amd import dist as cd
    from carle_amd import harness
    from carle_amd.env import CARLE
    from carle_amd.mcl import RND2D, AE2D, SpeedDetector, PufferDetector
    out = {"mode": args.mode}
    if args.mode == "nccl":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cd.force_collectives(True)
        r, w, dev = cd.init_from_env("nccl", single_rank_group=True)
        assert (r, w) == (0, 1) and cd.exchanging()
    else:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        assert not cd.exchanging()
    out["info"] = cd.collective_info(dev)

    # 1. reward moments: update -> flush = all_gather_into_tensor of the fp64 [W,3] triples + carle_merge_moments
    g = torch.Generator().manual_seed(1)
    stats = cd.RunningRewardStats(dev, sync_every=2)
    for t in range(5):
        stats.update((torch.randn(37, 1, generator=g) * (t + 1)).to(dev))
    stats.flush()
    out["stats_state"] = stats.state.cpu().tolist()
    out["stats_last"] = stats.last_batch.cpu().tolist()

    # 2. the evaluation stack (eval.py:83-87): SpeedDetector / PufferDetector exchange their batch-global scalars
    #    (all-reduce of fp64[4] per step), the trace ring is all-reduced by the drain
    torch.manual_seed(3)
    n = 5
    env = CARLE(width=64, height=64, instances=n, device=str(dev), global_action_predicates=True)
    w = PufferDetector(SpeedDetector(AE2D(RND2D(env, train_predictor=False), train_predictor=False)))
    w.env.reward_scale, w.reward_scale = 1e-2, 1e-3
    w.growth_threshold = 3
    w.eval()
    w.reset()
    env.universe = (torch.rand(n, 1, 64, 64, generator=g) < 0.3).float().to(dev)
    tr = cd.RunningRewardStats(dev, sync_every=1 << 30)
    tr.start_trace(w, 16)
    rewards = []
    for t in range(9):
        a = torch.zeros(n, 1, 64, 64)
        if t in (0, 5):
            a = (torch.rand(n, 1, 64, 64, generator=g) < 0.05).float()
        obs, reward, done, info = w.step(a.to(dev))
        tr.after_step(w, reward)
        rewards.append(reward.reshape(-1).cpu().tolist())
    out["eval_rewards"] = rewards
    out["eval_totals"] = tr.drain_trace().tolist()
    out["eval_cells"] = w.cells
    out["eval_speed"] = float(w.env.speed)
    # the master reset through the all-reduced flags
    obs, reward, done, info = env.step(torch.ones(n, 1, 64, 64, device=dev))
    out["reset_population"] = float(obs.sum())
    out["reset_step_number"] = env.step_number
    out["resets"] = env.master_resets

    # 3. data-parallel predictors: broadcast of rank 0's networks, all-reduce of the flat gradient before each Adam step
    torch.manual_seed(0)
    env = CARLE(width=64, height=64, instances=6, device=str(dev))
    ae = AE2D(RND2D(env))
    ae.reset()
    for wr in (ae, ae.env):
        wr.batch_size = 2
    out["dp_wrappers"] = len(cd.data_parallel(ae, broadcast=True))
    env.universe = (torch.rand(6, 1, 64, 64, generator=g) < 0.35).float().to(dev)
    tr_rewards = []
    for t in range(5):
        a = (torch.rand(6, 1, 64, 64, generator=g) < 0.1).float().to(dev)
        tr_rewards.append(ae.step(a)[1].reshape(-1).cpu().tolist())
    out["train_rewards"] = tr_rewards
    out["train_updates"] = int(ae.env.updates)
    out["train_sd"] = _digest(ae.state_dict())

    # 4. the harness loop (train_mcl.py): trace ring + drain
    from carle_amd.agents import RandomAgent
    torch.manual_seed(5)
    trace = harness.train(lambda **kw: RandomAgent(seed=11, **kw), instances=4, steps=(1, 5), rules=(((3,), (2, 3)),),
                          out_dir=None, verbose=False, width=64, height=64, device=str(dev))
    out["harness_trace"] = [float(x) for x in trace]

    torch.cuda.synchronize()
    if args.mode == "nccl":
        import torch.distributed as tdist
        tdist.barrier()
        tdist.destroy_process_group()
    print("DIST_CHILD " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

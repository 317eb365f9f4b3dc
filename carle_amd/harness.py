"""carle_amd.harness -- the reference's two harness loops on the HIP path.

``train`` follows carle/train_mcl.py:15-89 (reward-wrapper training under an agent, ruleset
cycling, state_dict + reward-trace dumps) and ``evaluate`` follows evaluation/eval.py:17-79
(stack wrappers, optional weights, score = mean summed reward).  Same signatures and return
values; differences: the per-step ``reward.sum().cpu()`` host round trips of the reference
(train_mcl.py:68-69, eval.py:69-70) are gone -- the kernel that finishes a step leaves the step's
reward moments in a device ring (``dist.RunningRewardStats.start_trace``: no launch, no allocation
per step where the step epilogue reduces the rewards; one moments launch otherwise) and the ring is
drained once per inner loop --, wrappers listed in ``mcl`` are actually used (the reference
hard-codes RND2D+AE2D), and environment kwargs can be passed through for small test geometries.

Batch sharding (one process per GPU, ``carle_amd.dist``; ``instances`` = this rank's shard): with a
process group up the per-step totals and the score are those of the WHOLE batch (one all-reduce of
the drained ring per inner loop), the predictors are trained data-parallel, and SpeedDetector /
PufferDetector exchange their batch-global scalars -- the sharded run reproduces the unsharded one.
"""
import os
import time

import numpy as np
import torch

from .env import CARLE
from .mcl import RND2D, AE2D


def _reward_trace(env, steps):
    """Per-step reward totals of one inner loop (train_mcl.py:68-69, eval.py:69-70) without a host round trip or a
    reduction launch per step, global under batch sharding: see ``RunningRewardStats.start_trace`` / ``drain_trace``."""
    from . import dist as cdist
    inner = env.inner_env if env.inner_env is not None else env
    stats = cdist.RunningRewardStats(inner.my_device, sync_every=1 << 30)      # flushed by the drain, once per loop
    stats.start_trace(env, max(int(steps), 1))
    return stats


def _data_parallel_or_warn(env):
    """Batch-sharded training (one process per GPU, carle_amd.dist): the reference's predictor loss is a mean over the
    WHOLE batch (carle/mcl.py:362,376).  With a process group up, the wrappers' predictors are trained data-parallel
    (``dist.data_parallel``: rank 0's networks broadcast once, the flat gradient all-reduced before every Adam step),
    so the sharded run takes the unsharded run's updates.  WORLD_SIZE > 1 WITHOUT a process group means every rank
    trains its own predictor on its shard: said once instead of silently."""
    import os
    import warnings
    import torch.distributed as tdist
    if tdist.is_available() and tdist.is_initialized():
        from . import dist as cdist
        cdist.data_parallel(env)
    elif int(os.environ.get("WORLD_SIZE", "1")) > 1:
        warnings.warn("carle_amd.harness.train under WORLD_SIZE > 1 without an initialised process group: every rank "
                      "trains its own RND2D / AE2D predictor on its shard of the batch (no gradient all-reduce); call "
                      "carle_amd.dist.init_from_env() first to train them data-parallel (DESIGN.md, multi-GPU)",
                      stacklevel=3)


def train(agent_fn, instances=16, steps=(64, 2048), rules=(((3,), (2, 3)),), mcl=(RND2D, AE2D),
          out_dir="./logs/mcl", verbose=True, **env_kwargs):
    env_kwargs.setdefault("use_cuda", True)
    env = CARLE(instances=instances, **env_kwargs)
    for wrapper in mcl:
        env = wrapper(env)
    _data_parallel_or_warn(env)
    agent = agent_fn(observation_width=env.inner_env.width, observation_height=env.inner_env.height,
                     action_width=env.inner_env.action_width, action_height=env.inner_env.action_height)
    exp_id = "mcl" + str(int(time.time()))
    rewards = []
    stats = _reward_trace(env, steps[1])
    for epoch in range(steps[0]):
        for ruleset in rules:
            env.inner_env.birth = list(ruleset[0])
            env.inner_env.survive = list(ruleset[1])
            obs = env.reset()
            t1 = time.time()
            for step in range(steps[1]):
                action = agent(obs)
                obs, reward, done, info = env.step(action)
                stats.after_step(env, reward)
            trace = stats.drain_trace().astype(np.float32)     # one host sync (and one all-reduce) per inner loop
            rewards.extend(list(trace))
            t2 = time.time()
            if verbose:
                print("steps / second = {:.3f}".format((steps[1] * instances) / max(t2 - t1, 1e-9)))
                print("round {}, ruleset {}, mean reward = {:.3e}".format(
                    epoch, ruleset, float(trace.sum()) / max(steps[1] * instances, 1)))
            if out_dir is not None:
                os.makedirs(os.path.join(out_dir, "models"), exist_ok=True)
                node = env
                while getattr(node, "env", None) is not None:      # every wrapper, outermost first
                    if hasattr(node, "my_name"):
                        torch.save(node.state_dict(),
                                   os.path.join(out_dir, "models", "{}_{}.pt".format(node.my_name, exp_id)))
                    node = node.env
        if out_dir is not None:
            os.makedirs(os.path.join(out_dir, "metrics"), exist_ok=True)
            np.save(os.path.join(out_dir, "metrics", "mcl_rewards_{}".format(exp_id)), rewards)
    return rewards


def evaluate(Agent, rules, wrappers, params_path=None, steps=1024, verbose=True, **env_kwargs):
    agent = Agent()
    if params_path is not None:
        agent.load_state_dict(torch.load(params_path))
    env = CARLE(**env_kwargs)
    for wrapper in wrappers:
        env = wrapper[0](env)
        env.reward_scale = wrapper[1]
        env.batch_size = steps * len(rules)
        if wrapper[2] is not None:
            env.load_state_dict(torch.load(wrapper[2], map_location="cpu"))
        env.eval()
    import torch.distributed as tdist
    if tdist.is_available() and tdist.is_initialized():
        from . import dist as cdist
        cdist.data_parallel(env)                     # batch sharding: rank 0's networks everywhere, one global update
    score, total_steps, score_trace = 0.0, 0, []
    stats = _reward_trace(env, steps)
    for ruleset in rules:
        env.inner_env.birth = list(ruleset[0])
        env.inner_env.survive = list(ruleset[0])     # sic: the reference assigns ruleset[0] twice (eval.py:59)
        obs = env.reset()
        for step in range(steps):
            action = agent(obs)
            obs, reward, done, info = env.step(action)
            stats.after_step(env, reward)
            total_steps += 1
        trace = stats.drain_trace().astype(np.float32)
        score += float(trace.sum())
        score_trace.extend(list(trace))
        if verbose:
            print("cumulative score = {:.3e} at total steps = {}, rulset = {}".format(score, total_steps, ruleset))
    return score / max(total_steps, 1), score_trace

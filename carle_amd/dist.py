"""carle_amd.dist -- batch sharding of CARLE over the GPUs of one node.

One process per GPU (``torch.distributed``, backend ``"nccl"`` = RCCL over xGMI on ROCm).
Universes are independent along the batch axis (carle/env.py:136: one conv channel per
instance), so rank r simply owns instances [r*N/W, (r+1)*N/W): the CA step, the action toggle
and the per-instance RND2D / AE2D bonuses need NO data-path collective.

The only exchange is the reward-normalisation running statistics (north star; absent from the
reference): each rank reduces its (N/W,1) rewards to a moment triple (count, mean, M2) on the
device (HIP kernel ``carle_batch_moments_f32``), the triples are all-gathered -- 24 bytes per
rank, latency-bound, one hop on the fully connected xGMI mesh -- and Chan-merged in rank order
on every rank, so all ranks hold bit-identical statistics.  The merged triple also yields the
batch-global scalars of the reference's harness (total reward, train_mcl.py:68-69).

Batch-global scalars of the reference under sharding:
  * ``SpeedDetector``'s speed = sqrt(sum over ALL instances of velocity^2) (mcl.py:789) and
    ``PufferDetector``'s live-cell total / "the action has a non-zero element" (mcl.py:832,835):
    with a process group up the wrappers exchange them -- one all-reduce(SUM) of four doubles per
    step (``batch_scalar_group``; ``per_rank_scalars`` opts a stack out);
  * the harness's per-step reward total (train_mcl.py:68-69, eval.py:69-70): the step kernels
    leave every step's reward moments in a device ring (``RunningRewardStats.start_trace``), one
    all-reduce of the ring per inner loop (``drain_trace``) makes the totals global;
  * the two whole-batch action predicates of CARLE.step (env.py:191 `if torch.sum(action)`,
    env.py:208 master reset on "every element of the whole action batch == 1"): evaluated per rank
    by default (the in-kernel decision of the one-launch step; synthetic agents never trigger the
    reset); ``CARLE(global_action_predicates=True)`` evaluates them on the action first and
    all-reduces the two flags (MAX) before the step kernel reads them (``all_reduce_flags``).
"""
import os

import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


_FORCE_COLLECTIVES = False


def force_collectives(on=True):
    """One-rank rehearsal of the RCCL path: with a process group of ONE rank up (``init_from_env(...,
    single_rank_group=True)``) every exchange of this module -- the all-gather of the reward moments, the all-reduce of
    the wrappers' batch-global scalars, of the trace totals, of the predictor gradients, the parameter broadcast --
    goes through the group although it is the identity there.  What a one-GPU box can execute of the multi-GPU path."""
    global _FORCE_COLLECTIVES
    _FORCE_COLLECTIVES = bool(on)


def exchanging(group=None):
    """True when the exchanges have somebody to talk to: more than one rank in ``group``, or a forced one-rank group."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or _FORCE_COLLECTIVES


def init_from_env(backend="nccl", share_devices=False, single_rank_group=False):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (torch.distributed.run,
    or bench.py's own launcher) and bind this process to its GPU.  Returns (rank, world, device).

    One process per GPU: LOCAL_RANK must name an existing device.  ``share_devices=True`` (gloo
    rehearsals of the multi-rank path on a node with fewer GPUs) wraps LOCAL_RANK around the
    device count instead of failing.  ``single_rank_group=True`` creates the group for WORLD_SIZE 1 too (see
    ``force_collectives``)."""
    w = int(os.environ.get("WORLD_SIZE", "1"))
    r = int(os.environ.get("RANK", "0"))
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    count = torch.cuda.device_count()            # does not initialise the GPU
    if count > 0 and lr >= count and not share_devices:
        raise RuntimeError(f"LOCAL_RANK={lr} but only {count} HIP device(s) are visible: one process "
                           f"per GPU (pass share_devices=True for a gloo rehearsal)")
    use_cuda = count > 0 and torch.cuda.is_available()
    dev = torch.device("cuda", lr % count) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(dev)
    if (w > 1 or single_rank_group) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = dev
        dist.init_process_group(backend, rank=r, world_size=w, **kw)
    return r, w, dev


def shard(total, r=None, w=None):
    """Contiguous block of instances owned by rank r: (start, count).  Sizes differ by at most
    one; the first ``total % w`` ranks take the extra instance."""
    r = rank() if r is None else r
    w = world() if w is None else w
    base, extra = divmod(int(total), w)
    count = base + (1 if r < extra else 0)
    start = r * base + min(r, extra)
    return start, count


def chan_merge(triples):
    """Merge (count, mean, M2) rows [W,3] in row order (Chan et al. pairwise update).  Pure
    tensor arithmetic on a handful of scalars: runs wherever the gathered tensor lives."""
    t = triples.to(torch.float64)
    n, mean, m2 = t[0, 0], t[0, 1], t[0, 2]
    for i in range(1, t.shape[0]):
        nb, mb, qb = t[i, 0], t[i, 1], t[i, 2]
        tot = n + nb
        safe = torch.where(tot > 0, tot, torch.ones_like(tot))
        delta = mb - mean
        mean = mean + delta * nb / safe
        m2 = m2 + qb + delta * delta * n * nb / safe
        n = tot
    return torch.stack([n, mean, m2])


class RunningRewardStats:
    """Running (count, mean, M2) of all rewards seen by ALL ranks.

    ``update(reward)`` costs: one single-block HIP kernel, one all-gather of 24 B/rank, a few
    scalar tensor ops.  No host synchronisation; ``mean``/``var``/``total`` read back lazily.
    """

    def __init__(self, device, sync_every=1):
        self.device = torch.device(device)
        self.state = torch.zeros(3, dtype=torch.float64, device=self.device)
        self.last_batch = torch.zeros(3, dtype=torch.float64, device=self.device)
        # sync_every > 1: fold the local triples of that many steps together before the
        # all-gather (a reporting interval): the exchange is latency-bound, so amortise it
        self.sync_every = max(1, int(sync_every))
        self._pending = torch.zeros(3, dtype=torch.float64, device=self.device)
        self._batch = torch.zeros(3, dtype=torch.float64, device=self.device)
        self._pending_steps = 0
        self._gathered = None
        self._trace = None                  # (cap, 3) ring of per-step batch triples (start_trace)
        self._trace_pos = 0

    # ---- per-step reward totals of a harness loop (train_mcl.py:68-69, eval.py:69-70) -----------------------------
    def start_trace(self, env, cap):
        """Keep the (count, mean, M2) triple of EVERY step of ``env`` in a device ring of ``cap`` rows instead of the
        last one only: where the step epilogue reduces the rewards itself (``attach``) the kernel that writes a step's
        moments takes the next row (CARLE_ST_TRACE_*: no launch per step, nothing to copy); otherwise ``after_step``
        points the stand-alone moments kernel at the row.  ``drain_trace`` turns the rows into per-step totals, global
        over the ranks.  Returns whether the epilogue feeds the ring."""
        self._trace = torch.zeros(max(int(cap), 2), 3, dtype=torch.float64, device=self.device)
        self._trace_pos = 0
        attached = self.attach(env)
        inner = self.__dict__.get("_inner_env")
        if inner is not None:
            inner._write_stats_pointers()
        return attached

    def stop_trace(self):
        self._trace = None
        self._trace_pos = 0
        inner = self.__dict__.get("_inner_env")
        if inner is not None:
            inner._write_stats_pointers()

    def drain_trace(self):
        """Per-step reward totals since ``start_trace`` / the last drain, oldest first, as a float64 numpy vector (at
        most ``cap``): sum over ALL ranks' instances -- ONE all-reduce of the vector per call, not one per step.  Also
        flushes the running statistics.  Synchronises (the only host round trip of an inner loop)."""
        if self._trace is None:
            raise RuntimeError("drain_trace() without start_trace()")
        inner = self.__dict__.get("_inner_env")
        if inner is not None:
            inner.settled_status()            # a one-launch step's moments are written by the next call on the block
        cap = self._trace.shape[0]
        k = min(self._trace_pos, cap)
        rows = self._trace[:k] if self._trace_pos <= cap else \
            torch.roll(self._trace, -(self._trace_pos % cap), 0)
        totals = (rows[:, 0] * rows[:, 1]).contiguous()
        if exchanging() and k > 0:
            dist.all_reduce(totals)
        self.flush()
        self._trace_pos = 0
        if inner is not None:
            inner._write_stats_pointers()     # the ring starts over
        return totals.cpu().numpy()

    def _trace_step(self, env, reward, stream):
        """One step in trace mode: nothing to do when the epilogue took a row, else the moments kernel writes it."""
        cap = self._trace.shape[0]
        if getattr(env, "_moments_merged", False):
            self._trace_pos += 1
            return
        row = self._trace[self._trace_pos % cap]
        if self.device.type == "cuda":
            from . import hip
            hip.batch_moments_f32(reward.reshape(-1), row, self._pending, stream)
        else:
            x = reward.reshape(-1).to(torch.float64)
            row.copy_(torch.stack([torch.tensor(float(x.numel()), dtype=torch.float64), x.mean() if x.numel() else x.sum(),
                                   ((x - x.mean()) ** 2).sum() if x.numel() else x.sum()]))
            self._pending.copy_(chan_merge(torch.stack([self._pending, row])))
        self._trace_pos += 1
        inner = self.__dict__.get("_inner_env")
        if inner is not None:                 # an attached environment whose step did not merge: keep the device index in step
            inner.settled_status()
            inner._write_stats_pointers()

    def local_moments(self, reward):
        from . import hip
        return hip.batch_moments_f32(reward.reshape(-1).contiguous())

    def update_from_local(self, local_triple):
        """All-gather the per-rank triples, merge them, fold into the running state."""
        local_triple = local_triple.to(self.device, torch.float64).reshape(1, 3)
        w = world()
        if exchanging():
            gathered = torch.empty(w, 3, dtype=torch.float64, device=self.device)
            dist.all_gather_into_tensor(gathered, local_triple)
        else:
            gathered = local_triple
        self.last_batch = chan_merge(gathered)
        self.state = chan_merge(torch.stack([self.state, self.last_batch]))
        return self.last_batch

    def update(self, reward, stream=None):
        """One HIP kernel per call: batch moments, merged on the device into the pending triple;
        every ``sync_every`` calls the pending triple is exchanged (``flush``).  ``stream``: raw
        hipStream_t to launch on (e.g. the reward stream of the wrappers)."""
        from . import hip
        hip.batch_moments_f32(reward.reshape(-1), self._batch, self._pending, stream)
        self._pending_steps += 1
        if self._pending_steps >= self.sync_every:
            return self.flush()
        return None

    def attach(self, env):
        """Let the step epilogue of ``env`` feed these statistics: the kernel that finishes a step
        (master reset, counters, reward vector) also reduces the step's rewards and merges them
        into the pending triple, so ``after_step`` has nothing to launch.  ``env`` must be the
        OUTERMOST environment the caller steps -- a bare ``CARLE`` (any storage), an ``RND2D``
        sitting directly on one (its bonus is part of the fused step), or the pair ``AE2D(RND2D(CARLE))``
        (tiled sizes: the pair's finish kernel leaves the partial sums; 64x64 forward passes: AE2D's bonus
        kernel does) -- because the epilogue only
        sees the rewards that exist when it runs.  Returns False (and changes nothing) otherwise;
        ``after_step`` then falls back to the stand-alone moments kernel."""
        inner = getattr(env, "inner_env", None) or env
        if not hasattr(inner, "_install_reward_stats") or self.device.type != "cuda":
            return False
        if env is not inner:
            from .mcl import RND2D, AE2D
            # ... or the pair AE2D(RND2D(CARLE)) at tiled sizes, whose rewards are final in the pair's finish kernel
            # (... or its forward-only form at 64x64, whose rewards are final in AE2D's bonus kernel)
            pair = type(env) is AE2D and (env._can_pair_step() != 0 or env._can_leave_moments())
            if not pair and (type(env) is not RND2D or env.env is not inner or env.train_predictor):
                return False
        inner._install_reward_stats(self)
        env.__dict__["_final_reward_stats"] = self
        self._inner_env = inner
        return True

    def after_step(self, env, reward, stream=None, steps=1):
        """Account for one finished step of ``env``: nothing to launch when its epilogue already
        merged the rewards (see ``attach``), else ``update(reward, stream)``.  ``steps``: a ``CARLE.run`` / ``multistep``
        call of that many steps whose launch merged every step's moments itself (one trace row per step)."""
        steps = max(1, int(steps)) if getattr(env, "_moments_merged", False) else 1
        if self._trace is not None:
            self._trace_step(env, reward, stream)
            self._trace_pos += steps - 1
            self._pending_steps += steps
            if self._pending_steps >= self.sync_every:
                return self.flush()
            return None
        if getattr(env, "_moments_merged", False):
            self._pending_steps += steps
            if self._pending_steps >= self.sync_every:
                return self.flush()
            return None
        return self.update(reward, stream)

    def flush(self):
        """Exchange whatever has accumulated since the last all-gather."""
        if self._pending_steps == 0:
            return None
        self._pending_steps = 0
        inner = self.__dict__.get("_inner_env")
        if inner is not None and inner.__dict__.get("_moments_pending"):
            inner.settled_status()               # moments a one-launch step left as partial sums: merged now
        if self.device.type != "cuda":           # CPU process groups (gloo tests): tensor ops
            local = self._pending.clone()
            self._pending.zero_()
            return self.update_from_local(local)
        from . import hip
        w = world()
        if exchanging():
            if self._gathered is None or self._gathered.shape[0] != w:
                self._gathered = torch.empty(w, 3, dtype=torch.float64, device=self.device)
            dist.all_gather_into_tensor(self._gathered, self._pending.reshape(1, 3))
            src = self._gathered
        else:
            src = self._pending
        hip.merge_moments(src, w, self.state, self.last_batch, self._pending)
        return self.last_batch

    def normalize(self, reward, eps=1e-8):
        from . import hip
        flat = reward.reshape(-1).contiguous()
        return hip.normalize_f32(flat, self.state, eps).reshape(reward.shape)

    # lazy host views
    @property
    def count(self):
        return float(self.state[0].item())

    @property
    def mean(self):
        return float(self.state[1].item())

    @property
    def var(self):
        c = self.count
        return float(self.state[2].item()) / c if c > 0 else 0.0

    @property
    def last_total(self):
        """Global sum of the rewards of the last update (= reward.sum() of the unsharded run)."""
        return float((self.last_batch[0] * self.last_batch[1]).item())


def global_all_ones(local_not_all_ones_flag):
    """Exact whole-batch master-reset predicate under sharding: 1-int all-reduce (MAX) of the
    per-rank "some element != 1" flag; returns a tensor that is 0 iff EVERY rank's action is all
    ones."""
    t = local_not_all_ones_flag.clone()
    if exchanging():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


def all_reduce_flags(status, slot, group=None):
    """The two whole-batch action predicates of CARLE.step (env.py:191, 208) made global: ANY_NONZERO(slot) and
    NOT_ALL_ONES(slot) of the status block are OR-ed over the ranks (all-reduce MAX of two int32) -- the master reset
    then fires iff EVERY rank's action is all ones, ``steps_since_action`` counts steps without an action on ANY rank.
    The flags must already be in the block (carle_action_flags_f32 / the pack kernels), the step kernel must read them
    from there (publish = False)."""
    flags = status[slot:slot + 3:2]
    t = flags.clone()
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    flags.copy_(t)


def batch_scalar_group(wrapper):
    """-> ``(group,)`` when the batch-global scalars of a hand-coded wrapper (SpeedDetector, PufferDetector) have to be
    exchanged -- a process group of more than one rank is up and the wrapper was not opted out -- else None.  ``group``
    None = the default group; ``wrapper._scalar_group`` names another one."""
    d = wrapper.__dict__
    if d.get("batch_scalars", "global") == "per_rank":
        return None
    if not (dist.is_available() and dist.is_initialized()):
        return None
    g = d.get("_scalar_group")
    return (g,) if exchanging(g) else None


def per_rank_scalars(env, on=True):
    """Opt the hand-coded wrappers of a stack out of (or back into) the exchange of their batch-global scalars: with
    ``on`` every rank's SpeedDetector / PufferDetector sees its own shard only (speed over the shard's instances, growth
    of the shard's live cells, the shard's action) -- NOT the unsharded run's rewards; for runs whose ranks are
    independent experiments rather than shards of one batch."""
    cur = env
    while cur is not None:
        if hasattr(cur, "inner_env") and cur.inner_env is not None:
            cur.__dict__["batch_scalars"] = "per_rank" if on else "global"
        cur = getattr(cur, "env", None)
    return env


def collective_info(dev=None):
    """What a bench / test record needs to prove N ranks: backend, whether the group is up, the RCCL version torch was
    built against, and every rank's (rank, local_rank, device index) gathered through the group itself."""
    info = {"collective_backend_initialised": bool(dist.is_available() and dist.is_initialized()),
            "backend": dist.get_backend() if dist.is_available() and dist.is_initialized() else None,
            "world_size": world()}
    try:
        v = torch.cuda.nccl.version()
        info["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception as e:                       # torch built without it: say so
        info["rccl_version"] = f"unavailable ({type(e).__name__})"
    mine = [rank(), int(os.environ.get("LOCAL_RANK", "0")),
            dev.index if dev is not None and dev.type == "cuda" and dev.index is not None else -1]
    if exchanging():
        on_device = dist.get_backend() == "nccl"
        t = torch.tensor(mine, dtype=torch.int64, device=dev if on_device else "cpu")
        rows = torch.empty(world() * 3, dtype=torch.int64, device=t.device)
        dist.all_gather_into_tensor(rows, t)
        rows = rows.reshape(-1, 3).tolist()
    else:
        rows = [mine]
    info["rank_device_map"] = [{"rank": r[0], "local_rank": r[1], "device": r[2]} for r in rows]
    return info


def data_parallel(env, group=None, broadcast=True):
    """Train the predictors of the reward wrappers around ``env`` (RND2D, AE2D; any nesting) data-parallel over the
    ranks of ``group`` -- the one place where batch sharding changes the reference's arithmetic: its loss is the mean
    over the WHOLE batch (carle/mcl.py:362,376), so the sharded gradient is sum_r (n_r / N) * g_r.

    ``broadcast``: rank 0's predictor and random-network parameters are sent to every rank first (the ranks must
    start from the same networks for the sharded run to be the unsharded one; skip it when they were built from
    the same seed).  Afterwards every Adam step of those wrappers all-reduces its flat gradient (4.5 KB at 64x64,
    66 KB at 256x256: latency-bound, one RCCL call per ``batch_size`` steps) on the stream the gradient kernels ran
    on.  Returns the wrappers it enabled.  A no-op list when there is one rank."""
    from .mcl import RND2D
    wrappers, cur = [], env
    while cur is not None:
        if isinstance(cur, RND2D):
            wrappers.append(cur)
        cur = getattr(cur, "env", None)
    if not exchanging(group) or not wrappers:
        return wrappers
    inner = wrappers[0].inner_env
    on_device = dist.get_backend(group) == "nccl"
    cdev = torch.device(inner.my_device) if on_device else torch.device("cpu")
    total = torch.tensor([float(inner.instances)], dtype=torch.float64, device=cdev)
    dist.all_reduce(total, group=group)
    weight = float(inner.instances) / float(total.item())
    src = dist.get_global_rank(group, 0) if group is not None else 0
    for wr in wrappers:
        if broadcast:
            nets = [wr.predictor] + ([wr.random_network] if getattr(wr, "random_network", None) is not None else [])
            params = [p for net in nets for p in net.parameters()]
            flat = torch.cat([p.detach().reshape(-1).to(cdev, torch.float32) for p in params])
            dist.broadcast(flat, src=src, group=group)
            off = 0
            with torch.no_grad():
                for p in params:
                    p.copy_(flat[off:off + p.numel()].view_as(p))        # in place: bumps the version the blobs watch
                    off += p.numel()
        wr.__dict__["_grad_sync"] = (group, weight)
    return wrappers

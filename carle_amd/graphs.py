"""HIP-graph replay of ``env.step(action)`` (build-side extension; no reference counterpart).

The reference's loop (carle/train_mcl.py:64-69, evaluation/eval.py:63-70) calls ``env.step`` once per
generation.  For the reference's own default geometry -- 16 instances of 256x256, RND2D + AE2D on top -- a step
is seven kernel launches and ~60 us of Python and launch overhead around ~35 us of GPU work; at N=1 the GPU work
is 3 us.  ``StepGraph`` captures everything one ``env.step(action)`` enqueues -- CA step, epilogue, every reward
wrapper's kernels -- into a HIP graph (``torch.cuda.CUDAGraph``: hipStreamBeginCapture / hipGraphLaunch on
ROCm) and replays it with one launch per step.

Contract (the usual one of graph replay):
  * ``action`` is a STATIC device tensor: write each step's action into it in place (``action.copy_(a)`` or
    have the agent write there) before calling the graph.  A ``RandomAgent(fused=True)`` action (toggles drawn in
    the step kernel) cannot be captured: its Philox offset is a kernel argument.
  * The environment's state ping-pongs between two fixed buffers; the returned ``obs`` / ``reward`` tensors are
    the same objects every second call -- consume (or clone) them before the step after next.  (The reference
    returns a fresh tensor per step, env.py:229.)
  * Replay repeats the captured launches; Python-side bookkeeping of wrappers does not run.  Valid for ``CARLE``
    (all storages) and for reward wrappers that only enqueue kernels per step: RND2D / AE2D without predictor
    updates (``train_predictor=False``), CornerBonus, ParsimonyBonus, SpeedDetector and PufferDetector (their
    per-step history lives in fixed device buffers) -- the whole evaluation stack of evaluation/eval.py:83-87.
    Predictor training and RLE logging are refused.
  * Between calls the caller may assign ``env.universe`` (patterns), call ``env.reset()`` or take eager steps:
    the graph copies a foreign state into its buffer before replaying.  Changing the rules re-captures.
  * Build it after ``env.reset()``: construction takes two eager warm-up steps and two capture steps with the
    action buffer's current contents, then restores the CA state, the status block, the step parity, the device
    state of the wrappers (``_graph_state()``: centre of mass, growth history) and the moment accumulators of an
    attached ``RunningRewardStats``.
  * The lookup tables of the reward nets are frozen into the capture ("tables valid"): when a predictor / random
    network parameter changes afterwards (``load_state_dict``, an optimiser step) the next call re-captures.

``backend="calls"`` replays the step WITHOUT a HIP graph: the library calls one ``env.step`` makes (their ctypes
functions and argument tuples) are recorded once per step parity and issued again -- a handful of ctypes calls per step,
none of the Python of the wrappers.  hipGraphLaunch costs ~10 us per replay on this stack and loses against eager steps
for most stacks (tools/graph_time.py); a recorded call list costs ~2 us per library call.  Same contract as above, plus:
a step may only reach the device through the library (the capture verifies it: the replayed step must reproduce the
eager step's observation and reward bit for bit, else the constructor raises), and the replays are enqueued on the
stream that was current at capture.  ``plan(action)`` may pass a different action tensor of the captured shape per call
(its pointer replaces the captured one): no copy into a static buffer.
"""
import torch

from . import hip
from . import _lib

# entry points that launch nothing (sizes, versions, errors): called during a step, not part of its replay
_QUERIES = ("_floats", "_doubles", "carle_abi_version", "carle_last_error", "carle_build_arch", "carle_reload_knobs")


# wrappers whose step reaches the device through the library only (backend="calls"; the capture also verifies it)
_LIBRARY_ONLY_WRAPPERS = ("RND2D", "AE2D", "SpeedDetector", "PufferDetector")


class _CallRecorder:
    """While active, every launching entry point of the library records (function, arguments) before it runs."""

    def __init__(self):
        self.calls = []

    def __enter__(self):
        lib = self._lib = _lib.load()
        self._saved = {}
        for name in _lib.SIGNATURES:
            fn = getattr(lib, name)
            self._saved[name] = fn
            if name.endswith(_QUERIES[:2]) or name in _QUERIES:
                continue

            def proxy(*args, _fn=fn):
                self.calls.append((_fn, args))
                return _fn(*args)
            lib.__dict__[name] = proxy
        return self

    def __exit__(self, *exc):
        for name, fn in self._saved.items():
            self._lib.__dict__[name] = fn
        return False


class StepGraph:
    check_params = True        # compare the reward nets' parameter versions every call (a few us per wrapper)

    def __init__(self, env, action, backend="graph"):
        hip.require_gpu()
        if backend not in ("graph", "calls"):
            raise ValueError("backend must be 'graph' or 'calls'")
        self.backend = backend
        base = env if getattr(env, "inner_env", None) is None else env.inner_env
        if getattr(action, "materialize", None) is not None:
            raise TypeError("a fused RandomAgent action cannot be captured (its Philox offset is a kernel "
                            "argument); materialise the action into a static tensor")
        if not isinstance(action, torch.Tensor) or not action.is_cuda:
            raise TypeError("action must be a static tensor on the environment's device")
        if base.logging:
            raise NotImplementedError("RLE logging runs on the host every step; not capturable")
        if getattr(base, "_share_predicates", None) is not None and base._share_predicates() is not None:
            raise NotImplementedError("global_action_predicates all-reduces two flags over the process group every step: a "
                                      "collective is not part of a replayed step")
        w = env
        while w is not base:
            if getattr(w, "train_predictor", False):
                raise NotImplementedError(f"{type(w).__name__} trains its predictor (window bookkeeping and Adam "
                                          "run on the host): set train_predictor=False to replay the forward pass")
            if getattr(w, "async_reward", False):
                raise NotImplementedError("async_reward=True forks to a second stream per step; capture the "
                                          "same-stream form")
            if hasattr(w, "grid_buffer"):
                raise NotImplementedError(f"{type(w).__name__} rotates its frame buffer on the host every step (other "
                                          "frames, other pointers): its step cannot be replayed")
            if type(w).__name__ in ("SpeedDetector", "PufferDetector"):
                from .mcl import _scalar_group
                if _scalar_group(w) is not None:
                    raise NotImplementedError(f"{type(w).__name__} exchanges its batch-global scalars over the process group "
                                              "every step (batch sharding): a collective is not part of a replayed step")
            if backend == "calls" and type(w).__name__ not in _LIBRARY_ONLY_WRAPPERS:
                raise NotImplementedError(f"{type(w).__name__}.step also launches torch kernels: its step is not made of "
                                          "library calls alone -- use backend='graph'")
            w = w.env
        self.env, self.base, self.action = env, base, action
        self._graphs = None
        self._capture()

    def _wrappers(self):
        w = self.env
        while w is not self.base:
            yield w
            w = w.env

    def _param_key(self):
        """Versions of every reward-net parameter the captured kernels' tables were built from."""
        key = []
        for w in self._wrappers():
            for name in ("_pred_blob", "_rand_blob"):
                blob = getattr(w, name, None)
                if blob is not None:
                    ps = blob.params if blob.params is not None else list(blob.module.parameters())
                    key.extend((p.data_ptr(), p._version) for p in ps)
        return tuple(key)

    def _side_state(self):
        """Device tensors outside the CA state that the warm-up / capture steps modify."""
        ts = []
        for w in self._wrappers():
            gs = getattr(w, "_graph_state", None)
            if gs is not None:
                ts.extend(gs())
        stats = self.base.__dict__.get("_reward_stats")
        if stats is not None:
            ts.extend([stats._batch, stats._pending])
        return ts

    # ------------------------------------------------------------------ capture
    def _capture(self):
        env, base = self.env, self.base
        state = base._raw_state()                      # raises before reset()
        # a one-launch step before us may have left its counters PENDING and its reward moments as per-workgroup
        # partial sums registered in the status block (words 20..23): settle them NOW, or the warm-up steps would merge
        # them into accumulators that the restore below rolls back, while the restored status block would point at
        # partial buffers that by then hold the capture steps' rewards
        base.settled_status()
        keep_state, keep_status, keep_slot = state.clone(), base._status.clone(), base._slot
        side_before = self._side_state()
        side_keep = {id(t): t.clone() for t in side_before}
        self._bufs = [torch.empty_like(state), torch.empty_like(state)]
        self.__dict__["_mem_pools"] = []                # (a re-capture releases the previous recording's pools)
        self._rule = base.rule_mask()
        # warm-up on a side stream: lazy initialisation (tables, workspaces, parameter blobs) must not be captured
        side = torch.cuda.Stream(device=state.device)
        side.wait_stream(torch.cuda.current_stream(state.device))
        with torch.cuda.stream(side):
            env.step(self.action)
            env.step(self.action)
        torch.cuda.current_stream(state.device).wait_stream(side)
        graphs = []
        pool = None
        for p in (0, 1):
            self._bufs[p].copy_(keep_state)
            base._set_raw_state(self._bufs[p])
            base._slot = p
            base.__dict__["_out_override"] = self._bufs[p ^ 1]
            if self.backend == "graph":
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool):
                    result = env.step(self.action)
                pool = g.pool()
            else:
                # every tensor the step allocates (bonus vectors, sums, the reward) comes from a private memory pool that
                # lives as long as the recording: the recorded pointers stay valid and nobody else is handed the blocks
                mpool = torch.cuda.MemPool()
                self.__dict__.setdefault("_mem_pools", []).append(mpool)
                with torch.cuda.use_mem_pool(mpool, device=state.device), _CallRecorder() as rec:
                    result = env.step(self.action)
                aptr = self.action.data_ptr()          # (function, arguments, positions of the action pointer)
                g = [(fn, args, tuple(i for i, a in enumerate(args) if a == aptr)) for fn, args in rec.calls]
            if base.__dict__.pop("_out_override", None) is not None or base._raw_state() is not self._bufs[p ^ 1]:
                raise RuntimeError("the captured step did not write the graph's state buffer")
            graphs.append((g, result, base.__dict__.get("_moments_merged", False),
                           env.__dict__.get("_moments_merged", False), base.__dict__.get("_moments_pending", False)))
        self._graphs = graphs
        self._action_ptr = self.action.data_ptr()
        # the recorded calls carry the raw handle of the stream that was current at capture
        self._stream = hip._stream(state)
        self._params = self._param_key()
        # back to where the caller was: wrapper state that existed before is restored, state the warm-up created
        # (lazily allocated histories) starts empty
        for t in self._side_state():
            if id(t) in side_keep:
                t.copy_(side_keep[id(t)])
            else:
                t.zero_()
        base._status.copy_(keep_status)
        base._slot = keep_slot
        self._bufs[keep_slot].copy_(keep_state)
        base._set_raw_state(self._bufs[keep_slot])
        if self.backend == "calls":
            self._verify_calls(keep_state, keep_status, keep_slot, side_keep)

    def _verify_calls(self, keep_state, keep_status, keep_slot, side_keep):
        """The recorded calls must BE the step: an eager step and a replayed one from the same state agree bit for bit in
        observation and reward (a wrapper that also launches torch kernels in its step would not)."""
        env, base = self.env, self.base

        def restore():
            for t in self._side_state():
                if id(t) in side_keep:
                    t.copy_(side_keep[id(t)])
                else:
                    t.zero_()
            base._status.copy_(keep_status)
            base._slot = keep_slot
            self._bufs[keep_slot].copy_(keep_state)
            base._set_raw_state(self._bufs[keep_slot])

        # ... with an action that toggles something: the action must reach the library as it is (a storage whose step
        # converts the action with torch operations first would replay the conversion's stale result)
        keep_action = self.action.clone()
        gen = torch.Generator(device=self.action.device).manual_seed(7)
        self.action.copy_((torch.rand(self.action.shape, device=self.action.device, generator=gen) < 0.3).to(self.action.dtype))
        base.__dict__["_out_override"] = self._bufs[keep_slot ^ 1]
        obs, reward = env.step(self.action)[:2]
        base.__dict__.pop("_out_override", None)
        want = (obs.clone(), reward.clone())
        restore()
        obs, reward = self._replay(keep_slot, self._action_ptr)[:2]
        same = torch.equal(obs, want[0]) and torch.equal(reward, want[1])
        restore()
        self.action.copy_(keep_action)
        if not same:
            raise NotImplementedError("this step is not made of library calls alone (a wrapper or the action / observation "
                                      "conversion of this storage launches torch kernels per step): use backend='graph'")

    # ------------------------------------------------------------------ replay
    def __call__(self, action=None):
        """One ``env.step(action)`` with the action buffer's current contents; returns the captured call's
        ``(obs, reward, done, info)`` objects.  backend="calls": ``action`` may be another device tensor of the captured
        shape and dtype (used in place, no copy)."""
        base = self.base
        if base.rule_mask() != self._rule or (self.check_params and self._param_key() != self._params):
            self._capture()
        p = base._slot
        buf = self._bufs[p]
        if self.backend == "calls" and hip._stream(buf) != self._stream:
            # the recorded launches go to the capture stream; the state copy below and the caller's action producer run
            # on the CURRENT stream: replaying under another stream context would race
            raise RuntimeError("StepGraph(backend='calls') replays on the stream that was current at capture: call it "
                               "under that stream (or build the StepGraph under the stream it will be used on)")
        state = base._raw_state()
        if state is not buf:                            # reset(), an assigned pattern, eager steps in between
            buf.copy_(state)
        if action is None or action is self.action:
            act, ptr = self.action, self._action_ptr
        else:
            if self.backend != "calls":
                raise TypeError("a HIP graph replays its static action buffer: copy the action into it")
            if action.shape != self.action.shape or action.dtype != self.action.dtype or action.device != self.action.device \
                    or not action.is_contiguous():
                raise ValueError("action must match the captured action tensor (shape, dtype, device, contiguous)")
            act, ptr = action, action.data_ptr()
        result = self._replay(p, ptr)
        base.action = act
        return result

    def _replay(self, p, action_ptr):
        base = self.base
        g, result, merged, outer_merged, pending = self._graphs[p]
        if self.backend == "graph":
            g.replay()
        else:
            swap = action_ptr != self._action_ptr
            # eager steps make the environment's device current for each launch; so does the replay (an environment on
            # cuda:1 replayed while cuda:0 is current would launch on a foreign device's stream handle)
            with hip._on_device(self._bufs[0].device):
                for fn, args, where in g:
                    if swap and where:
                        args = list(args)
                        for i in where:
                            args[i] = action_ptr
                    rc = fn(*args)
                    if rc:
                        _lib.check(rc, getattr(fn, "__name__", "library call"))
        base._set_raw_state(self._bufs[p ^ 1])
        base._slot = p ^ 1
        base._moments_merged = merged
        if self.env is not base:
            self.env.__dict__["_moments_merged"] = outer_merged
        if pending:
            base.__dict__["_moments_pending"] = True
        return result

    step = __call__

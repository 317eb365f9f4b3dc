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
"""
import torch

from . import hip


class StepGraph:
    check_params = True        # compare the reward nets' parameter versions every call (a few us per wrapper)

    def __init__(self, env, action):
        hip.require_gpu()
        base = env if getattr(env, "inner_env", None) is None else env.inner_env
        if getattr(action, "materialize", None) is not None:
            raise TypeError("a fused RandomAgent action cannot be captured (its Philox offset is a kernel "
                            "argument); materialise the action into a static tensor")
        if not isinstance(action, torch.Tensor) or not action.is_cuda:
            raise TypeError("action must be a static tensor on the environment's device")
        if base.logging:
            raise NotImplementedError("RLE logging runs on the host every step; not capturable")
        w = env
        while w is not base:
            if getattr(w, "train_predictor", False):
                raise NotImplementedError(f"{type(w).__name__} trains its predictor (window bookkeeping and Adam "
                                          "run on the host): set train_predictor=False to replay the forward pass")
            if getattr(w, "async_reward", False):
                raise NotImplementedError("async_reward=True forks to a second stream per step; capture the "
                                          "same-stream form")
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
        keep_state, keep_status, keep_slot = state.clone(), base._status.clone(), base._slot
        side_before = self._side_state()
        side_keep = {id(t): t.clone() for t in side_before}
        self._bufs = [torch.empty_like(state), torch.empty_like(state)]
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
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool):
                result = env.step(self.action)
            if base.__dict__.pop("_out_override", None) is not None or base._raw_state() is not self._bufs[p ^ 1]:
                raise RuntimeError("the captured step did not write the graph's state buffer")
            pool = g.pool()
            graphs.append((g, result, base.__dict__.get("_moments_merged", False)))
        self._graphs = graphs
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

    # ------------------------------------------------------------------ replay
    def __call__(self):
        """One ``env.step(action)`` with the action buffer's current contents; returns the captured call's
        ``(obs, reward, done, info)`` objects."""
        base = self.base
        if base.rule_mask() != self._rule or (self.check_params and self._param_key() != self._params):
            self._capture()
        p = base._slot
        buf = self._bufs[p]
        state = base._raw_state()
        if state is not buf:                            # reset(), an assigned pattern, eager steps in between
            buf.copy_(state)
        g, result, merged = self._graphs[p]
        g.replay()
        base._set_raw_state(self._bufs[p ^ 1])
        base._slot = p ^ 1
        base.action = self.action
        base._moments_merged = merged
        return result

    step = __call__

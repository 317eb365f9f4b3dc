"""carle_amd.env -- the CARLE environment surface on top of the HIP hot path.

Mirrors the public surface of the reference's ``carle.env.CARLE`` (riveSunder/carle,
carle/env.py:15-242) so that train/eval loops written against it run unchanged:
``CARLE(**kwargs)``, ``reset() -> obs``, ``step(action) -> (obs, reward, done, info)``, the
``universe`` / ``birth`` / ``survive`` attributes, the rule-string setters and the
``neighborhood.weight`` parameter that appears in every wrapper ``state_dict``.

What is different underneath:
  * the ~19 ATen ops / 15 memory passes of one reference step are ONE HIP kernel
    (``carle_ca_step_f32``) plus a tiny epilogue kernel (``carle_ca_finish``);
  * the reference's two host synchronisations per step (``if torch.sum(action)``, env.py:191,
    and ``torch.mean(action) == 1.0``, env.py:208) are evaluated on the device: the step kernel
    publishes two flags, the epilogue applies the master reset and keeps ``step_number`` /
    ``steps_since_action`` in device memory; reading those attributes synchronises lazily;
  * this build is GPU-only: without a HIP device ``reset``/``step`` raise (no CPU fallback).
"""
import os
import time
import weakref

import numpy as np
import torch
import torch.nn as nn

from . import hip
from .rle import RLEMixin

_MOORE = ((1.0, 1.0, 1.0), (1.0, 0.0, 1.0), (1.0, 1.0, 1.0))
# CARLE_PACKED_STATE=0: the 64x64 step always reads the fp32 state, as up to round 4 (A/B switch; read once at import).
# Default: a step reads the bit-packed copy the previous step left of its own output -- valid only for the very tensor that
# step returned, untouched by torch since (``CARLE._packed_state_of``); anything else reads the fp32 cells.
_PACKED_STATE = os.environ.get("CARLE_PACKED_STATE", "1") != "0"


def _resolve_device(kwargs):
    """Device choice of carle/env.py:25-37, widened to cuda:0..7 and defaulting to the GPU.

    The reference defaults to the CPU; this implementation has no CPU compute path, so with no
    device argument (or use_cuda=True) the first HIP device is used.  An explicit "cpu" is kept
    as-is: construction, rule parsing and state_dict handling work, compute calls raise.
    """
    dev = kwargs.get("device", None)
    if dev is not None:
        dev = torch.device(dev)
        if dev.type == "cuda" and dev.index is None:     # "cuda" -> the current device, indexed:
            dev = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
        return dev                                       # tensor.device comparisons need the index
    if kwargs.get("use_cuda", False) or torch.cuda.is_available():
        return torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
    return torch.device("cpu")


# attributes rewritten every step: bypass nn.Module.__setattr__ (parameter/buffer/module
# bookkeeping, ~3 us each) -- they are plain Python / tensor attributes
_PLAIN_ATTRS = frozenset(("universe", "action", "_slot", "log", "_moments_merged"))


def _done_zeros(n):
    """The fresh CPU ``done`` tensor of a step, zeros (N,1) fp32 (env.py:237).  Built from a calloc'ed numpy array:
    ``torch.zeros`` above 32768 elements is a parallel fill that wakes torch's whole OpenMP pool (one thread per
    visible core -- 256 on a 16-CPU GPU box); the spinning workers exhaust the container's CPU quota and the kernel
    freezes the process for the rest of the 100 ms period (measured: 92 ms stalls every few dozen steps at N=65536,
    ``nr_throttled`` in /sys/fs/cgroup/cpu.stat; none with this form)."""
    return torch.from_numpy(np.zeros((n, 1), dtype=np.float32))


class CARLE(RLEMixin, nn.Module):
    """Batched toroidal Life-like cellular automaton with a gym-style interface."""

    def __new__(cls, **kwargs):
        # storage="b1" / "u8" select the bit-packed / uint8 state variants (same surface, 1 bit or
        # 1 byte per cell in HBM instead of 4 bytes)
        if cls is CARLE and kwargs.get("storage", "f32") == "b1":
            return super().__new__(CARLEPacked)
        if cls is CARLE and kwargs.get("storage", "f32") == "u8":
            return super().__new__(CARLEBytes)
        return super().__new__(cls)

    def __setattr__(self, name, value):
        if name in _PLAIN_ATTRS:
            object.__setattr__(self, name, value)
        else:
            super().__setattr__(name, value)

    def __init__(self, **kwargs):
        super().__init__()
        self.inner_env = None                                   # carle/env.py:20
        self.width = kwargs.get("width", 256)
        self.height = kwargs.get("height", 256)
        self.action_width = kwargs.get("action_width", 64)
        self.action_height = kwargs.get("action_height", 64)
        self.use_cuda = kwargs.get("use_cuda", False)
        self.my_device = _resolve_device(kwargs)
        self.use_grad = kwargs.get("use_grad", False)
        self.alive_rate = kwargs.get("alive_rate", 0.0)         # stored, never used (env.py:43)
        self.instances = kwargs.get("instances", 1)
        self.logging = kwargs.get("logging", False)
        # reference behaviour for an empty birth/survive list is a TypeError (env.py:221-224);
        # by default this build treats an empty list as "never" (a superset); strict mirrors it
        self.strict_reference_errors = kwargs.get("strict_reference_errors", False)
        # the two whole-batch action predicates of step() (env.py:191 `if torch.sum(action)`,
        # env.py:208 `torch.mean(action) == 1.0`).  Default: evaluated inside the step kernel as
        # "some element is non-zero" / "every element equals 1.0" -- identical for {0,1}-valued
        # actions (what every agent of the reference emits).  True: evaluated exactly as the
        # reference does, with torch.sum / torch.mean over the action as passed (two extra device
        # reductions per step, still no host synchronisation): actions whose values cancel, or
        # average to exactly 1.0 without all being 1, then behave as in the reference.
        self.exact_action_predicates = bool(kwargs.get("exact_action_predicates", False))
        # batch sharding (carle_amd.dist): the two predicates are defined on the WHOLE batch's action.  Default: each
        # rank decides from its shard inside its one-launch step.  True: the flags are evaluated on the action first and
        # OR-ed over the ranks (all-reduce MAX of two int32 per step) before the step kernel reads them -- the master
        # reset then fires iff every rank's action is all ones, as in the unsharded run.  No effect without a process group.
        self.global_action_predicates = bool(kwargs.get("global_action_predicates", False))
        # 64x64 fp32 batches of more than 16 instances: every step also leaves its output as 64 row words per grid, and the
        # next step reads THOSE (512 B) instead of the fp32 cells (16 KiB) -- as long as ``universe`` is the very tensor
        # the last step returned and no torch operation has written to it since (identity + version counter; anything else
        # reads the fp32 cells: ``_packed_state_of``).  Same trajectories bit for bit.  What the check cannot see is a write
        # through ``universe.data`` or through the raw device pointer (DLPack, a foreign kernel): code that edits the state
        # that way passes packed_state=False (or sets CARLE_PACKED_STATE=0 for the process).
        self.packed_state = bool(kwargs.get("packed_state", _PACKED_STATE))
        self.storage = kwargs.get("storage", "f32")
        if self.storage not in ("f32", "u8", "b1"):
            raise ValueError("storage must be 'f32' (reference format), 'u8' or 'b1' (bit-packed)")

        self.set_neighborhood()
        self.set_action_padding()

        self.allowed_rules = [str(num) for num in range(9)]
        self.birth = [3]                                         # Conway's Life, env.py:58-59
        self.survive = [2, 3]

        self._status = None          # device int32[32]: flags + counters (include/carle_hip.h)
        self._slot = 0
        self._reward_stats = None    # dist.RunningRewardStats whose triples the step epilogue feeds
        self._moments_merged = False
        self.universe = None
        self.action = None
        self.log = []
        self.instance_id = str(int(time.time()))

    # ------------------------------------------------------------------ rules (env.py:62-85)
    def birth_rule_from_string(self, my_string="B3"):
        self.birth = sorted({int(ch) for ch in my_string if ch in self.allowed_rules})

    def survive_rule_from_string(self, my_string="S23"):
        self.survive = sorted({int(ch) for ch in my_string if ch in self.allowed_rules})

    def rules_from_string(self, my_string="B3/S23"):
        parts = my_string.split("/")
        self.birth_rule_from_string(parts[0])
        self.survive_rule_from_string(parts[1])

    def rule_mask(self):
        """18-bit rule mask from the CURRENT ``birth``/``survive`` lists (re-read every step:
        callers assign them directly, train_mcl.py:56-57, eval.py:58-59)."""
        if self.strict_reference_errors and (len(self.birth) == 0 or len(self.survive) == 0):
            raise TypeError("reduce() of empty sequence with no initial value")
        return hip.rule_mask(self.birth, self.survive)

    # ------------------------------------------------------------------ construction helpers
    def set_neighborhood(self):
        """Registers the Moore-neighbourhood conv so that ``neighborhood.weight`` (1,1,3,3)
        exists in state_dicts exactly as in the reference (env.py:87-116).  The HIP kernel has
        the Moore kernel built in; a loaded state_dict with different weights is rejected."""
        self.neighborhood = nn.Conv2d(1, 1, 3, padding=1, padding_mode="circular", bias=False)
        with torch.no_grad():
            self.neighborhood.weight.copy_(torch.tensor(_MOORE).reshape(1, 1, 3, 3))
        self.neighborhood.to(self.my_device if self.my_device.type == "cpu" or
                             torch.cuda.is_available() else "cpu")
        self.register_load_state_dict_post_hook(_check_moore)

    def set_action_padding(self):
        """Centre placement of the action window, env.py:119-132 (same arithmetic, so the odd /
        non-square inconsistencies of the reference are NOT extended: see ``_check_domain``)."""
        asym_w = (self.width - self.action_width) % 2
        asym_h = (self.height - self.action_height) % 2
        self.action_width -= (self.width % 2)
        self.action_height -= (self.height % 2)
        wp = (self.width - self.action_width) // 2
        hp = (self.height - self.action_height) // 2
        self.action_padding = nn.ZeroPad2d(padding=(hp, hp + asym_h, wp, wp + asym_w))

    def _check_domain(self):
        h, w, ah, aw = self.height, self.width, self.action_height, self.action_width
        if h != w or h % 2 or ah != aw or ah > h or (h - ah) % 2 or ah < 1:
            raise ValueError(
                f"unsupported geometry {h}x{w} / action {ah}x{aw}: the reference is only "
                "consistent for square even grids with a centred square window, (H-AH) even "
                "(carle/env.py:119-132)")

    # ------------------------------------------------------------------ counters (device side)
    def _ensure_status(self):
        if self._status is None or self._status.device != self.my_device:
            self._status = hip.new_status(self.my_device)
            self._slot = 0
            self._write_stats_pointers()

    def _install_reward_stats(self, stats):
        """Called by ``dist.RunningRewardStats.attach``: the step epilogue reduces the rewards of a
        step into ``stats`` when asked to (CARLE_STEP_MERGE_MOMENTS), through two device pointers
        kept in the status block (rewritten after every ``reset``)."""
        self._reward_stats = stats
        self._write_stats_pointers()

    def _write_stats_pointers(self):
        stats = self.__dict__.get("_reward_stats")
        if stats is None or self._status is None:
            return
        trace = stats.__dict__.get("_trace")
        out = stats._batch if trace is None else trace       # a trace: MOMENTS_OUT is a ring of rows (CARLE_ST_TRACE_*)
        ptrs = torch.tensor([out.data_ptr(), stats._pending.data_ptr()], dtype=torch.int64)
        self._status[hip.ST_MOMENTS_OUT:hip.ST_MOMENTS_OUT + 4] = \
            ptrs.view(torch.int32).to(self._status.device, non_blocking=True)
        ring = torch.tensor([0 if trace is None else trace.shape[0],
                             0 if trace is None else stats._trace_pos % trace.shape[0]], dtype=torch.int32)
        self._status[hip.ST_TRACE_CAP:hip.ST_TRACE_CAP + 2] = ring.to(self._status.device, non_blocking=True)

    def _bits_buffer(self, out, slot):
        """Where the one-launch step of a 64- / 256- / 512-column fp32 grid leaves a bit-packed copy of its output (one buffer
        per step parity) -- only while a reward net that reads it sits on this environment (``want_packed_copy``)."""
        if out.dtype != torch.float32 or out.shape[3] not in (64, 256, 512) or out.shape[2] != out.shape[3]:
            return None
        # ... or, for batches of 64x64 grids beyond the single-workgroup kernel, always: the next step then reads its input
        # state from the copy (512 B per grid) instead of the fp32 cells (16 KiB) -- ``_packed_state_of``
        if not self.__dict__.get("want_packed_copy") and not (self.packed_state and (out.shape[3] != 64 or out.shape[0] > 16)):
            return None
        bufs = self.__dict__.get("_bits_bufs")
        shape = (out.shape[0], 1, out.shape[2], out.shape[3] // 64)
        if bufs is None or tuple(bufs[0].shape) != shape or bufs[0].device != out.device:
            bufs = self.__dict__["_bits_bufs"] = [torch.empty(shape, dtype=torch.int64, device=out.device) for _ in (0, 1)]
        return bufs[slot]

    def _packed_state_of(self, u_in, bits_out):
        """The bit-packed copy of the CURRENT state ``u_in`` if the last step left one (``u_in`` is the very tensor that step
        returned and no torch operation has written to it since: identity + version counter), for the one-launch step that
        reads it instead of the fp32 cells (``carle_env_step_f32_packed``: 64x64 grids).  None: read the fp32 state."""
        if bits_out is None or not self.packed_state or u_in.shape[3] not in (64, 256, 512) or u_in.shape[2] != u_in.shape[3]:
            return None
        bits = self.packed_copy_of(u_in)
        if bits is None or bits is bits_out or bits.shape != bits_out.shape:      # (the two parity buffers never alias)
            return None
        # a step recorded into a HIP graph is replayed on whatever its input buffer holds by then: record the fp32 reader
        if torch.cuda.is_current_stream_capturing():
            return None
        return bits

    def packed_copy_of(self, obs):
        """The bit-packed copy (int64 (N,1,H,W/64)) the last step left of ``obs``, or None: only for the very tensor the
        step returned, unmodified since."""
        hit = self.__dict__.get("_obs_bits")
        if hit is not None and hit[0]() is obs and hit[1] == obs._version:
            return hit[2]
        return None

    def settled_status(self):
        """The status block with every finished step applied to the counters: the one-launch step
        (CARLE_STEP_DEFERRED, carle_hip.h) leaves its counter update pending until the next call on the block."""
        if self._status is not None and self._status.is_cuda:
            hip.env_settle(self._status)
            self.__dict__["_moments_pending"] = False
        return self._status

    def _settle_pending_moments(self):
        """Reward moments a one-launch CA + RND2D step left as partial sums (mcl.RND2D._fused_step) are merged before
        a call whose kernels update the moment accumulators on their own (small-batch / multi-step launches)."""
        if self.__dict__.get("_moments_pending"):
            self.settled_status()

    def _read_counter(self, idx):
        if self._status is None:
            return 0
        host = self.settled_status().cpu()        # lazy synchronisation point
        return int(host[idx:idx + 2].view(torch.int64).item())

    def _write_counter(self, idx, value):
        if self._status is None:
            return
        self.settled_status()                     # a pending step must not be applied on top of the new value
        v = torch.tensor([int(value)], dtype=torch.int64).view(torch.int32)
        self._status[idx:idx + 2] = v.to(self._status.device, non_blocking=True)

    @property
    def step_number(self):
        return self._read_counter(hip.ST_STEP_NUMBER)

    @step_number.setter
    def step_number(self, value):
        self._write_counter(hip.ST_STEP_NUMBER, value)

    @property
    def steps_since_action(self):
        return self._read_counter(hip.ST_STEPS_SINCE_ACTION)

    @steps_since_action.setter
    def steps_since_action(self, value):
        self._write_counter(hip.ST_STEPS_SINCE_ACTION, value)

    @property
    def master_resets(self):
        """Number of whole-batch master resets so far (build-side diagnostic)."""
        return self._read_counter(hip.ST_RESETS)

    # ------------------------------------------------------------------ gym surface
    def reset(self):
        """All-dead universe, counters zeroed; returns the universe tensor itself (env.py:134-148)."""
        hip.require_gpu()
        self._check_domain()
        self.universe = torch.zeros(self.instances, 1, self.height, self.width,
                                    dtype=torch.float32, device=self.my_device)
        self._ensure_status()
        self._settle_pending_moments()            # reward moments of the last step must not vanish with the counters
        self._status.zero_()
        self._write_stats_pointers()
        self._slot = 0
        self.instance_id = str(int(time.time()))
        self.log = []
        return self.universe

    def get_observation(self):
        return self.universe

    # --- hooks of carle_amd.graphs.StepGraph (the state buffers of a captured step are fixed) ---
    def _new_out(self, like):
        """The buffer the next step writes its state to: fresh (the reference returns a new tensor every step,
        env.py:229), unless a StepGraph capture has named one."""
        o = self.__dict__.pop("_out_override", None)
        if o is not None:
            if o.shape != like.shape or o.dtype != like.dtype or o.device != like.device:
                raise RuntimeError("StepGraph state buffer does not match the environment's state")
            return o
        return torch.empty_like(like)

    def _raw_state(self):
        """The tensor the step kernels read (fp32 / uint8 universe; the packed words for storage='b1')."""
        return self._device_universe()

    def _set_raw_state(self, t):
        self.universe = t

    def _device_action(self, action):
        """Tensor conversion / rank padding / device move of env.py:152-159, then the centre crop
        of an oversize action (env.py:164-169).  Returns (cropped fp32 action, was_cropped,
        uncropped device action)."""
        if not isinstance(action, torch.Tensor):
            action = torch.as_tensor(action, dtype=torch.float32)
        while action.dim() < 4:
            action = action.unsqueeze(0)
        if action.device != self.my_device:
            action = action.to(self.my_device, non_blocking=True)
        if action.dtype != torch.float32:
            action = action.float()
        full = action
        cropped = False
        if action.shape[3] > self.action_width and action.shape[1] < self.width:
            off_y = (self.width - self.action_width) // 2
            off_x = (self.height - self.action_height) // 2
            action = action[:, :, off_y:action.shape[2] - off_y, off_x:action.shape[3] - off_x]
            cropped = True
        assert action.shape[2] == self.action_width, \
            f"action width is wrong {action.shape[2]} not {self.action_width}, {tuple(action.shape)}"
        assert action.shape[3] == self.action_height, \
            f"action height is wrong {action.shape[3]} not {self.action_height}, {tuple(action.shape)}"
        if action.shape[0] != self.instances:
            if action.shape[0] != 1:
                raise ValueError(f"action batch {action.shape[0]} != instances {self.instances}")
            action = action.expand(self.instances, -1, -1, -1)   # broadcast like logical_xor
        return action.contiguous(), cropped, full

    def _exact_predicates(self, action, slot):
        """env.py:191 / env.py:208 with the reference's own reductions, written into the flag slot
        of the status block on the device (ANY_NONZERO = sum != 0, NOT_ALL_ONES = mean != 1.0)."""
        a = action if action.dtype == torch.float32 else action.float()
        flags = torch.stack([torch.sum(a) != 0, torch.mean(a) != 1.0]).to(torch.int32)
        self._status[slot::2][:2] = flags                # words slot and 2 + slot

    def _share_predicates(self):
        """``global_action_predicates`` under an exchanging process group: the function that ORs this step's two flags
        over the ranks (carle_amd.dist.all_reduce_flags), else None."""
        if not self.global_action_predicates:
            return None
        from . import dist as cdist
        return cdist.all_reduce_flags if cdist.exchanging() else None

    def _device_universe(self):
        u = self.universe
        if u is None:
            raise RuntimeError("call reset() before step()/apply_action()")
        if not isinstance(u, torch.Tensor):
            u = torch.as_tensor(u, dtype=torch.float32)
        if u.dim() != 4 or tuple(u.shape) != (self.instances, 1, self.height, self.width):
            raise ValueError(f"universe has shape {tuple(u.shape)}, expected "
                             f"{(self.instances, 1, self.height, self.width)}")
        if u.device != self.my_device or u.dtype != torch.float32 or not u.is_contiguous():
            u = u.to(self.my_device, torch.float32).contiguous()
        return u

    def apply_action(self, action):
        """Toggle only (env.py:150-182): universe <- 1.0 * xor(universe, pad(action)).

        Stand-alone use of the reference method (``step`` fuses the toggle into the CA kernel);
        runs the elementwise HIP kernel ``carle_ca_toggle_f32``.
        """
        hip.require_gpu()
        self._check_domain()
        a, _, _ = self._device_action(action)
        u = self._device_universe()
        out = torch.empty_like(u)
        hip.ca_toggle_f32(u, a, out)
        self.universe = out

    def step(self, action):
        """One environment step (env.py:188-242): toggle, master-reset check, one CA generation.

        Returns ``(obs, reward, done, info)`` with ``obs is self.universe`` (N,1,H,W) fp32 on the
        device, ``reward`` zeros (N,1) on the device, ``done`` zeros (N,1) on the CPU and
        ``info`` a list of N dicts -- the reference's exact return structure (env.py:236-242).
        """
        return self._step(action, None)

    def _step(self, action, launch):
        """``launch`` (optional): replaces the plain library call -- a wrapper that sits directly on
        this environment passes a closure that enqueues step + its reward bonus in one call."""
        hip.require_gpu()
        self._check_domain()
        self._ensure_status()
        self.action = action
        if self.logging:
            self.log_universe()
        u_in = self._device_universe()
        rule = self.rule_mask()
        slot = self._slot
        out = self._new_out(u_in)
        reward = torch.empty(self.instances, 1, dtype=torch.float32, device=self.my_device)
        # the step's reward is final (this env is the outermost one a RunningRewardStats is attached
        # to): the epilogue kernel reduces it, no separate moments launch
        merge = launch is None and self.__dict__.get("_final_reward_stats") is not None
        self._moments_merged = False
        if getattr(action, "materialize", None) is not None:      # agents.DeviceRandomAction
            if tuple(action.shape) != (self.instances, 1, self.action_width, self.action_height):
                raise AssertionError(f"action shape {action.shape} does not match the environment")
            if self._share_predicates() is None and (
                    (self.height == 64 and self.action_width % 8 == 0) or
                    (self.height in (256, 512) and self.action_width % 4 == 0 and
                     (self.width - self.action_width) % 8 == 0)):
                # the toggles are drawn inside the step kernel: no action tensor exists
                hip.env_step_f32_rand(u_in, out, reward, rule, self._status, slot, self.action_height,
                                      self.action_width, action.p, action.seed, action.offset, None, merge)
                self._moments_merged = merge
                self._slot = slot ^ 1
                self.universe = out
                return self.universe, reward, _done_zeros(self.instances), [{}] * self.instances
            action = action.materialize()
        act, cropped, full = self._device_action(action)
        publish = not cropped
        share = self._share_predicates()
        if self.exact_action_predicates:
            self._exact_predicates(full, slot)
            publish = False
        elif cropped or share:
            # the two whole-batch predicates are defined on the action AS PASSED (env.py:191,208);
            # for an oversize action evaluate them on the uncropped tensor (cold path, no sync)
            hip.action_flags_f32(full.contiguous(), self._status, slot)
            publish = False
        if share:
            share(self._status, slot)
        if launch is None:
            bits = self._bits_buffer(out, slot)
            state_bits = self._packed_state_of(u_in, bits)
            if state_bits is not None:
                # the last step left the state bit-packed as well: this one reads 512 B instead of 16 KiB per grid
                hip.env_step_f32_packed(state_bits, act, out, reward, rule, self._status, slot, publish, merge, bits_out=bits)
                self.__dict__["_obs_bits"] = (weakref.ref(out), out._version, bits)
            elif hip.env_step_f32(u_in, act, out, reward, rule, self._status, slot, publish, merge, bits_out=bits):
                # the launch left a bit-packed copy of the new state: the reward nets (and the next step) read that
                self.__dict__["_obs_bits"] = (weakref.ref(out), out._version, bits)
            self._moments_merged = merge
        else:
            launch(u_in, act, out, reward, rule, self._status, slot, publish)
        self._slot = slot ^ 1
        self.universe = out
        done = _done_zeros(self.instances)
        info = [{}] * self.instances
        return self.universe, reward, done, info

    def run(self, actions=None, steps=None, agent=None):
        """Several consecutive ``step`` calls in ONE library call (``carle_env_run_f32``; build-side
        extension).  The state, counters, flags and return value are those of the loop

            for t in range(k): obs, reward, done, info = env.step(action_t)

        (carle/train_mcl.py:64-69) for action sources that do not look at the observations:
          * ``actions``: a (k, N, 1, AH, AW) tensor, one action per step;
          * ``agent``: a ``carle_amd.agents.RandomAgent`` -- its next ``steps`` draws, made inside the
            step kernel (the agent's call counter advances by ``steps``);
          * neither: ``steps`` steps without an action.
        One ctypes round trip instead of k: for small batches the per-step cost is the kernel launch
        (N <= 16 at 64x64: one launch per step), not Python.  fp32 storage only.  The tensor that held the state on
        entry (a previously returned observation) is never written."""
        hip.require_gpu()
        self._check_domain()
        self._ensure_status()
        self._settle_pending_moments()
        if self.storage != "f32" or self.logging or self.exact_action_predicates:
            raise NotImplementedError("run() covers storage='f32' without logging / exact predicates")
        u = self._device_universe()
        ah = aw = 0
        p, seed, offset = -1.0, 0, 0
        if actions is not None:
            if not isinstance(actions, torch.Tensor):
                actions = torch.as_tensor(actions, dtype=torch.float32)
            actions = actions.to(self.my_device, torch.float32).contiguous()
            if actions.dim() != 5 or tuple(actions.shape[1:]) != (self.instances, 1, self.action_width,
                                                                  self.action_height):
                raise AssertionError(f"actions must be (k, {self.instances}, 1, {self.action_width}, "
                                     f"{self.action_height}), got {tuple(actions.shape)}")
            k = actions.shape[0]
            if steps is not None and int(steps) != k:
                raise ValueError("steps disagrees with actions.shape[0]")
        else:
            if steps is None:
                raise ValueError("run() needs actions or steps")
            k = int(steps)
            if agent is not None:
                if not all(hasattr(agent, a) for a in ("toggle_rate", "seed", "calls")):
                    raise TypeError("agent must be a carle_amd.agents.RandomAgent")
                if (agent.action_width, agent.action_height) != (self.action_width, self.action_height):
                    raise AssertionError("agent action size does not match the environment")
                if not ((self.height == 64 and self.action_width % 8 == 0) or
                        (self.height in (256, 512) and self.action_width % 4 == 0 and
                         (self.width - self.action_width) % 8 == 0)):
                    raise NotImplementedError("in-kernel toggles need an aligned action window "
                                              "(carle_ca_step_f32_rand)")
                p, seed, offset = agent.toggle_rate, agent.seed, agent.calls
                ah, aw = self.action_height, self.action_width
        if k < 0:
            raise ValueError("steps must be >= 0")
        reward = torch.zeros(self.instances, 1, dtype=torch.float32, device=self.my_device)
        merge = self.__dict__.get("_final_reward_stats") is not None
        self._moments_merged = False
        if k > 0:
            # `u` may be a tensor an earlier step()/reset() returned as the observation: it is only read; the k
            # generations ping-pong between two private buffers (the reference returns a fresh tensor per step,
            # env.py:229 -- a caller still holding an older observation must keep seeing that generation)
            buf_a = torch.empty_like(u)
            buf_b = torch.empty_like(u) if k > 1 else None
            hip.env_run_f32(u, buf_a, buf_b, actions, reward, self.rule_mask(), self._status, self._slot, k, ah, aw,
                            p, seed, offset, merge)
            self.universe = buf_a if k & 1 else buf_b
            self._slot = (self._slot + k) & 1
            self._moments_merged = merge
            if agent is not None and actions is None:
                agent.calls += k
        self.action = None
        return self.universe, reward, _done_zeros(self.instances), [{}] * self.instances

    def multistep(self, steps):
        """``steps`` generations with no action in ONE kernel launch: the state, counters and return
        value of ``steps`` consecutive ``step(zeros)`` calls (env.py:188-242; build-side extension).
        The grid is bit-packed on the fly and stays in registers between generations
        (``carle_ca_multistep_b1``); sizes the packed kernels do not cover fall back to a loop."""
        hip.require_gpu()
        self._check_domain()
        self._ensure_status()
        self._settle_pending_moments()
        steps = int(steps)
        if steps < 0:
            raise ValueError("steps must be >= 0")
        if self.width not in (64, 128, 256, 512) or self.logging:
            zero = torch.zeros(self.instances, 1, self.action_width, self.action_height,
                               device=self.my_device)
            for _ in range(steps):
                self.step(zero)
        else:
            bits = hip.pack_f32_b1(self._device_universe())
            hip.ca_multistep_b1(bits, bits, self.rule_mask(), steps, self._status)
            self.universe = hip.unpack_b1_f32(bits)
        self.action = None
        self._moments_merged = False
        reward = torch.zeros(self.instances, 1, dtype=torch.float32, device=self.my_device)
        return self.get_observation(), reward, _done_zeros(self.instances), [{}] * self.instances

    # ------------------------------------------------------------------ cold helpers
    def render(self):
        rows = self.universe[0, 0].cpu()
        print("\n CA Universe")
        for r in rows:
            print("".join("o" if v else " " for v in r.tolist()))


class CARLEPacked(CARLE):
    """CARLE with bit-packed state in HBM (``CARLE(storage="b1")``).

    Same gym surface and semantics as ``CARLE``; the universe lives as int64 words
    ``universe_bits`` (N,1,H,W/64), bit j of word i = column 64*i+j, and every step moves 1/32 of
    the bytes of the fp32 format.  ``obs="b1"`` (default) returns the packed tensor from
    ``reset``/``step``; ``obs="f32"`` materialises the reference-format observation with the
    unpack kernel.  ``universe`` stays readable/writable as fp32 (unpacked / packed on demand).
    Actions may be fp32 (packed on the fly, exact predicates) or already packed int64.
    """

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.obs_format = kwargs.get("obs", "b1")
        if self.obs_format not in ("f32", "b1"):
            raise ValueError("obs must be 'f32' or 'b1'")
        self.__dict__["_bits"] = None
        self.__dict__["_f32"] = None

    def _check_domain(self):
        super()._check_domain()
        if self.width not in (64, 128, 256, 512) or self.action_width % 64:
            raise ValueError("storage='b1' needs H == W in {64,128,256,512} and AW % 64 == 0")

    # fp32 view of record: unpacked lazily, packed on assignment
    @property
    def universe(self):
        d = self.__dict__
        if d["_bits"] is None:
            return None
        if d["_f32"] is None:
            d["_f32"] = hip.unpack_b1_f32(d["_bits"])
        return d["_f32"]

    @universe.setter
    def universe(self, value):
        d = self.__dict__
        if value is None:
            d["_bits"], d["_f32"] = None, None
            return
        hip.require_gpu()
        u = value if isinstance(value, torch.Tensor) else torch.as_tensor(value, dtype=torch.float32)
        if tuple(u.shape) != (self.instances, 1, self.height, self.width):
            raise ValueError(f"universe has shape {tuple(u.shape)}")
        u = u.to(self.my_device, torch.float32).contiguous()
        d["_bits"] = hip.pack_f32_b1(u)
        d["_f32"] = None

    @property
    def universe_bits(self):
        return self.__dict__["_bits"]

    @universe_bits.setter
    def universe_bits(self, value):
        self.__dict__["_bits"] = value
        self.__dict__["_f32"] = None

    def _obs(self):
        return self.universe if self.obs_format == "f32" else self.__dict__["_bits"]

    def _raw_state(self):
        if self.__dict__["_bits"] is None:
            raise RuntimeError("call reset() before step()")
        return self.__dict__["_bits"]

    def _set_raw_state(self, t):
        self.universe_bits = t

    def reset(self):
        hip.require_gpu()
        self._check_domain()
        self.__dict__["_bits"] = torch.zeros(self.instances, 1, self.height, self.width // 64,
                                             dtype=torch.int64, device=self.my_device)
        self.__dict__["_f32"] = None
        self._ensure_status()
        self._settle_pending_moments()            # reward moments of the last step must not vanish with the counters
        self._status.zero_()
        self._write_stats_pointers()
        self._slot = 0
        self.instance_id = str(int(time.time()))
        self.log = []
        return self._obs()

    def get_observation(self):
        return self._obs()

    def apply_action(self, action):
        a, _, _ = self._device_action(action)
        out = torch.empty_like(self.universe)
        hip.ca_toggle_f32(self.universe, a, out)
        self.universe = out

    def step(self, action):
        hip.require_gpu()
        self._check_domain()
        self._ensure_status()
        self.action = action
        d = self.__dict__
        bits = d["_bits"]
        if bits is None:
            raise RuntimeError("call reset() before step()")
        slot = self._slot
        publish = True
        if isinstance(action, torch.Tensor) and action.dtype == torch.int64:
            act = action                                   # already packed (N,1,AH,AW/64)
            if act.device != self.my_device:
                act = act.to(self.my_device, non_blocking=True)
            if tuple(act.shape) != (self.instances, 1, self.action_height, self.action_width // 64):
                raise AssertionError(f"packed action has shape {tuple(act.shape)}")
            act = act.contiguous()
        else:
            if getattr(action, "materialize", None) is not None:      # agents.DeviceRandomAction
                if (tuple(action.shape) == (self.instances, 1, 64, 64) and self.action_width == 64
                        and not self.exact_action_predicates and self._share_predicates() is None):
                    # the toggles are drawn inside the packed step kernel: no action tensor exists
                    out = self._new_out(bits)
                    reward = torch.empty(self.instances, 1, dtype=torch.float32, device=self.my_device)
                    merge = self.__dict__.get("_final_reward_stats") is not None
                    hip.env_step_b1_rand(bits, out, reward, self.rule_mask(), self._status, slot, 64, 64,
                                         action.p, action.seed, action.offset, merge)
                    self._moments_merged = merge
                    self._slot = slot ^ 1
                    d["_bits"], d["_f32"] = out, None
                    return self._obs(), reward, _done_zeros(self.instances), [{}] * self.instances
                action = action.materialize()
            a32, cropped, full = self._device_action(action)
            if self.exact_action_predicates:
                self._exact_predicates(full, slot)
                act = hip.pack_f32_b1(a32)
            elif cropped:
                hip.action_flags_f32(full.contiguous(), self._status, slot)
                act = hip.pack_f32_b1(a32)
            else:
                act = hip.pack_f32_b1(a32, self._status, slot)   # predicates of the fp32 values
            publish = False
        share = self._share_predicates()
        if share:
            if publish:
                raise NotImplementedError("global_action_predicates needs fp32 actions (a pre-packed action's predicates are "
                                          "only known inside the step kernel)")
            share(self._status, slot)
        out = self._new_out(bits)
        reward = torch.empty(self.instances, 1, dtype=torch.float32, device=self.my_device)
        merge = self.__dict__.get("_final_reward_stats") is not None
        hip.env_step_b1(bits, act, out, reward, self.rule_mask(), self._status, slot, publish, merge)
        self._moments_merged = merge
        self._slot = slot ^ 1
        d["_bits"] = out
        d["_f32"] = None
        done = _done_zeros(self.instances)
        info = [{}] * self.instances
        return self._obs(), reward, done, info


    def multistep(self, steps):
        hip.require_gpu()
        self._check_domain()
        self._ensure_status()
        d = self.__dict__
        if d["_bits"] is None:
            raise RuntimeError("call reset() before multistep()")
        if int(steps) < 0:
            raise ValueError("steps must be >= 0")
        out = torch.empty_like(d["_bits"])
        hip.ca_multistep_b1(d["_bits"], out, self.rule_mask(), int(steps), self._status)
        d["_bits"], d["_f32"] = out, None
        self.action = None
        self._moments_merged = False
        reward = torch.zeros(self.instances, 1, dtype=torch.float32, device=self.my_device)
        return self._obs(), reward, _done_zeros(self.instances), [{}] * self.instances


class CARLEBytes(CARLE):
    """CARLE with uint8 state in HBM (``CARLE(storage="u8")``): the reference layout at one byte
    per cell, a quarter of the fp32 traffic, still an ordinary (N,1,H,W) tensor.

    ``universe`` IS the uint8 tensor (``obs is env.universe`` holds as in the reference);
    ``obs="f32"`` makes ``reset``/``step`` return an fp32 copy instead.  Assigning an fp32 (or
    bool) universe converts it with the truthiness of carle/env.py:182.  Actions may be fp32
    (what the reference's agents emit) or uint8.
    """

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.obs_format = kwargs.get("obs", "u8")
        if self.obs_format not in ("f32", "u8"):
            raise ValueError("obs must be 'f32' or 'u8'")

    def _obs(self):
        return self.universe.float() if self.obs_format == "f32" else self.universe

    def reset(self):
        hip.require_gpu()
        self._check_domain()
        self.universe = torch.zeros(self.instances, 1, self.height, self.width, dtype=torch.uint8,
                                    device=self.my_device)
        self._ensure_status()
        self._settle_pending_moments()            # reward moments of the last step must not vanish with the counters
        self._status.zero_()
        self._write_stats_pointers()
        self._slot = 0
        self.instance_id = str(int(time.time()))
        self.log = []
        return self._obs()

    def get_observation(self):
        return self._obs()

    def _device_universe(self):
        u = self.universe
        if u is None:
            raise RuntimeError("call reset() before step()/apply_action()")
        if not isinstance(u, torch.Tensor):
            u = torch.as_tensor(u)
        if u.dim() != 4 or tuple(u.shape) != (self.instances, 1, self.height, self.width):
            raise ValueError(f"universe has shape {tuple(u.shape)}, expected "
                             f"{(self.instances, 1, self.height, self.width)}")
        if u.dtype != torch.uint8:
            u = (u != 0).to(torch.uint8)
        if u.device != self.my_device or not u.is_contiguous():
            u = u.to(self.my_device).contiguous()
        return u

    def apply_action(self, action):
        hip.require_gpu()
        self._check_domain()
        a, _, _ = self._device_action(action)
        u = self._device_universe()
        out = torch.empty_like(u)
        hip.ca_toggle_u8(u, a, out)
        self.universe = out

    def multistep(self, steps):
        hip.require_gpu()
        self._check_domain()
        self._ensure_status()
        steps = int(steps)
        if steps < 0:
            raise ValueError("steps must be >= 0")
        if self.width not in (64, 128, 256, 512) or self.logging:
            zero = torch.zeros(self.instances, 1, self.action_width, self.action_height,
                               dtype=torch.uint8, device=self.my_device)
            for _ in range(steps):
                self.step(zero)
        else:
            bits = hip.pack_u8_b1(self._device_universe())
            hip.ca_multistep_b1(bits, bits, self.rule_mask(), steps, self._status)
            self.universe = hip.unpack_b1_u8(bits)
        self.action = None
        self._moments_merged = False
        reward = torch.zeros(self.instances, 1, dtype=torch.float32, device=self.my_device)
        return self._obs(), reward, _done_zeros(self.instances), [{}] * self.instances

    def step(self, action):
        hip.require_gpu()
        self._check_domain()
        self._ensure_status()
        self.action = action
        if self.logging:
            self.log_universe()
        u_in = self._device_universe()
        slot = self._slot
        if getattr(action, "materialize", None) is not None:      # agents.DeviceRandomAction
            aw = self.action_width
            if (tuple(action.shape) == (self.instances, 1, aw, self.action_height) and not self.exact_action_predicates
                    and self._share_predicates() is None
                    and ((self.height == 64 and aw % 8 == 0) or
                         (self.height in (256, 512) and aw % 4 == 0 and (self.width - aw) % 8 == 0))):
                # the toggles are drawn inside the step kernel: no action tensor exists
                out = self._new_out(u_in)
                reward = torch.empty(self.instances, 1, dtype=torch.float32, device=self.my_device)
                merge = self.__dict__.get("_final_reward_stats") is not None
                hip.env_step_u8_rand(u_in, out, reward, self.rule_mask(), self._status, slot, self.action_height, aw,
                                     action.p, action.seed, action.offset, merge)
                self._moments_merged = merge
                self._slot = slot ^ 1
                self.universe = out
                return self._obs(), reward, _done_zeros(self.instances), [{}] * self.instances
            action = action.materialize()
        share = self._share_predicates()
        if (isinstance(action, torch.Tensor) and action.dtype == torch.uint8 and action.is_cuda and share is None
                and tuple(action.shape) == (self.instances, 1, self.action_width, self.action_height)):
            act, cropped = action.contiguous(), False              # hot path: uint8 on the device
            if self.exact_action_predicates:
                self._exact_predicates(act, slot)
                cropped = True                                     # = do not publish from the kernel
        else:
            act, cropped, full = self._device_action(action)
            if self.exact_action_predicates:
                self._exact_predicates(full, slot)
                cropped = True
            elif cropped or share:
                hip.action_flags_f32(full.contiguous(), self._status, slot)
                cropped = True
            if share:
                share(self._status, slot)
        out = self._new_out(u_in)
        reward = torch.empty(self.instances, 1, dtype=torch.float32, device=self.my_device)
        merge = self.__dict__.get("_final_reward_stats") is not None
        hip.env_step_u8(u_in, act, out, reward, self.rule_mask(), self._status, slot, not cropped, merge)
        self._moments_merged = merge
        self._slot = slot ^ 1
        self.universe = out
        return self._obs(), reward, _done_zeros(self.instances), [{}] * self.instances


def _check_moore(module, incompatible_keys):
    w = module.neighborhood.weight.detach().cpu().reshape(3, 3)
    if not torch.equal(w, torch.tensor(_MOORE)):
        raise NotImplementedError(
            "loaded neighborhood.weight is not the Moore kernel; the HIP CA kernel implements "
            "the Moore neighbourhood only (carle/env.py:95-96)")

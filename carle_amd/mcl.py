"""carle_amd.mcl -- reward wrappers ("mesocorticolimbic" system) on top of the HIP hot path.

Mirrors ``carle.mcl.{Motivator, RND2D, AE2D}`` of the reference (riveSunder/carle,
carle/mcl.py:29-84, 233-421, 424-556): same constructor, attribute names, module / parameter
names (so ``evaluation/RND2D_mcl.pt`` / ``AE2D_mcl.pt`` style state_dicts load strictly), same
``reset`` / ``step`` / ``get_bonus*`` methods.

Underneath, every forward pass that produces a reward (``get_bonus``, ``get_bonus_only``,
``get_bonus_accumulate``, ``forward``, ``random_forward``) runs the hand-written HIP kernels
``carle_rnd2d_bonus_f32`` / ``carle_ae2d_bonus_f32`` (eval-mode semantics: Dropout = identity).
The predictor UPDATE (mcl.py:333-341, 354-386): the gradient of the update window is accumulated by
the training kernels (``carle_rnd2d_train_*`` / ``carle_ae2d_train_*``: fused at 64x64, tiled at
multiples of 64, staged otherwise) in the same pass that produces the bonus, and ``optimizer.step()``
is ``carle_adam_update_f32`` (torch.optim.Adam's arithmetic in one launch).  There is no autograd path
behind ``step`` / ``get_bonus_*``: a geometry the gradient kernels do not cover (H = W not a multiple of 8)
raises ``NotImplementedError`` when an update is asked for (the forward bonus works wherever the staged
kernels are defined); only a caller's own ``update_predictor(loss)`` differentiates its own loss.  The hand-coded
wrappers (Parsimony / Corner / Speed / Puffer) keep their per-step state on the device: no host synchronisation per step.
"""
import weakref

import os

import torch
import torch.nn as nn

from . import hip


# bookkeeping attributes rewritten every step: plain Python / tensor values that need none of
# nn.Module.__setattr__'s parameter / buffer / sub-module handling (~1 us each)
_PLAIN_WRAPPER_ATTRS = frozenset((
    "_ws", "_tables_key", "_tables_key_next", "_bonus_buf", "_step_bonus", "buffer_length",
    "accumulate_loss", "_grad_used", "live_cells", "speed", "velocity", "center_of_mass",
    "growing_steps"))


class Motivator(nn.Module):
    """Wrapper base (mcl.py:29-84): holds ``env``, resolves ``inner_env``, forwards rules."""

    def __setattr__(self, name, value):
        if name in _PLAIN_WRAPPER_ATTRS:
            object.__setattr__(self, name, value)
        else:
            super().__setattr__(name, value)

    def __init__(self, env, **kwargs):
        super().__init__()
        # registration order fixes the state_dict key order: inner_env.* before env.*
        if env.inner_env is None:
            self.inner_env = env
        else:
            self.inner_env = env.inner_env
        self.env = env
        self.height = self.inner_env.height
        self.width = self.inner_env.width
        self.action_height = self.inner_env.action_height
        self.action_width = self.inner_env.action_width
        self.birth = self.inner_env.birth
        self.survive = self.inner_env.survive
        self.my_device = self.inner_env.my_device

    def rules_from_string(self, my_string="B3/S23"):
        self.inner_env.rules_from_string(my_string)
        self.birth = self.inner_env.birth
        self.survive = self.inner_env.survive

    def birth_rule_from_string(self, my_string="b3"):
        self.inner_env.birth_rule_from_string(my_string)
        self.birth = self.inner_env.birth

    def survive_rule_from_string(self, my_string="s23"):
        self.inner_env.survive_rule_from_string(my_string)
        self.survive = self.inner_env.survive

    def reset(self):
        return self.env.reset()

    def step(self, action):
        return self.env.step(action)

    def set_no_grad(self):
        pass

    def set_grad(self):
        pass


class _ParamBlob:
    """Flat fp32 copy of a module's parameters in state_dict order, rebuilt only when a
    parameter changed (in-place version counter) or moved (data_ptr / device)."""

    def __init__(self, module):
        self.module = module
        self.key = None
        self.blob = None
        self.params = None
        self.version = 0            # bumped whenever the blob is rebuilt (= the values changed)
        self._previous = None       # the blob of the last generation: kernels enqueued on another
        # stream may still read it when it is replaced, so it is released one rebuild later (an
        # update window = many cross-stream joins later), never straight into the allocator

    def get(self, device):
        # the Parameter objects of an nn.Sequential are stable (.to()/load_state_dict/optimisers
        # mutate them in place): enumerate the module tree once, not on every step
        ps = self.params
        if ps is None:
            ps = self.params = list(self.module.parameters())
        key = (device,) + tuple([(p.data_ptr(), p._version) for p in ps])
        if key != self.key:
            self._previous = self.blob
            with torch.no_grad():
                self.blob = torch.cat([p.detach().reshape(-1).to(device, torch.float32) for p in ps])
            self.key = key
            self.version += 1
        return self.blob

    def updated_in_place(self, device):
        """A kernel rewrote the parameters AND this blob (carle_adam_update_f32): the blob is current, its users'
        derived data (lookup tables) is not.  The parameters' autograd version counters are bumped so that other
        observers of the tensors see the change too."""
        torch._C._increment_version(list(self.params))
        self.key = (device,) + tuple([(p.data_ptr(), p._version) for p in self.params])
        self.version += 1


class RND2D(Motivator):
    """Random network distillation bonus on (N,1,H,W) grids (mcl.py:233-421)."""

    def __init__(self, env, **kwargs):
        super().__init__(env, **kwargs)
        self.my_name = "RND2D"
        self.learning_rate = 6e-2
        self.reward_scale = 1.0
        self.rnd_dim = 16
        self.initialize_predictor()          # predictor BEFORE the random net: same RNG order
        self.initialize_random_network()     # as the reference (mcl.py:248-249)
        self.buffer_length = 0
        self.batch_size = 64
        self.updates = 0
        self._ws = None                      # kernel workspace, reused across steps
        self._tables_key = None              # what the fused-path tables in _ws were built from
        self._tables_key_next = (False, None)
        # obs values are exactly 0/1 (always true for what CARLE.step returns; detected by
        # identity with inner_env.universe).  Set True to assert it for foreign tensors.
        self.assume_binary_obs = False
        self.async_reward = bool(kwargs.get("async_reward", False))
        self._bonus_buf = None
        self._step_bonus = None              # bonus vector of the fused CA + RND2D step (n + 1)
        self._grad = None                    # flat gradient the fused training kernel accumulates
        self._grad_used = False
        self._partial = None                 # per-wave partial gradient rows (scratch)
        # False: step() only evaluates the bonus (forward pass), no window retention / Adam update
        self.train_predictor = bool(kwargs.get("train_predictor", True))
        self.accumulate_loss = 0.0
        # fp32 state of 256 / 512 columns: ask the environment's one-launch step for a bit-packed copy of every new
        # state (CARLE._bits_buffer) -- the tiled kernels of this wrapper read that instead of the fp32 cells
        if getattr(self.inner_env, "storage", None) == "f32" and os.environ.get("CARLE_PACKED_COPY", "1") != "0":
            self.inner_env.__dict__["want_packed_copy"] = True       # (CARLE_PACKED_COPY=0: A/B switch)
            # ... and a reward net stacked directly on an RND2D (AE2D(RND2D(env)), train_mcl.py:36-40): the RND2D's one-launch
            # 64x64 step hands the observation over bit-packed as well (512 B per grid for the 64x64 kernels above it)
            if type(env) is RND2D and env.env is self.inner_env:
                env.__dict__["_hand_bits"] = True

    # ------------------------------------------------------------------ networks
    def initialize_predictor(self):
        dense_nodes = (self.inner_env.width // 8) * (self.inner_env.height // 8)
        self.predictor = nn.Sequential(
            nn.Conv2d(1, 4, 3, padding=1, stride=1),
            nn.Dropout(p=0.1),
            nn.ReLU(),
            nn.MaxPool2d(2, 2, padding=0),
            nn.MaxPool2d(2, 2, padding=0),
            nn.Conv2d(4, 1, 3, padding=1, stride=1),
            nn.Dropout(p=0.1),
            nn.ReLU(),
            nn.MaxPool2d(2, 2, padding=0),
            nn.Dropout(p=0.1),
            nn.Flatten(),
            nn.Linear(dense_nodes, self.rnd_dim),
            nn.Tanh())
        self.optimizer = torch.optim.Adam(self.predictor.parameters(), lr=self.learning_rate)
        self._pred_blob = _ParamBlob(self.predictor)

    def initialize_random_network(self):
        dense_nodes = (self.inner_env.width // 8) * (self.inner_env.height // 8)
        self.random_network = nn.Sequential(
            nn.Conv2d(1, 2, 3, padding=1, stride=1),
            nn.ReLU(),
            nn.MaxPool2d(2, 2, padding=0),
            nn.MaxPool2d(2, 2, padding=0),
            nn.Conv2d(2, 1, 3, padding=1, stride=1),
            nn.ReLU(),
            nn.MaxPool2d(2, 2, padding=0),
            nn.Flatten(),
            nn.Linear(dense_nodes, self.rnd_dim),
            nn.Tanh())
        self._rand_blob = _ParamBlob(self.random_network)
        self.set_grad()

    def set_grad(self):
        for p in self.predictor.parameters():
            p.requires_grad = True
        for p in self.random_network.parameters():
            p.requires_grad = False

    def set_no_grad(self):
        for p in self.predictor.parameters():
            p.requires_grad = False
        for p in self.random_network.parameters():
            p.requires_grad = False

    # ------------------------------------------------------------------ HIP forward passes
    def _obs(self, obs, allow_packed=False):
        """-> (device tensor, binary?)  fp32 (N,1,H,W), or int64 bit-packed when allowed."""
        hip.require_gpu()
        dev = self.inner_env.my_device
        env = self.inner_env
        if obs.dtype == torch.int64:                      # bit-packed observation (storage="b1")
            if obs.device != dev or not obs.is_contiguous():
                obs = obs.to(dev).contiguous()
            if allow_packed and obs.shape[2] == 64 * obs.shape[3] and self._packed_any_size:
                return obs, True
            if allow_packed and obs.shape[2] == 64 and obs.shape[3] == 1:
                return obs, True
            return hip.unpack_b1_f32(obs), True
        if obs.dtype == torch.uint8:                      # uint8 observation (storage="u8"): alive bits
            if obs.device != dev or not obs.is_contiguous():
                obs = obs.to(dev).contiguous()
            if allow_packed and obs.dim() == 4 and obs.shape[2] == obs.shape[3] and obs.shape[2] % 64 == 0:
                return hip.pack_u8_b1(obs), True             # 64x64: fused kernels; larger: tiled kernels
            return (obs != 0).to(torch.float32), True
        binary = self.assume_binary_obs or (getattr(env, "storage", "f32") == "f32" and
                                            obs is env.__dict__.get("universe"))
        if allow_packed and self._packed_any_size and obs.dtype == torch.float32:
            bits = env.packed_copy_of(obs) if hasattr(env, "packed_copy_of") else None
            if bits is not None:                          # the step left a bit-packed copy of this very observation
                return bits, True
        if obs.device != dev or obs.dtype != torch.float32 or not obs.is_contiguous():
            obs = obs.to(dev, torch.float32).contiguous()
        if obs.dim() == 3:
            obs = obs.unsqueeze(1)
        return obs, binary

    def _tables_valid(self, obs, binary, versions):
        """True iff the fused 64x64 kernel can reuse the lookup tables already in ``self._ws``:
        same workspace tensor, same parameter-blob versions, and the previous call was fused too."""
        h = obs.shape[2]
        w = obs.shape[3] * 64 if obs.dtype == torch.int64 else obs.shape[3]
        fused = binary and h == w and h % 64 == 0        # 64x64: workgroup per grid; larger: per 64x64 tile
        # the batch geometry belongs to the key: the tiled workspaces also hold per-grid arrival counters, laid out by
        # (n, size) and zeroed whenever the tables are rebuilt (carle_rnd_ae_bonus / carle_rnd_ae_train)
        geom = (obs.shape[0], h)
        key = (id(self._ws), versions) + geom if fused and self._ws is not None else None
        valid = key is not None and key == self._tables_key
        self._tables_key_next = (fused, versions, geom)
        return valid

    def _commit_tables(self):
        fused, versions, geom = self._tables_key_next
        self._tables_key = (id(self._ws), versions) + geom if fused else None

    def _hip_forward(self, obs, want_outputs=False, reward=None, bonus_out=None, stream=None):
        obs, binary = self._obs(obs, allow_packed=True)
        dev = obs.device
        pb, rb = self._pred_blob.get(dev), self._rand_blob.get(dev)
        valid = self._tables_valid(obs, binary, (self._pred_blob.version, self._rand_blob.version))
        res = hip.rnd2d_bonus_f32(obs, pb, rb, self._ws, want_outputs, binary, reward,
                                  self.reward_scale, bonus_out, stream, valid)
        self._ws = res[-1]
        self._commit_tables()
        return res

    def random_forward(self, obs):
        """Target embedding (N,16) of the frozen random network (mcl.py:319-321)."""
        return self._hip_forward(obs, True)[2]

    def forward(self, obs):
        """Predictor embedding (N,16) (mcl.py:323-330); HIP kernel, no autograd graph."""
        return self._hip_forward(obs, True)[1]

    def get_bonus(self, obs):
        """Per-instance bonus (N,) = mean_k |(target - prediction)^2| (mcl.py:343-352)."""
        return self._hip_forward(obs)[0]

    def get_bonus_only(self, obs):
        return self.get_bonus(obs)

    # ------------------------------------------------------------------ predictor update
    def _torch_loss(self, obs):
        """The same loss through the nn.Sequential modules, WITH autograd (update path only)."""
        with torch.no_grad():
            target = self.random_network(obs)
        prediction = self.predictor(obs)
        return torch.mean(torch.abs((target - prediction) ** 2), dim=[1])

    def update_predictor(self, loss):
        loss.backward()
        # torch's optimizer.step() below consumes the autograd .grad tensors only: the kernel gradient of a window in
        # flight (self._grad) is reduced where it is consumed, in _apply_update
        self._sync_gradients(self._pred_blob.params or list(self.predictor.parameters()), kernel_gradient=False)
        self.optimizer.step()
        self.optimizer.zero_grad(set_to_none=True)
        self.updates += 1

    def get_bonus_update(self, obs):
        """Bonus + immediate predictor update on this observation (mcl.py:354-366): the gradient of
        mean_n(loss_n) through the HIP training kernels (fused / tiled / staged, as ``get_bonus_accumulate``),
        then one Adam step -- no autograd graph."""
        self.predictor.zero_grad(set_to_none=True)           # drops whatever a window had accumulated, as the reference
        self._clear_kept_gradient()
        if self._grad is not None and self._grad_used:
            self._grad.zero_()
            self._grad_used = False
        o = self._obs(obs, allow_packed=True)[0]
        bonus = torch.empty(o.shape[0], dtype=torch.float32, device=o.device)
        self._fused_train(obs, None, bonus, None, batch=1)
        self._apply_update()
        self._after_update()
        return bonus

    def get_bonus_accumulate(self, obs):
        """Bonus now, predictor update every ``batch_size`` calls (mcl.py:368-386)."""
        o = self._obs(obs, allow_packed=True)[0]
        n = o.shape[0]
        bonus = torch.empty(n, dtype=torch.float32, device=o.device)
        if self.buffer_length == 0:
            self.predictor.zero_grad(set_to_none=True)
            self._clear_kept_gradient()
            self.accumulate_loss = 0.0
        self._fused_train(obs, None, bonus, None)
        self.buffer_length += 1
        if self.buffer_length >= self.batch_size:
            self._finish_window()
        return bonus

    def _after_update(self):
        self.optimizer.zero_grad(set_to_none=True)           # RND2D.update_predictor, mcl.py:339
        self.updates += 1

    # AE2D.update_predictor steps WITHOUT zeroing the gradients (mcl.py:498-502): the window's gradient stays in .grad
    # until the next window starts, and a backward in between (a caller's own update_predictor) accumulates on top of it
    _update_keeps_grads = False

    def _clear_kept_gradient(self):
        """Start of a window / of ``get_bonus_update``: the gradient the previous update left behind (AE2D: as ``.grad``
        views of the flat kernel gradient) is dropped from the flat buffer too."""
        if self.__dict__.get("_grad_kept"):
            self._grad.zero_()
            self.__dict__["_grad_kept"] = False

    # ------------------------------------------------------------------ gym surface
    def _bonus_into(self, obs, reward, stream=None):
        """reward += reward_scale * bonus(obs), fused into the bonus kernel (mcl.py:403-405); in
        training mode the predictor's gradient is accumulated on the way (the training kernels)."""
        n = reward.shape[0]
        if self._bonus_buf is None or self._bonus_buf.shape[0] != n or self._bonus_buf.device != reward.device:
            self._bonus_buf = torch.empty(n, dtype=torch.float32, device=reward.device)
        if not self.train_predictor:
            self._hip_forward(obs, False, reward, self._bonus_buf, stream)
            return
        if self.buffer_length == 0:
            self.predictor.zero_grad(set_to_none=True)
            self._clear_kept_gradient()
            self.accumulate_loss = 0.0
        self._fused_train(obs, reward, self._bonus_buf, stream)
        self.buffer_length += 1
        if self.buffer_length >= self.batch_size:
            self._finish_window()

    def _finish_window(self):
        """End of an update window (mcl.py:380-384, 333-341): one Adam step on the window's gradient."""
        self._apply_update()
        self._after_update()
        self.buffer_length = 0

    # ---- optimizer.step() ------------------------------------------------------------------------------------
    def _apply_update(self):
        """optimizer.step() on the gradient of the window: the flat gradient the HIP training kernels accumulated
        goes through ``carle_adam_update_f32`` (torch.optim.Adam's arithmetic in one launch, parameters and the
        kernels' parameter blob updated in place, Adam's moments shared with ``self.optimizer.state`` so that
        ``optimizer.state_dict()`` keeps working).  Whenever a caller's own autograd gradients (``.grad`` left by its
        ``loss.backward()``) or a non-default optimiser are involved, the kernel gradient is added to ``.grad`` and the
        step is the optimiser's own."""
        params = self._pred_blob.params or list(self.predictor.parameters())
        self._sync_gradients(params)
        if self._grad_used and self._hip_adam_applies(params):
            self._hip_adam_step(params)
            return
        if self._grad_used:
            off = 0
            for p in params:
                gslice = self._grad[off:off + p.numel()].view_as(p)
                p.grad = gslice.clone() if p.grad is None else p.grad + gslice
                off += p.numel()
            self._grad.zero_()
            self._grad_used = False
        self.optimizer.step()

    def _sync_gradients(self, params, kernel_gradient=True):
        """Data-parallel predictor (``carle_amd.dist.data_parallel``): the reference's loss is the mean over the WHOLE
        batch (mcl.py:362,376), so under batch sharding the window's gradient is the instance-weighted sum of the
        ranks' gradients -- one all-reduce of the flat gradient (1 117 floats at 64x64, 16.5 k at 256x256) per Adam
        step, on the stream the gradient kernels ran on.  Every rank then takes the same Adam step.
        ``kernel_gradient=False`` (``update_predictor``: torch's own step on a caller's loss) reduces the autograd
        ``.grad`` tensors only and leaves the flat kernel gradient of a window in flight alone: it is weighted and
        reduced exactly once, by the ``_apply_update`` that consumes it."""
        sync = self.__dict__.get("_grad_sync")
        if sync is None:
            return
        import torch.distributed as tdist
        group, weight = sync
        if kernel_gradient and self._grad_used:
            self._grad.mul_(weight)
            tdist.all_reduce(self._grad, group=group)
        grads = [p.grad for p in params if p.grad is not None]       # a caller's own backward
        if grads:
            flat = torch.cat([g.reshape(-1) for g in grads]).mul_(weight)
            tdist.all_reduce(flat, group=group)
            off = 0
            for g in grads:
                g.copy_(flat[off:off + g.numel()].view_as(g))
                off += g.numel()

    def _hip_adam_applies(self, params):
        opt = self.optimizer
        if type(opt) is not torch.optim.Adam or len(opt.param_groups) != 1 or len(params) > 8:
            return False
        g = opt.param_groups[0]
        if g.get("amsgrad") or g.get("maximize") or g.get("weight_decay", 0) != 0 or g.get("differentiable"):
            return False
        if len(g["params"]) != len(params) or any(a is not b for a, b in zip(g["params"], params)):
            return False
        dev = self._grad.device
        return all(p.grad is None and p.requires_grad and p.device == dev and p.dtype == torch.float32
                   and p.is_contiguous() for p in params)

    def _hip_adam_step(self, params):
        dev = self._grad.device
        total = self._grad.numel()
        st = self.__dict__.get("_adam")
        if st is None or st[0].device != dev or st[0].numel() != total:
            st = self.__dict__["_adam"] = [torch.zeros(total, dtype=torch.float32, device=dev),
                                          torch.zeros(total, dtype=torch.float32, device=dev)]
        m, v = st
        # Adam's moments are shared with the torch optimiser: its per-parameter state tensors are views of the flat
        # vectors.  A state somebody else installed (optimizer.load_state_dict, earlier torch steps) is adopted.  The
        # step count is ALWAYS the optimiser's own: a torch-path optimizer.step() in between (update_predictor, the
        # autograd replay) advances it on the shared views, and the bias corrections must follow
        off, step = 0, 0
        for p in params:
            k = p.numel()
            ps = self.optimizer.state.get(p)
            if ps and "exp_avg" in ps:
                step = max(step, int(float(ps["step"])))
                if ps["exp_avg"].data_ptr() != m.data_ptr() + 4 * off:
                    m[off:off + k].copy_(ps["exp_avg"].reshape(-1))
                    v[off:off + k].copy_(ps["exp_avg_sq"].reshape(-1))
            off += k
        off = 0
        for p in params:
            k = p.numel()
            ps = self.optimizer.state.get(p)
            if not ps or "exp_avg" not in ps or ps["exp_avg"].data_ptr() != m.data_ptr() + 4 * off:
                self.optimizer.state[p] = {"step": torch.tensor(float(step)), "exp_avg": m[off:off + k].view_as(p),
                                           "exp_avg_sq": v[off:off + k].view_as(p)}
            off += k
        step += 1
        g = self.optimizer.param_groups[0]
        blob = self._pred_blob.get(dev)
        keep = self._update_keeps_grads
        hip.adam_update_f32([p.data for p in params], blob, self._grad, m, v, g["lr"], g["betas"][0], g["betas"][1],
                            g["eps"], step, zero_grad=not keep)
        off = 0
        for p in params:
            self.optimizer.state[p]["step"].fill_(float(step))        # host tensors
            if keep:
                # as the reference leaves it after AE2D.update_predictor: the window's gradient stays in .grad (views of
                # the flat buffer, no copy) until the next window's zero_grad
                p.grad = self._grad[off:off + p.numel()].view_as(p)
                off += p.numel()
        if keep:
            self.__dict__["_grad_kept"] = True
        self._pred_blob.updated_in_place(dev)
        self._grad_used = False

    # bit-packed observations of any square size that is a multiple of 64 go to the kernels as they are
    # (tiled forward, nets_tiled.hip)
    _packed_any_size = True

    def _tiled_train_buffers(self, o):
        """Workspace, scratch and gradient buffers of the tiled training kernels for the binary observations ``o``
        (as ``_fused_train`` sets them up); returns (predictor blob, random-network blob or None, tables valid,
        table key to commit, gradient scale of one step)."""
        from . import _lib
        lib = _lib.load()
        is_ae = _HIP_NET_FAMILY.get(type(self)) == "ae"
        dev, n, size = o.device, o.shape[0], o.shape[2]
        pb = self._pred_blob.get(dev)
        rb = None if is_ae else self._rand_blob.get(dev)
        need = (lib.carle_ae2d_workspace_floats if is_ae else lib.carle_rnd2d_workspace_floats)(n, size, size)
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.float32, device=dev)
            self._tables_key = None
        pneed = (lib.carle_ae2d_train_scratch_floats if is_ae else lib.carle_rnd2d_train_scratch_floats)(n, size, size)
        if self._partial is None or self._partial.numel() < pneed or self._partial.device != dev:
            self._partial = torch.empty(pneed, dtype=torch.float32, device=dev)
        if self._grad is None or self._grad.device != dev or self._grad.numel() != pb.numel():
            self._grad = torch.zeros(pb.numel(), dtype=torch.float32, device=dev)
        if self._bonus_buf is None or self._bonus_buf.shape[0] != n or self._bonus_buf.device != dev:
            self._bonus_buf = torch.empty(n, dtype=torch.float32, device=dev)
        versions = (self._pred_blob.version,) if is_ae else (self._pred_blob.version, self._rand_blob.version)
        # (n, size) belong to the key: the workspace of the paired training call also holds per-grid arrival counters,
        # laid out by the batch geometry and zeroed whenever the tables are rebuilt (carle_rnd_ae_train)
        key = (id(self._ws), versions, n, size)
        if is_ae:
            grad_scale = 2.0 / (size * size * n * self.batch_size)
        else:
            grad_scale = 2.0 / (self.rnd_dim * n * self.batch_size)
        return pb, rb, self._tables_key == key, key, grad_scale

    def _fused_train(self, obs, reward, bonus_out, stream, batch=None):
        """Bonus + gradient in one library call (carle_rnd2d_train_* / carle_ae2d_train_*): the fused
        forward+backward kernel for 64x64 binary observations, tiled kernels for binary grids of a multiple of 64,
        staged forward + staged backward HIP kernels for every other square size that is a multiple of 8.  Anything
        else raises: there is no autograd path behind the bonus calls."""
        # the gradient kernels implement exactly these classes' networks
        family = _HIP_NET_FAMILY.get(type(self))
        if family is None:
            raise NotImplementedError(f"{type(self).__name__}: the HIP gradient kernels implement the RND2D / AE2D networks of "
                                      "carle/mcl.py only; a subclass with another architecture must train through its own "
                                      "update_predictor(loss)")
        is_ae = family == "ae"
        o, binary = self._obs(obs, allow_packed=True)
        packed = o.dtype == torch.int64
        fused64 = binary and o.shape[2] == 64 and (packed or o.shape[3] == 64)
        size_w = o.shape[3] * 64 if packed else o.shape[3]
        # binary grids with H = W a multiple of 64 (the reference's default 256x256): tiled kernels (nets_tiled.hip)
        tiled = (binary and not fused64 and o.dim() == 4
                 and o.shape[2] == size_w and o.shape[2] % 64 == 0 and o.shape[2] > 64)
        if not fused64 and not tiled:
            if packed:
                o, binary = self._obs(obs)                 # unpacked fp32
                packed = False
            if o.dim() != 4 or o.shape[2] != o.shape[3] or o.shape[2] % 8 or o.shape[2] < 8:
                raise NotImplementedError(f"{type(self).__name__}: predictor updates need square observations with H = W a "
                                          f"multiple of 8 (got {tuple(o.shape)}): the HIP gradient kernels cover nothing "
                                          "else and there is no autograd fallback (set train_predictor=False for the "
                                          "forward bonus alone)")
        uses_tables = fused64 or tiled
        from . import _lib
        lib = _lib.load()
        dev, n, size = o.device, o.shape[0], o.shape[2]
        pb = self._pred_blob.get(dev)
        rb = None if is_ae else self._rand_blob.get(dev)
        need = (lib.carle_ae2d_workspace_floats if is_ae else lib.carle_rnd2d_workspace_floats)(n, size, size)
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.float32, device=dev)
            self._tables_key = None
        if fused64:
            pneed = (lib.carle_ae2d_train_partial_floats if is_ae else lib.carle_rnd2d_train_partial_floats)(n)
        else:
            pneed = (lib.carle_ae2d_train_scratch_floats if is_ae else lib.carle_rnd2d_train_scratch_floats)(n, size, size)
        if self._partial is None or self._partial.numel() < pneed or self._partial.device != dev:
            self._partial = torch.empty(pneed, dtype=torch.float32, device=dev)
        if self._grad is None or self._grad.device != dev or self._grad.numel() != pb.numel():
            self._grad = torch.zeros(pb.numel(), dtype=torch.float32, device=dev)
        versions = (self._pred_blob.version,) if is_ae else (self._pred_blob.version, self._rand_blob.version)
        valid = uses_tables and self._tables_key == (id(self._ws), versions, n, size)
        if is_ae:
            grad_scale = 2.0 / (size * size * n * (self.batch_size if batch is None else batch))
            hip.ae2d_train(o, pb, self._ws, self._partial, self._grad, bonus_out, reward,
                           self.reward_scale, grad_scale, valid, stream, binary=uses_tables)
        else:
            grad_scale = 2.0 / (self.rnd_dim * n * (self.batch_size if batch is None else batch))
            hip.rnd2d_train(o, pb, rb, self._ws, self._partial, self._grad, bonus_out, reward,
                            self.reward_scale, grad_scale, valid, stream, binary=uses_tables)
        self._tables_key = (id(self._ws), versions, n, size) if uses_tables else None
        self._grad_used = True
        return True

    def step(self, action):
        """env.step, then reward += reward_scale * bonus(obs)  (mcl.py:398-407).

        With ``async_reward=True`` the bonus runs on a dedicated HIP stream
        (``inner_env.reward_stream``): the compute-bound reward networks of step t then overlap
        the bandwidth-bound CA kernel of step t+1.  ``obs`` is ready on the caller's stream as
        usual; ``reward`` must be consumed on the reward stream or after ``wait_rewards()``.
        """
        if isinstance(action, torch.Tensor):
            action = action.to(self.inner_env.my_device)
        self.__dict__["_moments_merged"] = False
        if self._can_fuse_step(action):
            return self._fused_step(action)
        obs, reward, done, info = self.env.step(action)
        rs = self._reward_stream()
        if rs is None:
            self._bonus_into(obs, reward)
            return obs, reward, done, info
        rs_raw = rs.fork(obs, reward)                    # reward stream ordered after CA step t
        # EVERYTHING of the bonus runs with the reward stream current: the kernels (explicit stream
        # argument) and every torch-side preparation they consume -- parameter-blob rebuild after an
        # Adam step, gradient / workspace allocation and zero fill, observation format conversion,
        # the window replay and the optimiser -- so producer and consumer are always on one stream
        with rs.context():
            self._bonus_into(obs, reward, rs_raw)
        return obs, reward, done, info

    # -- one-call fast path: this wrapper sits directly on a 64x64 fp32 CARLE, forward only ----
    def _can_fuse_step(self, action):
        env = self.env
        return (type(self) is RND2D and env is self.inner_env and not self.train_predictor and
                isinstance(action, torch.Tensor) and env.height == 64 and
                getattr(env, "storage", "f32") == "f32" and not env.logging)

    def _fused_step(self, action):
        env = self.inner_env
        dev = env.my_device
        pb, rb = self._pred_blob.get(dev), self._rand_blob.get(dev)
        from . import _lib
        need = _lib.load().carle_rnd2d_workspace_floats(env.instances, 64, 64)
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.float32, device=dev)
            self._tables_key = None
        n = env.instances
        # n + 1 slots: the fused kernel leaves the bonus of the all-dead grid behind the batch
        if self._step_bonus is None or self._step_bonus.shape[0] != n + 1 or self._step_bonus.device != dev:
            self._step_bonus = torch.empty(n + 1, dtype=torch.float32, device=dev)
        versions = (self._pred_blob.version, self._rand_blob.version)
        valid = self._tables_key == (id(self._ws), versions)
        rs = self._reward_stream()

        merge = self.__dict__.get("_final_reward_stats") is not None     # this wrapper is outermost
        # one launch per step (CARLE_STEP_DEFERRED): the reward moments of step t are merged by the next call on the
        # status block from per-workgroup partial sums kept here (two buffers, one per flag slot)
        partials = None
        if merge:
            pneed = _lib.load().carle_rnd_step_partials_doubles(n)
            partials = self.__dict__.get("_mom_partials")
            if partials is None or partials.numel() < pneed or partials.device != dev:
                env.settled_status()                         # nothing may still point into the old buffer
                partials = self.__dict__["_mom_partials"] = torch.zeros(pneed, dtype=torch.float64, device=dev)

        bits_bufs = None
        # a reward net above reads the observation bit-packed -- and so does this step itself, next time (CARLE._packed_state_of)
        if self.__dict__.get("_hand_bits") or (getattr(env, "packed_state", False) and n > 16):
            bits_bufs = self.__dict__.get("_bits_bufs")
            if bits_bufs is None or bits_bufs[0].shape[0] != n or bits_bufs[0].device != dev:
                bits_bufs = self.__dict__["_bits_bufs"] = [torch.empty(n, 1, 64, 1, dtype=torch.int64, device=dev) for _ in (0, 1)]

        def launch(u_in, act, out, reward, rule, status, slot, publish):
            main_raw = hip._stream(u_in)
            if rs is None:
                ev, rs_raw = None, None
            else:
                rs_raw = rs.raw
                ev = rs.event_for_fork(main_raw, out, reward)
            fusable = act.shape[3] % 8 == 0     # (the library fuses -- and can merge the moments -- only with an 8-aligned window)
            bits = bits_bufs[slot] if bits_bufs is not None else None
            # the state as the previous step left it bit-packed (only if ``u_in`` is that very tensor, untouched): 512 B
            # instead of 16 KiB per grid
            state_bits = env._packed_state_of(u_in, bits) if fusable and n > 16 else None
            if hip.env_step_f32_rnd(u_in, act, out, reward, rule, status, slot, publish, pb, rb, self._ws,
                                    self._step_bonus, self.reward_scale, valid, ev, main_raw, rs_raw,
                                    merge and fusable, deferred=True, partials=partials, bits_out=bits, bits_in=state_bits):
                env.__dict__["_obs_bits"] = (weakref.ref(out), out._version, bits)
            self.__dict__["_moments_merged"] = merge and fusable
            if merge and fusable:
                env.__dict__["_moments_pending"] = True
        self.__dict__["_moments_merged"] = False
        result = env._step(action, launch)
        self._tables_key = (id(self._ws), versions)
        return result

    def _reward_stream(self):
        if not self.async_reward:
            return None
        env = self.inner_env
        rs = env.__dict__.get("reward_stream")
        if rs is None:
            from .streams import RewardStream
            rs = RewardStream(env.my_device)
            env.__dict__["reward_stream"] = rs
        return rs

    def wait_rewards(self):
        """Make the caller's current stream wait for all reward work enqueued so far."""
        rs = self.inner_env.__dict__.get("reward_stream")
        if rs is not None:
            rs.join()

    def reset(self):
        obs = self.env.reset()
        self.to(self.inner_env.my_device)
        self.updates = 0
        return obs


class AE2D(RND2D):
    """Convolutional auto-encoder reconstruction-error bonus (mcl.py:424-556)."""

    def __init__(self, env, **kwargs):
        super().__init__(env, **kwargs)
        # assigned after the optimiser exists, as in the reference: Adam really runs at 6e-2
        self.learning_rate = 9e-2
        self.my_name = "AE2D"

    def initialize_random_network(self):
        pass

    def initialize_predictor(self):
        self.predictor = nn.Sequential(
            nn.Conv2d(1, 4, 3, padding=1, stride=1),
            nn.Dropout(p=0.1),
            nn.ReLU(),
            nn.MaxPool2d(2, 2, padding=0),
            nn.Conv2d(4, 2, 3, padding=1, stride=1),
            nn.Dropout(p=0.1),
            nn.ReLU(),
            nn.MaxPool2d(2, 2, padding=0),
            nn.ConvTranspose2d(2, 1, 4, padding=1, stride=2),
            nn.Dropout(p=0.1),
            nn.ReLU(),
            nn.ConvTranspose2d(1, 1, 4, padding=1, stride=2),
            nn.Dropout(p=0.1),
            nn.Sigmoid())
        self.optimizer = torch.optim.Adam(self.predictor.parameters(), lr=self.learning_rate)
        self._pred_blob = _ParamBlob(self.predictor)

    def set_grad(self):
        for p in self.predictor.parameters():
            p.requires_grad = True

    def set_no_grad(self):
        for p in self.predictor.parameters():
            p.requires_grad = False

    def _can_leave_moments(self):
        """This AE2D is the top of AE2D(RND2D(CARLE)) at 64x64 (train_mcl.py:36-40), forward passes: its bonus kernel finishes
        the step's rewards and can leave their moments for the running statistics (carle_ae2d_bonus_b1_stats)."""
        rnd, env = self.env, self.inner_env
        return (type(self) is AE2D and type(rnd) is RND2D and rnd.env is env and not self.async_reward and not rnd.async_reward
                and not self.train_predictor and not rnd.train_predictor and env.height == env.width == 64
                and getattr(env, "storage", "f32") == "f32")

    def _hip_forward(self, obs, want_outputs=False, reward=None, bonus_out=None, stream=None):
        obs, binary = self._obs(obs, allow_packed=True)
        pb = self._pred_blob.get(obs.device)
        valid = self._tables_valid(obs, binary, (self._pred_blob.version,))
        status = self.inner_env.__dict__.get("_status")
        if (reward is not None and not want_outputs and stream is None and obs.dtype == torch.int64 and obs.shape[2] == 64
                and obs.shape[3] == 1 and status is not None and self.__dict__.get("_final_reward_stats") is not None):
            # the rewards this kernel finishes are the step's final ones and a RunningRewardStats is attached to this wrapper:
            # their moments leave as per-workgroup partial sums, merged by the next launch on the status block
            from . import _lib
            lib = _lib.load()
            n, dev = obs.shape[0], obs.device
            need = lib.carle_ae2d_workspace_floats(n, 64, 64)
            if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
                self._ws = torch.empty(max(need, 1), dtype=torch.float32, device=dev)
                valid = False
            pneed = lib.carle_ae2d_stats_partials_doubles(n)
            partials = self.__dict__.get("_mom_partials64")
            if partials is None or partials.numel() < pneed or partials.device != dev:
                self.inner_env.settled_status()              # nothing may still point into the old buffer
                partials = self.__dict__["_mom_partials64"] = torch.zeros(pneed, dtype=torch.float64, device=dev)
            bonus = bonus_out if bonus_out is not None else torch.empty(n, dtype=torch.float32, device=dev)
            hip.ae2d_bonus_b1_stats(obs, pb, self._ws, bonus, reward, self.reward_scale, valid, partials, status)
            self._commit_tables()
            self.__dict__["_moments_merged"] = True
            self.inner_env.__dict__["_moments_pending"] = True
            return bonus, self._ws
        res = hip.ae2d_bonus_f32(obs, pb, self._ws, want_outputs, binary, reward, self.reward_scale,
                                 bonus_out, stream, valid)
        self._ws = res[-1]
        self._commit_tables()
        return res

    # -- stacked on an RND2D, forward only, tiled sizes: both bonuses in one launch (carle_rnd_ae_bonus) ----------
    def _can_pair_step(self):
        """0: the wrappers step one after the other; 1: paired forward (both without predictor updates); 2: paired
        training step (both with predictor updates through the HIP gradient kernels)."""
        rnd, env = self.env, self.inner_env
        if not (type(self) is AE2D and type(rnd) is RND2D and rnd.env is env and not self.async_reward
                and not rnd.async_reward and env.height == env.width and env.height % 64 == 0):
            return 0
        if env.height == 64:
            # 64x64: the wrappers step one after the other -- two launches: RND2D's one-launch step (which also hands the
            # observation over bit-packed) and AE2D's bonus kernel.  (One kernel for the three was built in round 5 and
            # measured slower: EXPERIMENTS.md 5.4.)
            return 0
        if not self.train_predictor and not rnd.train_predictor:
            return 1
        if self.train_predictor and rnd.train_predictor:
            return 2
        return 0

    def step(self, action):
        """AE2D(RND2D(CARLE)) -- the stack of train_mcl.py:34-38 / eval.py:83-85 -- at the reference's default geometry
        (binary grids, H = W a multiple of 64 above 64) without predictor updates: the environment step, then BOTH
        wrappers' bonuses and reward updates in ONE launch (the tile workgroups of the two nets side by side, the last
        tile of a grid finishes it) instead of four -- a step there is bound by its dependent dispatches.  Bit-identical to the
        two wrappers stepping one after the other."""
        mode = self._can_pair_step()
        if not mode:
            return super().step(action)
        rnd, env = self.env, self.inner_env
        if isinstance(action, torch.Tensor):
            action = action.to(env.my_device)
        rnd.__dict__["_moments_merged"] = False
        self.__dict__["_moments_merged"] = False
        obs, reward, done, info = env.step(action)
        o, binary = self._obs(obs, allow_packed=True)
        if not binary or o.dim() != 4:
            rnd._bonus_into(obs, reward)
            self._bonus_into(obs, reward)
            return obs, reward, done, info
        dev, n, size = o.device, o.shape[0], o.shape[2]
        needs = self.__dict__.get("_pair_needs")             # (n, size) -> workspace floats of both nets, moment pairs: one
        if needs is None or needs[0] != (n, size):           # library call each per geometry, not per step (host-bound step)
            from . import _lib
            lib = _lib.load()
            needs = self.__dict__["_pair_needs"] = ((n, size), lib.carle_rnd2d_workspace_floats(n, size, size),
                                                    lib.carle_ae2d_workspace_floats(n, size, size),
                                                    lib.carle_rnd_ae_partials_doubles(n, size, size))
        partials, status = self._pair_moment_buffers(needs[3], dev)
        if mode == 2:
            return self._paired_train_step(o, obs, reward, done, info, partials, status)
        for w, need in ((rnd, needs[1]), (self, needs[2])):
            if w._ws is None or w._ws.numel() < need or w._ws.device != dev:
                w._ws = torch.empty(max(need, 1), dtype=torch.float32, device=dev)
                w._tables_key = None
            if w._bonus_buf is None or w._bonus_buf.shape[0] != n or w._bonus_buf.device != dev:
                w._bonus_buf = torch.empty(n, dtype=torch.float32, device=dev)
        pb, rb, ab = rnd._pred_blob.get(dev), rnd._rand_blob.get(dev), self._pred_blob.get(dev)
        rvalid = rnd._tables_valid(o, True, (rnd._pred_blob.version, rnd._rand_blob.version))
        avalid = self._tables_valid(o, True, (self._pred_blob.version,))
        hip.rnd_ae_bonus(o, pb, rb, rnd._ws, rnd._bonus_buf, rnd.reward_scale, rvalid, ab, self._ws, self._bonus_buf,
                         self.reward_scale, avalid, reward, partials=partials, status=status)
        rnd._commit_tables()
        self._commit_tables()
        self._pair_moments_left(partials)
        return obs, reward, done, info

    def _pair_moment_buffers(self, need, dev):
        """RunningRewardStats attached to this pair (``attach``): the pair's finish kernel leaves per-workgroup partial sums
        of the step's final rewards and registers them in the environment's status block; the next environment step (or
        ``settled_status``) merges them -- no moments launch per step."""
        if self.__dict__.get("_final_reward_stats") is None:
            return None, None
        env = self.inner_env
        partials = self.__dict__.get("_mom_partials")
        if partials is None or partials.numel() < need or partials.device != dev:
            env.settled_status()                             # nothing may still point into the old buffer
            partials = self.__dict__["_mom_partials"] = torch.zeros(need, dtype=torch.float64, device=dev)
        return partials, env._status

    def _pair_moments_left(self, partials):
        if partials is not None:
            self.__dict__["_moments_merged"] = True
            self.inner_env.__dict__["_moments_pending"] = True

    def _paired_train_step(self, o, obs, reward, done, info, partials=None, status=None):
        """Training mode of the pair (train_mcl.py:64-69): what ``RND2D._bonus_into`` followed by ``AE2D._bonus_into``
        do -- window bookkeeping, bonus + reward + gradient accumulation, Adam at the end of each wrapper's window --
        with the kernels of both wrappers in four launches (carle_rnd_ae_train) instead of nine.  Bit-identical."""
        rnd = self.env
        for w in (rnd, self):
            if w.buffer_length == 0:
                w.predictor.zero_grad(set_to_none=True)
                w._clear_kept_gradient()
                w.accumulate_loss = 0.0
        pb, rb, rvalid, rkey, rgs = rnd._tiled_train_buffers(o)
        ab, _, avalid, akey, ags = self._tiled_train_buffers(o)
        hip.rnd_ae_train(o, pb, rb, rnd._ws, rnd._partial, rnd._grad, rnd._bonus_buf, rnd.reward_scale, rgs, rvalid,
                         ab, self._ws, self._partial, self._grad, self._bonus_buf, self.reward_scale, ags, avalid, reward,
                         partials=partials, status=status)
        rnd._tables_key, self._tables_key = rkey, akey
        self._pair_moments_left(partials)
        for w in (rnd, self):
            w._grad_used = True
            w.buffer_length += 1
            if w.buffer_length >= w.batch_size:
                w._finish_window()
        return obs, reward, done, info

    def random_forward(self, obs):
        raise AttributeError("AE2D has no random network (mcl.py:433-434)")

    def forward(self, obs):
        """Reconstruction (instances,1,H,W) (mcl.py:436-443); HIP kernel, no autograd graph."""
        recon = self._hip_forward(obs, True)[1]
        return recon.reshape(self.inner_env.instances, 1, self.inner_env.height,
                             self.inner_env.width)

    def _torch_loss(self, obs):
        prediction = self.predictor(obs).reshape(obs.shape)
        return torch.mean(torch.abs((obs - prediction) ** 2), dim=[1, 2, 3])

    def update_predictor(self, loss):
        loss.backward()
        self._sync_gradients(self._pred_blob.params or list(self.predictor.parameters()), kernel_gradient=False)
        self.optimizer.step()

    _update_keeps_grads = True

    def _after_update(self):
        self.updates += 0                    # AE2D.update_predictor neither zeroes grads nor
        # counts updates (mcl.py:498-502); grads are zeroed at the start of the next window


def _reward_live_grids(w, reward, err, offset, sign):
    """``reward += reward_scale * (offset + sign * err)``, zero where the instance has no live cell, and
    ``live_cells = obs.sum()`` (mcl.py:686-697, 715-726: a Python loop with one host synchronisation per instance
    there) -- one launch on the per-instance sums of the universe (carle_bonus_nonempty_f32), no synchronisation."""
    sums = _live_sums(w.inner_env, reward.device)
    total = torch.empty((), dtype=torch.float32, device=reward.device)
    hip.bonus_nonempty_f32(reward, err, sums, offset, sign, w.reward_scale, total)
    w.live_cells = total


class PredictionBonus(AE2D):
    """Frame-prediction reward (mcl.py:558-699): the AE2D predictor is applied to the OLDEST frame of a short buffer
    (``prediction_steps`` = 5 steps back once the buffer is full; the frame itself on the first call) and compared with
    the newest frame; ``reward += reward_scale * (0.1 - error)``, zero for instances without a live cell.

    HIP path: binary frames with H = W a multiple of 64 (what ``CARLE.step`` returns at 64 / 128 / 256 / 512) are kept
    bit-packed and go through AE2D's fused 64x64 / tiled kernels with the error term and the output layer's gradient
    taken against the target frame (``carle_ae2d_predict_bonus_b1`` / ``carle_ae2d_predict_train_b1``); any other H = W
    multiple of 8 and non-binary frames through the per-layer kernels (``..._f32``); the window's gradient goes through
    ``carle_adam_update_f32`` as AE2D's.  Eval-mode semantics (Dropout = identity) as everywhere.  Frames are private
    copies: the reference keeps the tensor objects ``env.step`` returned (a new tensor per step there, env.py:229)."""

    def __init__(self, env, **kwargs):
        super().__init__(env, **kwargs)
        # assigned after the optimiser exists, as in the reference (mcl.py:563): Adam still runs at 6e-2
        self.learning_rate = 3e-4
        self.my_name = "PredictionBonus"
        self.prediction_steps = 5
        self.grid_buffer = []
        self.__dict__["_scratch"] = None

    def _frame(self, obs):
        """A private copy of the observation for the frame buffer: bit-packed (512 bytes per 64x64 grid) when it is binary
        with H = W a multiple of 64 -- whatever CARLE.step returned -- fp32 otherwise."""
        o, binary = self._obs(obs, allow_packed=True)
        if o.dtype == torch.int64:
            return o.clone()                     # (the environment's own buffers are rewritten by later steps)
        if o.dim() != 4 or o.shape[2] != o.shape[3] or o.shape[2] % 8 or o.shape[2] < 8:
            raise NotImplementedError("PredictionBonus: the HIP kernels need square grids with H = W a multiple of 8 "
                                      f"(got {tuple(o.shape)})")
        if binary and o.shape[2] % 64 == 0:
            return hip.pack_f32_b1(o)
        return o.clone() if o is obs or o.data_ptr() == obs.data_ptr() else o

    def _predict(self, obs, with_grad, batch=None):
        """mcl.py:581-605 (+ the gradient of mean_n(loss) / batch for the update paths)."""
        self.grid_buffer.append(self._frame(obs))
        first, target = self.grid_buffer[0], self.grid_buffer[-1]
        if first.dtype != target.dtype:          # a caller mixed binary and non-binary frames: per-layer kernels on fp32
            first, target = [hip.unpack_b1_f32(f) if f.dtype == torch.int64 else f for f in (first, target)]
        packed = first.dtype == torch.int64
        dev, n, size = first.device, first.shape[0], first.shape[2]
        pb = self._pred_blob.get(dev)
        grad = None
        if with_grad:
            if self._grad is None or self._grad.device != dev or self._grad.numel() != pb.numel():
                self._grad = torch.zeros(pb.numel(), dtype=torch.float32, device=dev)
            grad = self._grad
        gs = 2.0 / (size * size * n * (self.batch_size if batch is None else batch))
        key = (id(self._ws), (self._pred_blob.version,), n, size)
        valid = packed and self._ws is not None and self._tables_key == key
        bonus, _, self._ws, self.__dict__["_scratch"] = hip.ae2d_predict(
            first, target, pb, self._ws, self.__dict__.get("_scratch"), grad, gs, tables_valid=valid)
        # packed frames: the conv1 tables of AE2D's fused / tiled kernels now sit in the workspace; fp32 frames: the
        # per-layer kernels overwrote it
        self._tables_key = (id(self._ws), (self._pred_blob.version,), n, size) if packed else None
        if with_grad:
            self._grad_used = True
        if len(self.grid_buffer) > self.prediction_steps:
            self.grid_buffer.pop(0)
        return bonus

    def get_bonus(self, obs):
        return self._predict(obs, False)

    def get_bonus_only(self, obs):
        return self.get_bonus(obs)

    def get_bonus_update(self, obs):
        """mcl.py:644-656: zero_grad, the loss of this call, one Adam step on the gradient of its batch mean."""
        self.predictor.zero_grad(set_to_none=True)
        self._clear_kept_gradient()
        if self._grad is not None and self._grad_used:
            self._grad.zero_()
            self._grad_used = False
        bonus = self._predict(obs, True, batch=1)
        self._apply_update()
        self._after_update()
        return bonus

    def get_bonus_accumulate(self, obs):
        """mcl.py:658-677: the loss now, an Adam step on the window's mean gradient every ``batch_size`` calls."""
        if self.buffer_length == 0:
            self.predictor.zero_grad(set_to_none=True)
            self._clear_kept_gradient()
            self.accumulate_loss = 0.0
        bonus = self._predict(obs, True)
        self.buffer_length += 1
        if self.buffer_length >= self.batch_size:
            self._finish_window()
        return bonus

    def _bonus_into(self, obs, reward, stream=None):
        raise NotImplementedError("PredictionBonus forms its reward in step() (0.1 - error, mcl.py:686), not by "
                                  "reward += scale * bonus")

    def step(self, action):
        """mcl.py:679-699."""
        if isinstance(action, torch.Tensor):
            action = action.to(self.inner_env.my_device)
        obs, reward, done, info = self.env.step(action)
        err = self.get_bonus_accumulate(obs) if self.train_predictor else self.get_bonus(obs)
        _reward_live_grids(self, reward, err, 0.1, -1.0)
        return obs, reward, done, info


class SurpriseBonus(AE2D):
    """Reconstruction error as the reward, zero for instances without a live cell (mcl.py:701-728); the bonus and the
    predictor updates are AE2D's (fused / tiled / staged HIP kernels)."""

    def __init__(self, env, **kwargs):
        super().__init__(env, **kwargs)
        self.ca_steps = 3

    def step(self, action):
        if isinstance(action, torch.Tensor):
            action = action.to(self.inner_env.my_device)
        obs, reward, done, info = self.env.step(action)
        err = self.get_bonus_accumulate(obs) if self.train_predictor else self.get_bonus(obs)
        _reward_live_grids(self, reward, err, 0.0, 1.0)
        return obs, reward, done, info


# which gradient-kernel family serves a wrapper class (RND2D._fused_train): exact types only
_HIP_NET_FAMILY = {RND2D: "rnd", AE2D: "ae", SurpriseBonus: "ae"}


# ---------------------------------------------------------------------------------------------
# hand-coded reward wrappers: one HIP reduction pass per step
# (carle_instance_sums_f32); the remaining arithmetic is on (N,)-sized tensors
# ---------------------------------------------------------------------------------------------

def _as_device_f32(x, device):
    if not isinstance(x, torch.Tensor):
        if getattr(x, "materialize", None) is not None:          # agents.DeviceRandomAction
            return x.materialize()
        x = torch.as_tensor(x, dtype=torch.float32)
    return x.to(device, torch.float32)


class ParsimonyBonus(Motivator):
    """Scales the reward down when many cells are toggled (mcl.py:86-105).

    reward <- 100 * reward / max(sum(action_i), 100), per instance.  (The reference divides the
    (N,1) reward by an (N,) vector, which broadcasts to (N,N) for N > 1; this implementation keeps
    the per-instance meaning and is identical for N == 1, the only case the reference exercises.)
    """

    def __init__(self, env, **kwargs):
        super().__init__(env, **kwargs)
        self.parsimony_threshold = 128

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        a = _as_device_f32(action, self.inner_env.my_device)
        while a.dim() < 4:
            a = a.unsqueeze(0)
        toggles = hip.instance_sums_f32(a.contiguous())[:, 0:1]
        if toggles.shape[0] != reward.shape[0]:
            toggles = toggles.expand(reward.shape[0], 1)
        reward = 100.0 * reward / torch.clamp(toggles, min=100.0)
        return obs, reward, done, info


def _cached_device(owner, name, dev):
    """Device fp32 copy of a host-side weight plane attribute, re-made only when the attribute was replaced or
    modified in place (the reference keeps these planes on the CPU and moves them every step, mcl.py:218-224)."""
    t = getattr(owner, name)
    key = (id(t), t._version, str(dev))
    cache = owner.__dict__.setdefault("_dev_cache", {})
    hit = cache.get(name)
    if hit is None or hit[0] != key:
        hit = cache[name] = (key, t.to(dev, torch.float32).contiguous())
    return hit[1]


class MorphoBonus(Motivator):
    """A bonus for matching a desired body / machine plan (mcl.py:107-195): the grid before the step, toggled by the
    action, is correlated with the target patterns (each RLE pattern in six orientations: 8 x 8 kernels, -1 on dead cells,
    the live cells sharing a weight of 15); ``reward += reward_scale * (max + min)`` of the correlation per instance.

    The reference names ``glider_1.rle`` / ``glider_2.rle`` beside ``mcl.py`` as the default patterns (mcl.py:140-141) and
    does not contain them, so its constructor raises; this package ships the two glider phases under those names
    (``carle_amd/glider_1.rle``, ``glider_2.rle``), and ``add_rle_pattern`` takes any other file.  As in the reference the
    action must have the universe's shape (mcl.py:172 subtracts them: 64 x 64 environments with the default window).
    HIP path: ``carle_morpho_match_f32`` (correlation, maxima / minima and the reward update; two launches)."""

    def __init__(self, env, **kwargs):
        super().__init__(env, **kwargs)
        self.use_grad = kwargs["use_grad"] if "kwargs" in kwargs.keys() else False      # (sic, mcl.py:115)
        self.my_name = "MorphoBonus"
        self.reward_scale = 1.0
        self.target_patterns = torch.Tensor().to(self.my_device)
        self.add_default_patterns()
        if self.use_grad:
            for param in self.inner_env.parameters():
                param.requires_grad = True
            self.target_patterns.requires_grad = True

    def add_default_patterns(self):
        file_path = os.path.split(os.path.abspath(__file__))[0]
        self.add_rle_pattern(os.path.join(file_path, "glider_1.rle"))
        self.add_rle_pattern(os.path.join(file_path, "glider_2.rle"))

    def add_rle_pattern(self, rle_path, dim=8):
        """mcl.py:144-167.  (``read_rle`` also sets the environment's rule from the file's header, as there.)"""
        pad_1 = nn.ZeroPad2d((1, 1, 2, 1))
        my_rle = self.inner_env.read_rle(rle_path)
        pattern = pad_1(self.inner_env.rle_to_grid(my_rle))[:dim, :dim]
        pattern[pattern == 0] = -1
        pattern = pattern.unsqueeze(0).unsqueeze(0).to(self.my_device)
        pattern[pattern == 1] *= 15. / pattern[pattern == 1].sum()          # normalize kernel
        for oriented in (pattern, pattern.flip(2), pattern.flip(3), pattern.transpose(2, 3).flip(2),
                         pattern.transpose(2, 3).flip(3), pattern.transpose(2, 3)):
            self.target_patterns = torch.cat([self.target_patterns, oriented])

    def step(self, action):
        inner = self.inner_env
        dev = inner.my_device
        grid = _as_device_f32(inner.universe, dev)
        if grid.dim() == 3:
            grid = grid.unsqueeze(1)
        a = _as_device_f32(action, dev)
        while a.dim() < 4:
            a = a.unsqueeze(0)
        if a.shape != grid.shape:
            a = a.expand(grid.shape)               # torch broadcasting, as ``universe - action`` (raises where that does)
        patterns = self.target_patterns.detach().to(dev, torch.float32).contiguous()
        if patterns.dim() != 4 or tuple(patterns.shape[1:]) != (1, 8, 8):
            raise NotImplementedError("MorphoBonus: the HIP kernel takes 8 x 8 patterns (add_rle_pattern's default dim)")
        match, self.__dict__["_partial"] = hip.morpho_match_f32(grid.contiguous(), a.contiguous(), patterns,
                                                                partial=self.__dict__.get("_partial"))
        obs, reward, done, info = self.env.step(action)
        reward += self.reward_scale * (match[:, 0:1] + match[:, 1:2])
        return obs, reward, done, info

    def reset(self):
        obs = self.env.reset()
        inner = self.inner_env
        # imagine a little seed for nucleating patterns (mcl.py:189-190: CPU generator, then moved) -- always of the
        # UNIVERSE's shape (a packed observation has another one)
        seed = (torch.rand(inner.instances, 1, inner.height, inner.width) > 0.995).to(self.my_device).float()
        if getattr(inner, "storage", "f32") == "f32" and obs.dtype == torch.float32 and obs is inner.universe:
            obs += seed                                      # the reference's in-place add: obs IS the state tensor
        else:
            # uint8 / bit-packed state (or a wrapper below that returned another tensor): through the setter, so that the
            # state the step kernels read holds the seed -- not a lazily unpacked fp32 view of it
            inner.universe = _as_device_f32(inner.universe, self.my_device) + seed
            obs = inner.get_observation()
        return obs


class CornerBonus(Motivator):
    """Masked-sum reward / punishment regions (mcl.py:197-231)."""

    def __init__(self, env, **kwargs):
        super().__init__(env, **kwargs)
        self.my_name = "CornerBonus"
        self.reward_scale = 1.0
        h, w = self.inner_env.height, self.inner_env.width
        self.reward_mask = torch.zeros(1, 1, h, w)
        self.punish_mask = torch.zeros(1, 1, h, w)
        self.reward_mask[:, :, :16, :16] = 1.0
        for ii in range(96):
            self.reward_mask[:, :, ii - 4:ii + 4, ii - 4:ii + 4] = 1.0
        self.punish_mask[:, :, -64:, -64:] = -1.0
        self.punish_mask[:, :, :64, -64:] = -1.0

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        dev = self.inner_env.my_device
        sums = hip.instance_sums_f32(_as_device_f32(obs, dev), _cached_device(self, "reward_mask", dev),
                                     _cached_device(self, "punish_mask", dev))
        reward += self.reward_scale * sums[:, 1:2]
        reward += self.reward_scale * sums[:, 2:3]
        return obs, reward, done, info


def _live_sums(inner, dev):
    """(N,4) sums of the inner universe, column 0 = live cells per instance (mcl.py:769,832).  Cached on the inner
    environment for the tensor object they were computed from: a fresh universe tensor per step (env.py:229), so
    stacked wrappers (eval.py:86-87: SpeedDetector and PufferDetector) reduce the grid once, not once each."""
    u = inner.universe
    hit = inner.__dict__.get("_live_cache")
    if hit is not None and hit[0]() is u and hit[1] == u._version:
        return hit[2]
    sums = hip.instance_sums_f32(_as_device_f32(u, dev))
    inner.__dict__["_live_cache"] = (weakref.ref(u), u._version, sums)
    return sums


def _scalar_group(w):
    """``(group,)`` when this wrapper's batch-global scalars are exchanged over a process group (batch sharding,
    carle_amd.dist.batch_scalar_group), None for a single process / a per-rank stack."""
    import torch.distributed as tdist
    if not (tdist.is_available() and tdist.is_initialized()):
        return None
    from . import dist as cdist
    return cdist.batch_scalar_group(w)


def _exchange_buffer(w, dev):
    """fp64[4] on the device: [sum v^2, had a previous centre of mass, live cells, action had a non-zero] of this shard,
    all-reduced in place (carle_wrapper_scalars_local_f32 / _apply_f32)."""
    e = w.__dict__.get("_exch")
    if e is None or e.device != dev:
        e = w.__dict__["_exch"] = torch.zeros(4, dtype=torch.float64, device=dev)
    return e


def _all_reduce_sum(t, grp):
    import torch.distributed as tdist
    tdist.all_reduce(t, group=grp[0])


class SpeedDetector(Motivator):
    """Rewards centre-of-mass motion outside the action window (mcl.py:730-799).

    Per step: ONE reduction pass over the observation (carle_instance_sums_f32 with the two masked weight planes) and
    one single-block kernel (carle_speed_update_f32: centre of mass, velocity, the batch-coupled speed, reward +=
    speed) -- no host synchronisation, no per-step tensor allocation (the previous centre of mass, velocity, speed and
    live-cell vector live in fixed device buffers, so a StepGraph can replay the step).  ``center_of_mass`` /
    ``velocity`` / ``speed`` / ``live_cells`` are views of those buffers, with the reference's first-step values
    (``center_of_mass`` None before the first step, ``speed`` None before the second)."""

    def __init__(self, env, **kwargs):
        super().__init__(env, **kwargs)
        self.reward_scale = 1.0
        self.center_of_mass = None
        self.speed_modulator = 32.0
        mass_weight_w = torch.arange(self.inner_env.height).reshape(1, -1)
        mass_weight_h = torch.arange(self.inner_env.width).reshape(-1, 1)
        action_mask = torch.ones(1, 1, self.action_height, self.action_width)
        action_mask = self.inner_env.action_padding(action_mask)
        action_mask = torch.ones_like(action_mask) - action_mask
        self.mass_weight_h = (mass_weight_h * action_mask).to(torch.float32)
        self.mass_weight_w = (mass_weight_w * action_mask).to(torch.float32)
        self.growing_steps = 0
        self.smooth_velocity = None
        self.speed = None
        self.velocity = torch.tensor([self.inner_env.instances, 1, 0., 0.])
        self.live_cells = None
        self.__dict__["_dev_state"] = None      # (com (2,N), velocity (2,N), speed (1), live (N), has_prev int32[1])
        self.__dict__["_steps_seen"] = 0

    def _dev_buffers(self, n, dev):
        st = self.__dict__["_dev_state"]
        if st is None or st[0].shape[1] != n or st[0].device != dev:
            st = (torch.zeros(2, n, dtype=torch.float32, device=dev), torch.zeros(2, n, dtype=torch.float32, device=dev),
                  torch.zeros(1, dtype=torch.float32, device=dev), torch.zeros(n, dtype=torch.float32, device=dev),
                  torch.zeros(1, dtype=torch.int32, device=dev))
            if self.center_of_mass is not None:             # a caller-assigned centre of mass carries over
                st[0].copy_(torch.as_tensor(self.center_of_mass, dtype=torch.float32).reshape(2, n))
                st[4].fill_(1)
            self.__dict__["_dev_state"] = st
        return st

    def _graph_state(self):
        """Device tensors a StepGraph capture must snapshot around its warm-up steps."""
        st = self.__dict__["_dev_state"]
        return list(st) if st is not None else []

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        self._apply(obs, reward)
        return obs, reward, done, info

    def _apply(self, obs, reward):
        """The wrapper's own part of a step: reduction + update behind the step of the stack below."""
        inner = self.inner_env
        dev = inner.my_device
        o = _as_device_f32(obs, dev)
        # sum(obs * w) over the whole grid with the action window masked out of the weights;
        # live cells are counted on the inner universe (= obs for the fp32 storage) as in the reference (mcl.py:769)
        sums = hip.instance_sums_f32(o, _cached_device(self, "mass_weight_h", dev), _cached_device(self, "mass_weight_w", dev))
        same = inner.__dict__.get("universe") is obs
        if same:
            inner.__dict__["_live_cache"] = (weakref.ref(obs), obs._version, sums)
        live4 = None if same else _live_sums(inner, dev)
        com, vel, speed, live, has_prev = self._dev_buffers(sums.shape[0], dev)
        if not reward.is_contiguous():                      # (never for the rewards this package produces)
            raise ValueError("SpeedDetector needs a contiguous reward tensor")
        grp = _scalar_group(self)
        if grp is None:
            hip.speed_update_f32(sums, live4, com, has_prev, vel, speed, live, reward.reshape(-1))
        else:
            # batch sharding: the speed couples ALL instances (mcl.py:789) -- this shard's sum of squares, one all-reduce,
            # the global speed onto this shard's rewards
            exch = _exchange_buffer(self, dev)
            hip.wrapper_scalars_local(sums, live4, com, has_prev, vel, live, None, exch, 1)
            _all_reduce_sum(exch, grp)
            hip.wrapper_scalars_apply(exch, 1, speed, None, None, None, reward.reshape(-1))
        self._after_update(com, vel, speed, live)

    def _after_update(self, com, vel, speed, live):
        """The attributes the reference's step leaves (mcl.py:781-797), as views of the device buffers."""
        seen = self.__dict__["_steps_seen"]
        if seen == 0 and self.center_of_mass is None:
            self.center_of_mass = com                       # mcl.py:781-782: stored, nothing else
        else:
            self.speed = speed[0]
            self.velocity = vel
            self.center_of_mass = com
        self.__dict__["_steps_seen"] = seen + 1
        self.live_cells = live


class PufferDetector(Motivator):
    """Rewards sustained growth in the absence of actions (mcl.py:804-853).

    The reference reads the live-cell total and ``torch.sum(action)`` on the host every step to drive a Python list;
    here the history is a device ring of ``growth_threshold + 1`` totals and the "no action this step" predicate is the
    flag the step kernel left in the status block (env.py:191; ``CARLE(exact_action_predicates=True)`` makes it the
    reference's own ``torch.sum`` truthiness): one reduction pass (shared with a SpeedDetector below) + one
    single-block kernel (carle_puffer_update_f32), no host synchronisation.  ``cells`` / ``live_cells`` read the
    device state back lazily."""

    def __init__(self, env, **kwargs):
        super().__init__(env, **kwargs)
        self.my_name = "PufferDetector"
        self.reward_scale = 1.0
        # growth threshold is also the length of the history (mcl.py:822-823)
        self.growth_threshold = 512
        self.growing_steps = 0
        self.__dict__["_ring"] = None           # (ring (cap,), ring_state int32[2] = [head, count], live_total (1,))
        self.__dict__["_host_cells"] = []       # history assigned by the caller before the ring exists

    def _ring_state(self, dev):
        r = self.__dict__["_ring"]
        cap = int(self.growth_threshold) + 1
        if r is None or r[0].numel() != cap or r[0].device != dev:
            carry = self.cells if r is not None else self.__dict__["_host_cells"]
            carry = list(carry)[-(cap - 1):] if cap > 1 else []
            ring = torch.zeros(max(cap, 2), dtype=torch.float32, device=dev)
            state = torch.zeros(2, dtype=torch.int32, device=dev)
            if carry:
                ring[:len(carry)] = torch.tensor(carry, dtype=torch.float32)
                state[1] = len(carry)
            r = self.__dict__["_ring"] = (ring, state, torch.zeros(1, dtype=torch.float32, device=dev))
        return r

    def _graph_state(self):
        r = self.__dict__["_ring"]
        return list(r) if r is not None else []

    @property
    def cells(self):
        """The history of live-cell totals (oldest first), read back from the device ring (synchronises)."""
        r = self.__dict__.get("_ring")
        if r is None:
            return list(self.__dict__.get("_host_cells", []))
        ring, (head, count) = r[0].cpu().tolist(), r[1].cpu().tolist()
        return [ring[(head + i) % len(ring)] for i in range(count)]

    @cells.setter
    def cells(self, value):
        self.__dict__["_host_cells"] = list(value)
        self.__dict__["_ring"] = None

    @property
    def live_cells(self):
        r = self.__dict__.get("_ring")
        return 0.0 if r is None else float(r[2].item())

    @live_cells.setter
    def live_cells(self, value):
        pass                                    # recomputed every step (mcl.py:832)

    def _any_flag_ptr(self, action, dev):
        """Device address of "this step's action had a non-zero element" (mcl.py:835), None for no action."""
        if action is None:
            return None
        inner = self.inner_env
        st = inner.__dict__.get("_status")
        if st is None:
            # no status block to ask (foreign inner environment): the reference's predicate as a device flag
            flag = (torch.sum(_as_device_f32(action, dev)) != 0).to(torch.int32).reshape(1)
            self.__dict__["_flag_keep"] = flag
            return flag.data_ptr()
        # ANY_NONZERO of the slot the step just used (the flags stay until the next step recycles them)
        return st.data_ptr() + 4 * (inner._slot ^ 1)

    def _sharded_step(self, action, grp):
        """Batch sharding (carle_amd.dist): the live-cell total and the action's truthiness are the WHOLE batch's
        (mcl.py:832,835).  Directly on a SpeedDetector that exchanges too, both wrappers share the one all-reduce."""
        sp, inner = self.env, self.inner_env
        dev = inner.my_device
        both = type(sp) is SpeedDetector and _scalar_group(sp) is not None
        obs, reward, done, info = (sp.env if both else sp).step(action)
        if not reward.is_contiguous():
            raise ValueError("PufferDetector needs a contiguous reward tensor")
        ring, state, total = self._ring_state(dev)
        exch = _exchange_buffer(self, dev)
        flag_ptr = self._any_flag_ptr(action, dev)
        if both:
            o = _as_device_f32(obs, dev)
            sums = hip.instance_sums_f32(o, _cached_device(sp, "mass_weight_h", dev), _cached_device(sp, "mass_weight_w", dev))
            same = inner.__dict__.get("universe") is obs
            if same:
                inner.__dict__["_live_cache"] = (weakref.ref(obs), obs._version, sums)
            live4 = None if same else _live_sums(inner, dev)
            com, vel, speed, live, has_prev = sp._dev_buffers(sums.shape[0], dev)
            hip.wrapper_scalars_local(sums, live4, com, has_prev, vel, live, flag_ptr, exch, 3)
            _all_reduce_sum(exch, grp)
            hip.wrapper_scalars_apply(exch, 3, speed, ring, state, total, reward.reshape(-1))
            sp._after_update(com, vel, speed, live)
        else:
            sums = _live_sums(inner, dev)
            hip.wrapper_scalars_local(sums, None, None, None, None, None, flag_ptr, exch, 2)
            _all_reduce_sum(exch, grp)
            hip.wrapper_scalars_apply(exch, 2, None, ring, state, total, reward.reshape(-1))
        return obs, reward, done, info

    def step(self, action):
        grp = _scalar_group(self)
        if grp is not None:
            return self._sharded_step(action, grp)
        sp = self.env
        inner = self.inner_env
        dev = inner.my_device
        # directly on a SpeedDetector (the top of evaluation/eval.py:83-87) over fp32 state: the reduction and both
        # wrappers' updates in ONE launch (carle_speed_puffer_f32) instead of three
        pair = type(sp) is SpeedDetector and type(self) is PufferDetector and getattr(inner, "storage", "f32") == "f32"
        if pair:
            obs, reward, done, info = sp.env.step(action)
            pair = obs is inner.__dict__.get("universe") and obs.dtype == torch.float32 and obs.is_contiguous() \
                and reward.is_contiguous()
            if not pair:
                sp._apply(obs, reward)                      # (the stack below returned something else: Speed's own update)
        else:
            obs, reward, done, info = sp.step(action)
        sums = None
        if pair:
            sums = self.__dict__.get("_pair_sums")
            n = obs.shape[0]
            if sums is None or sums.shape[0] != n or sums.device != dev:
                sums = self.__dict__["_pair_sums"] = torch.zeros(n, 4, dtype=torch.float32, device=dev)
                self.__dict__["_pair_counter"] = torch.zeros(1, dtype=torch.int32, device=dev)
        else:
            sums = _live_sums(inner, dev)
        ring, state, total = self._ring_state(dev)
        flag_ptr = self._any_flag_ptr(action, dev)
        if pair:
            com, vel, speed, live, has_prev = sp._dev_buffers(obs.shape[0], dev)
            hip.speed_puffer_f32(obs, _cached_device(sp, "mass_weight_h", dev), _cached_device(sp, "mass_weight_w", dev), sums,
                                 self.__dict__["_pair_counter"], com, has_prev, vel, speed, live, flag_ptr, ring, state, total,
                                 reward.reshape(-1))
            inner.__dict__["_live_cache"] = (weakref.ref(obs), obs._version, sums)
            sp._after_update(com, vel, speed, live)
            return obs, reward, done, info
        if not reward.is_contiguous():
            raise ValueError("PufferDetector needs a contiguous reward tensor")
        hip.puffer_update_f32(sums, flag_ptr, ring, state, total, reward.reshape(-1))
        return obs, reward, done, info


def get_glider():
    """The glider used throughout the reference's demos (mcl.py:872-879)."""
    action = torch.zeros(1, 1, 64, 64)
    for r, c in ((32, 32), (33, 32), (33, 33), (34, 31), (34, 33)):
        action[0, 0, r, c] = 1.0
    return action


def get_morley_puffer():
    """Common puffer of the Move/Morley rule B368/S245 (mcl.py:881-893)."""
    action = torch.zeros(1, 1, 64, 64)
    cells = [(31, 32), (32, 32), (30, 33), (33, 33)] + [(r, 34) for r in range(29, 35)] + \
            [(30, 35), (30, 36), (33, 35), (33, 36), (31, 37), (32, 37)]
    for r, c in cells:
        action[0, 0, r, c] = 1.0
    return action

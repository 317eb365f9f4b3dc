"""Thin tensor-level wrappers over the C ABI: pointer / stream plumbing only, no arithmetic.

PyTorch is used for device memory and streams; every function here enqueues hand-written HIP
kernels on torch's current stream of the tensors' device and returns without synchronising.
"""
import ctypes

import torch

from . import _lib

STATUS_WORDS = 32
ST_LAST_RESET, ST_STEP_NUMBER, ST_STEPS_SINCE_ACTION, ST_RESETS = 4, 6, 8, 10
ST_MOMENTS_OUT, ST_MOMENTS_ACC = 12, 14      # two 64-bit device pointers (see carle_hip.h)
ST_TRACE_CAP, ST_TRACE_IDX = 24, 25          # ring of per-step reward moments behind MOMENTS_OUT (CARLE_ST_TRACE_*)


_gpu_ok = False


def require_gpu(device=None):
    global _gpu_ok
    if _gpu_ok:
        return
    if not torch.cuda.is_available():
        raise _lib.CarleHipError(
            "carle_amd needs a HIP device (MI355X): torch.cuda.is_available() is False and "
            "there is no CPU fallback")
    _lib.load()
    _gpu_ok = True


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream(t):
    """Raw hipStream_t of torch's CURRENT stream on the tensor's device (the stream the caller's
    other work is ordered on)."""
    if _raw_stream is not None:
        return _raw_stream(t.device.index)
    return torch.cuda.current_stream(t.device).cuda_stream


class _on_device:
    """Make the tensor's device current for the duration of a launch; free when it already is."""
    __slots__ = ("idx", "ctx")

    def __init__(self, device):
        self.idx = device.index
        self.ctx = None

    def __enter__(self):
        cur = _get_device() if _get_device is not None else torch.cuda.current_device()
        if cur != self.idx:
            self.ctx = torch.cuda.device(self.idx)
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


def _chk_grid(t, name, dtype):
    if not (t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise ValueError(f"{name} must be a contiguous {dtype} tensor on a HIP device, got "
                         f"{t.dtype} on {t.device}, contiguous={t.is_contiguous()}")


def rule_mask(birth, survive):
    """Lists of neighbour counts -> 18-bit mask (bit k survive, bit 9+k birth); the `elem ==
    count` membership of carle/env.py:221-224 (values outside 0..8 can never match)."""
    m = 0
    for k in survive:
        if k == int(k) and 0 <= int(k) <= 8:
            m |= 1 << int(k)
    for k in birth:
        if k == int(k) and 0 <= int(k) <= 8:
            m |= 1 << (9 + int(k))
    return m


def new_status(device):
    """Zeroed per-environment status block (layout: include/carle_hip.h CARLE_ST_*)."""
    return torch.zeros(STATUS_WORDS, dtype=torch.int32, device=device)


def env_settle(status):
    """Applies a pending one-launch (CARLE_STEP_DEFERRED) step to the counters of the status block; call before
    reading them.  One tiny kernel on the current stream, no host synchronisation."""
    lib = _lib.load()
    with _on_device(status.device):
        rc = lib.carle_env_settle(status.data_ptr(), _stream(status))
    _lib.check(rc, "carle_env_settle")
    return status


def reload_knobs():
    """Re-reads the CARLE_* experiment knobs from the environment (the library reads them once, at first use)."""
    _lib.load().carle_reload_knobs()


def ca_step_f32(u_in, action, u_out, rule, status=None, slot=0):
    """u_out <- one generation of (u_in XOR centred action); all (N,1,H,W)/(N,1,AH,AW) fp32."""
    lib = _lib.load()
    _chk_grid(u_in, "u_in", torch.float32)
    _chk_grid(u_out, "u_out", torch.float32)
    n, _, h, w = u_in.shape
    if u_out.shape != u_in.shape:
        raise ValueError("u_out shape differs from u_in")
    ap, ah, aw = None, 0, 0
    if action is not None:
        _chk_grid(action, "action", torch.float32)
        if action.shape[0] != n:
            raise ValueError("action batch differs from universe batch")
        ah, aw = action.shape[2], action.shape[3]
        ap = action.data_ptr()
    sp = status.data_ptr() if status is not None else None
    with _on_device(u_in.device):
        rc = lib.carle_ca_step_f32(u_in.data_ptr(), ap, u_out.data_ptr(), n, h, w, ah, aw, rule,
                                   sp, slot, _stream(u_in))
    _lib.check(rc, "carle_ca_step_f32")


def ca_finish(u_out, reward, status, slot, has_action):
    lib = _lib.load()
    n = u_out.shape[0]
    rp = reward.data_ptr() if reward is not None else None
    with _on_device(u_out.device):
        rc = lib.carle_ca_finish(u_out.data_ptr(), u_out.numel() * u_out.element_size(), rp, n,
                                 status.data_ptr(), slot, 1 if has_action else 0, _stream(u_out))
    _lib.check(rc, "carle_ca_finish")


def ca_toggle_f32(u_in, action, u_out):
    """u_out <- (u_in != 0) XOR centred (action != 0); stand-alone CARLE.apply_action."""
    lib = _lib.load()
    _chk_grid(u_in, "u_in", torch.float32)
    _chk_grid(u_out, "u_out", torch.float32)
    _chk_grid(action, "action", torch.float32)
    n, _, h, w = u_in.shape
    with _on_device(u_in.device):
        rc = lib.carle_ca_toggle_f32(u_in.data_ptr(), action.data_ptr(), u_out.data_ptr(), n, h, w,
                                     action.shape[2], action.shape[3], _stream(u_in))
    _lib.check(rc, "carle_ca_toggle_f32")


def _ptr(t):
    return t.data_ptr() if t is not None else None


def rnd2d_bonus_f32(obs, pred_params, rand_params, workspace=None, want_outputs=False, binary=False,
                    reward=None, reward_scale=1.0, bonus_out=None, stream=None, tables_valid=False):
    """RND2D bonus (N,) for obs (N,1,H,W) fp32 -- or int64 bit-packed (N,1,64,1).  params = flat
    fp32 blobs (see carle_hip.h).  binary=True: obs values are exactly 0/1 (fused 64x64 path).
    reward (N,1): additionally reward += reward_scale * bonus inside the kernel.  stream: raw
    hipStream_t to launch on (default: torch's current stream)."""
    lib = _lib.load()
    packed = obs.dtype == torch.int64
    _chk_grid(obs, "obs", torch.int64 if packed else torch.float32)
    n, _, h, w = obs.shape
    if packed:
        w *= 64
    need = lib.carle_rnd2d_workspace_floats(n, h, w)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(max(need, 1), dtype=torch.float32, device=obs.device)
        tables_valid = False
    bonus = bonus_out if bonus_out is not None else torch.empty(n, dtype=torch.float32, device=obs.device)
    pred = torch.empty(n, 16, dtype=torch.float32, device=obs.device) if want_outputs else None
    targ = torch.empty(n, 16, dtype=torch.float32, device=obs.device) if want_outputs else None
    flags = (1 if binary else 0) | (2 if tables_valid else 0)
    with _on_device(obs.device):
        fn = lib.carle_rnd2d_bonus_b1 if packed else lib.carle_rnd2d_bonus_f32
        rc = fn(obs.data_ptr(), n, h, w, pred_params.data_ptr(), rand_params.data_ptr(),
                workspace.data_ptr(), bonus.data_ptr(), _ptr(pred), _ptr(targ), _ptr(reward),
                float(reward_scale), flags, stream if stream is not None else _stream(obs))
    if rc != 0:
        _lib.check(rc, "carle_rnd2d_bonus")
    return (bonus, pred, targ, workspace) if want_outputs else (bonus, workspace)


def ae2d_bonus_f32(obs, params, workspace=None, want_recon=False, binary=False,
                   reward=None, reward_scale=1.0, bonus_out=None, stream=None, tables_valid=False):
    """AE2D bonus (N,) for obs (N,1,H,W) fp32 -- or int64 bit-packed (N,1,64,1); params = flat
    164-float blob.  binary / reward / stream as for rnd2d_bonus_f32."""
    lib = _lib.load()
    packed = obs.dtype == torch.int64
    _chk_grid(obs, "obs", torch.int64 if packed else torch.float32)
    n, _, h, w = obs.shape
    if packed:
        w *= 64
    need = lib.carle_ae2d_workspace_floats(n, h, w)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(max(need, 1), dtype=torch.float32, device=obs.device)
        tables_valid = False
    bonus = bonus_out if bonus_out is not None else torch.empty(n, dtype=torch.float32, device=obs.device)
    recon = torch.empty(n, 1, h, w, dtype=torch.float32, device=obs.device) if want_recon else None
    flags = (1 if binary else 0) | (2 if tables_valid else 0)
    with _on_device(obs.device):
        fn = lib.carle_ae2d_bonus_b1 if packed else lib.carle_ae2d_bonus_f32
        rc = fn(obs.data_ptr(), n, h, w, params.data_ptr(), workspace.data_ptr(), bonus.data_ptr(),
                _ptr(recon), _ptr(reward), float(reward_scale), flags,
                stream if stream is not None else _stream(obs))
    if rc != 0:
        _lib.check(rc, "carle_ae2d_bonus")
    return (bonus, recon, workspace) if want_recon else (bonus, workspace)


def ae2d_bonus_b1_stats(obs, params, workspace, bonus, reward, reward_scale, tables_valid, partials, status, stream=None):
    """AE2D bonus of bit-packed 64x64 observations with ``reward += scale * bonus`` AND the reward moments of the step left as
    per-workgroup partial sums registered in the environment's status block (carle_ae2d_bonus_b1_stats)."""
    lib = _lib.load()
    n = obs.shape[0]
    with _on_device(obs.device):
        rc = lib.carle_ae2d_bonus_b1_stats(obs.data_ptr(), n, 64, 64, params.data_ptr(), workspace.data_ptr(), bonus.data_ptr(),
                                           reward.data_ptr(), float(reward_scale), 2 if tables_valid else 0, partials.data_ptr(),
                                           status.data_ptr(), stream if stream is not None else _stream(obs))
    if rc != 0:
        _lib.check(rc, "carle_ae2d_bonus_b1_stats")


def ae2d_predict(obs, target, params, workspace=None, scratch=None, grad=None, grad_scale=0.0, want_recon=False,
                 bonus_out=None, stream=None, tables_valid=False):
    """PredictionBonus.get_bonus (carle/mcl.py:581-605): bonus (N,) = mean_hw (target - decoder(obs))^2 for two frames
    of the same universes -- both fp32 (N,1,H,W), H == W a multiple of 8 (per-layer kernels), or both int64 bit-packed
    (N,1,H,H/64), H a multiple of 64 (AE2D's fused 64x64 / tiled kernels with a target).  With ``grad`` (flat 164 floats)
    the predictor's gradient of grad_scale/2 * sum (target - decoder(obs))^2 is accumulated into it on the way.
    Returns (bonus, recon or None, workspace, scratch)."""
    lib = _lib.load()
    packed = obs.dtype == torch.int64
    _chk_grid(obs, "obs", torch.int64 if packed else torch.float32)
    _chk_grid(target, "target", torch.int64 if packed else torch.float32)
    if target.shape != obs.shape or target.device != obs.device:
        raise ValueError(f"ae2d_predict: obs {tuple(obs.shape)} and target {tuple(target.shape)} must match")
    n, _, h, w = obs.shape
    if packed:
        w *= 64
    need = lib.carle_ae2d_workspace_floats(n, h, w)
    if workspace is None or workspace.numel() < need or workspace.device != obs.device:
        workspace = torch.empty(max(need, 1), dtype=torch.float32, device=obs.device)
        tables_valid = False
    bonus = bonus_out if bonus_out is not None else torch.empty(n, dtype=torch.float32, device=obs.device)
    recon = None
    st = stream if stream is not None else _stream(obs)
    flags = 2 if tables_valid else 0
    with _on_device(obs.device):
        if grad is None:
            recon = torch.empty(n, 1, h, w, dtype=torch.float32, device=obs.device) if want_recon else None
            if packed:
                rc = lib.carle_ae2d_predict_bonus_b1(obs.data_ptr(), target.data_ptr(), n, h, w, params.data_ptr(),
                                                     workspace.data_ptr(), bonus.data_ptr(), _ptr(recon), flags, st)
            else:
                rc = lib.carle_ae2d_predict_bonus_f32(obs.data_ptr(), target.data_ptr(), n, h, w, params.data_ptr(),
                                                      workspace.data_ptr(), bonus.data_ptr(), _ptr(recon), st)
        else:
            if packed and h == 64:
                sneed = lib.carle_ae2d_train_partial_floats(n)
            else:
                sneed = lib.carle_ae2d_train_scratch_floats(n, h, w)
            if scratch is None or scratch.numel() < sneed or scratch.device != obs.device:
                scratch = torch.empty(max(sneed, 1), dtype=torch.float32, device=obs.device)
            if packed:
                rc = lib.carle_ae2d_predict_train_b1(obs.data_ptr(), target.data_ptr(), n, h, w, params.data_ptr(),
                                                     workspace.data_ptr(), scratch.data_ptr(), grad.data_ptr(),
                                                     bonus.data_ptr(), float(grad_scale), flags, st)
            else:
                rc = lib.carle_ae2d_predict_train_f32(obs.data_ptr(), target.data_ptr(), n, h, w, params.data_ptr(),
                                                      workspace.data_ptr(), scratch.data_ptr(), grad.data_ptr(),
                                                      bonus.data_ptr(), float(grad_scale), st)
    if rc != 0:
        _lib.check(rc, "carle_ae2d_predict")
    return bonus, recon, workspace, scratch


def rnd_ae_bonus(obs, rnd_pred, rnd_rand, rnd_ws, rnd_bonus, rnd_scale, rnd_valid, ae_params, ae_ws, ae_bonus, ae_scale,
                 ae_valid, reward, stream=None, partials=None, status=None):
    """Both bonuses of a stacked RND2D + AE2D pair in one launch (carle_rnd_ae_bonus): binary obs, fp32 (N,1,H,W) or
    int64 bit-packed, H == W a multiple of 64 above 64.  Workspaces must already have their full size."""
    lib = _lib.load()
    packed = obs.dtype == torch.int64
    n, _, h, w = obs.shape
    if packed:
        w *= 64
    with _on_device(obs.device):
        rc = lib.carle_rnd_ae_bonus(obs.data_ptr(), FMT_B1 if packed else FMT_F32, n, h, w, rnd_pred.data_ptr(),
                                    rnd_rand.data_ptr(), rnd_ws.data_ptr(), rnd_bonus.data_ptr(), float(rnd_scale),
                                    2 if rnd_valid else 0, ae_params.data_ptr(), ae_ws.data_ptr(), ae_bonus.data_ptr(),
                                    float(ae_scale), 2 if ae_valid else 0, _ptr(reward), _ptr(partials), _ptr(status),
                                    stream if stream is not None else _stream(obs))
    if rc != 0:
        _lib.check(rc, "carle_rnd_ae_bonus")


def rnd_ae_train(obs, rnd_pred, rnd_rand, rnd_ws, rnd_scratch, rnd_grad, rnd_bonus, rnd_scale, rnd_grad_scale, rnd_valid,
                 ae_params, ae_ws, ae_scratch, ae_grad, ae_bonus, ae_scale, ae_grad_scale, ae_valid, reward, stream=None,
                 partials=None, status=None):
    """Bonuses, reward updates and gradient accumulation of a stacked RND2D + AE2D pair in training mode, four launches
    (carle_rnd_ae_train): binary obs, fp32 (N,1,H,W) or int64 bit-packed, H == W a multiple of 64 above 64.  Workspaces,
    scratch and gradient buffers must already have their full size."""
    lib = _lib.load()
    packed = obs.dtype == torch.int64
    n, _, h, w = obs.shape
    if packed:
        w *= 64
    with _on_device(obs.device):
        rc = lib.carle_rnd_ae_train(obs.data_ptr(), FMT_B1 if packed else FMT_F32, n, h, w, rnd_pred.data_ptr(),
                                    rnd_rand.data_ptr(), rnd_ws.data_ptr(), rnd_scratch.data_ptr(), rnd_grad.data_ptr(),
                                    rnd_bonus.data_ptr(), float(rnd_scale), float(rnd_grad_scale), 2 if rnd_valid else 0,
                                    ae_params.data_ptr(), ae_ws.data_ptr(), ae_scratch.data_ptr(), ae_grad.data_ptr(),
                                    ae_bonus.data_ptr(), float(ae_scale), float(ae_grad_scale), 2 if ae_valid else 0,
                                    _ptr(reward), _ptr(partials), _ptr(status),
                                    stream if stream is not None else _stream(obs))
    if rc != 0:
        _lib.check(rc, "carle_rnd_ae_train")


def stream_fork(event_handle, from_stream, to_stream):
    """to_stream waits for everything enqueued so far on from_stream (raw handles)."""
    rc = _lib.load().carle_stream_fork(event_handle, from_stream, to_stream)
    if rc != 0:
        _lib.check(rc, "carle_stream_fork")


def batch_moments_f32(x, out=None, acc=None, stream=None):
    """(count, mean, M2) of a fp32 device vector as a fp64[3] device tensor (no host sync);
    ``acc`` (fp64[3] device) is additionally Chan-merged with the batch inside the kernel."""
    lib = _lib.load()
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()):
        raise ValueError("batch_moments_f32 needs a contiguous fp32 tensor on a HIP device")
    if out is None:
        out = torch.empty(3, dtype=torch.float64, device=x.device)
    with _on_device(x.device):
        rc = lib.carle_batch_moments_f32(x.data_ptr(), x.numel(), out.data_ptr(), _ptr(acc),
                                         stream if stream is not None else _stream(x))
    _lib.check(rc, "carle_batch_moments_f32")
    return out


def normalize_f32(x, moments, eps=1e-8):
    """(x - mean) / sqrt(M2/count + eps) with fp64[3] device moments."""
    lib = _lib.load()
    y = torch.empty_like(x)
    with _on_device(x.device):
        rc = lib.carle_normalize_f32(x.data_ptr(), y.data_ptr(), x.numel(), moments.data_ptr(),
                                     float(eps), _stream(x))
    _lib.check(rc, "carle_normalize_f32")
    return y


def _step_flags(publish_flags, merge_moments, deferred=False):
    """`publish_flags` argument of the carle_env_step_* entry points (CARLE_STEP_* bits)."""
    return (1 if publish_flags else 0) | (2 if merge_moments else 0) | (4 if deferred else 0)


_bits_written = ctypes.c_int(0)


def env_step_f32(u_in, action, u_out, reward, rule, status, slot, publish_flags=True, merge_moments=False,
                 deferred=True, bits_out=None):
    """CARLE.step hot path, one library call (shapes are validated by the caller; pointers by the library).
    deferred (default): ONE kernel launch where the library has a one-launch step (64x64 grids: master reset and
    reward inside the step kernel, counters settled lazily -- read them through ``env_settle``); else step kernel +
    epilogue kernel.  bits_out (int64 (N,1,H,W/64), optional): the launch also leaves a bit-packed copy of the new
    state there where it can (carle_env_step_f32_bits); returns True iff it did."""
    lib = _lib.load()
    n, _, h, w = u_in.shape
    if action is None:
        ap, ah, aw = None, 0, 0
    else:
        ap, ah, aw = action.data_ptr(), action.shape[2], action.shape[3]
    with _on_device(u_in.device):
        if bits_out is None:
            rc = lib.carle_env_step_f32(u_in.data_ptr(), ap, u_out.data_ptr(), reward.data_ptr(), n, h,
                                        w, ah, aw, rule, status.data_ptr(), slot,
                                        _step_flags(publish_flags, merge_moments, deferred), _stream(u_in))
        else:
            rc = lib.carle_env_step_f32_bits(u_in.data_ptr(), ap, u_out.data_ptr(), reward.data_ptr(), n, h,
                                             w, ah, aw, rule, status.data_ptr(), slot,
                                             _step_flags(publish_flags, merge_moments, deferred), bits_out.data_ptr(),
                                             ctypes.byref(_bits_written), _stream(u_in))
    if rc != 0:
        _lib.check(rc, "carle_env_step_f32")
    return bits_out is not None and _bits_written.value != 0


def env_step_f32_packed(bits_in, action, u_out, reward, rule, status, slot, publish_flags=True, merge_moments=False,
                        bits_out=None):
    """The one-launch step of 64x64 fp32 grids reading its input state from the bit-packed copy ``bits_in`` (int64 (N,1,64,1))
    the previous step left, instead of the fp32 cells (carle_env_step_f32_packed); writes ``u_out`` and ``bits_out``."""
    lib = _lib.load()
    n, _, h, w = u_out.shape
    if action is None:
        ap, ah, aw = None, 0, 0
    else:
        ap, ah, aw = action.data_ptr(), action.shape[2], action.shape[3]
    with _on_device(u_out.device):
        rc = lib.carle_env_step_f32_packed(bits_in.data_ptr(), ap, u_out.data_ptr(), reward.data_ptr(), n, h, w, ah, aw, rule,
                                           status.data_ptr(), slot, _step_flags(publish_flags, merge_moments, True),
                                           bits_out.data_ptr(), _stream(u_out))
    if rc != 0:
        _lib.check(rc, "carle_env_step_f32_packed")


def env_run_f32(u_in, buf_a, buf_b, actions, reward, rule, status, slot0, k, ah=0, aw=0, p=-1.0, seed=0, offset=0,
                merge_moments=False, deferred=True):
    """k CARLE.step calls in one library call (carle_env_run_f32): the state is READ from u_in and the steps
    ping-pong between the scratch buffers buf_a / buf_b (final state in buf_a for odd k, buf_b for even k; buf_b may
    be None for k == 1).  actions: (k,N,1,AH,AW) fp32 or None (then p >= 0 draws Bernoulli(p) toggles in the kernel,
    p < 0 means no action)."""
    lib = _lib.load()
    n, _, h, w = u_in.shape
    ap = None
    if actions is not None:
        _chk_grid(actions, "actions", torch.float32)
        if actions.dim() != 5 or actions.shape[0] != k or actions.shape[1] != n:
            raise ValueError(f"actions must have shape (k={k}, N={n}, 1, AH, AW), got {tuple(actions.shape)}")
        ah, aw, ap = actions.shape[3], actions.shape[4], actions.data_ptr()
    with _on_device(u_in.device):
        rc = lib.carle_env_run_f32(u_in.data_ptr(), _ptr(buf_a), _ptr(buf_b), ap, reward.data_ptr(), n, h, w, ah, aw,
                                   rule, float(p), int(seed), int(offset), int(k), status.data_ptr(), slot0,
                                   _step_flags(False, merge_moments, deferred), _stream(u_in))
    if rc != 0:
        _lib.check(rc, "carle_env_run_f32")


def merge_moments(gathered, world, state, last, pending):
    """Device-side Chan merge of all-gathered (count, mean, M2) rows into last / running state."""
    lib = _lib.load()
    with _on_device(state.device):
        rc = lib.carle_merge_moments(gathered.data_ptr(), world, state.data_ptr(), last.data_ptr(),
                                     _ptr(pending), _stream(state))
    _lib.check(rc, "carle_merge_moments")


# ---------------------------------------------------------------------------------------------
# bit-packed format (CARLE_FMT_B1): tensors are int64 (N,1,H,W/64), bit j of word i = col 64i+j
# ---------------------------------------------------------------------------------------------

def pack_f32_b1(src, status=None, slot=0, out=None):
    """fp32 (N,1,H,W) -> int64 (N,1,H,W/64); optionally publishes the fp32 action predicates."""
    lib = _lib.load()
    _chk_grid(src, "src", torch.float32)
    n, _, h, w = src.shape
    if out is None:
        out = torch.empty(n, 1, h, w // 64, dtype=torch.int64, device=src.device)
    with _on_device(src.device):
        rc = lib.carle_pack_f32_b1(src.data_ptr(), out.data_ptr(), n, h, w, _ptr(status), slot,
                                   _stream(src))
    _lib.check(rc, "carle_pack_f32_b1")
    return out


def unpack_b1_f32(src, out=None):
    """int64 (N,1,H,W/64) -> fp32 (N,1,H,W) of 0.0 / 1.0."""
    lib = _lib.load()
    _chk_grid(src, "src", torch.int64)
    n, _, h, ww = src.shape
    if out is None:
        out = torch.empty(n, 1, h, ww * 64, dtype=torch.float32, device=src.device)
    with _on_device(src.device):
        rc = lib.carle_unpack_b1_f32(src.data_ptr(), out.data_ptr(), n, h, ww * 64, _stream(src))
    _lib.check(rc, "carle_unpack_b1_f32")
    return out


def ca_step_b1(u_in, action, u_out, rule, status=None, slot=0):
    """One generation on packed grids; action = packed (N,1,AH,AW/64) int64 or None."""
    lib = _lib.load()
    _chk_grid(u_in, "u_in", torch.int64)
    _chk_grid(u_out, "u_out", torch.int64)
    n, _, h, ww = u_in.shape
    ap, ah, aw = None, 0, 0
    if action is not None:
        _chk_grid(action, "action", torch.int64)
        ah, aw = action.shape[2], action.shape[3] * 64
        ap = action.data_ptr()
    with _on_device(u_in.device):
        rc = lib.carle_ca_step_b1(u_in.data_ptr(), ap, u_out.data_ptr(), n, h, ww * 64, ah, aw, rule,
                                  _ptr(status), slot, _stream(u_in))
    _lib.check(rc, "carle_ca_step_b1")


def ca_multistep_b1(u_in, u_out, rule, k, status=None):
    """k action-free generations on packed grids in one launch (u_out may be u_in)."""
    lib = _lib.load()
    _chk_grid(u_in, "u_in", torch.int64)
    _chk_grid(u_out, "u_out", torch.int64)
    n, _, h, ww = u_in.shape
    if u_out.shape != u_in.shape:
        raise ValueError("u_out shape differs from u_in")
    with _on_device(u_in.device):
        rc = lib.carle_ca_multistep_b1(u_in.data_ptr(), u_out.data_ptr(), n, h, ww * 64, rule, int(k),
                                       _ptr(status), _stream(u_in))
    _lib.check(rc, "carle_ca_multistep_b1")


def env_step_b1(u_in, action, u_out, reward, rule, status, slot, publish_flags=True, merge_moments=False, deferred=True):
    """CARLE.step hot path on packed state, one library call (``deferred`` as for env_step_f32: one launch for no
    action / the 64-column window)."""
    lib = _lib.load()
    n, _, h, ww = u_in.shape
    if action is None:
        ap, ah, aw = None, 0, 0
    else:
        ap, ah, aw = action.data_ptr(), action.shape[2], action.shape[3] * 64
    with _on_device(u_in.device):
        rc = lib.carle_env_step_b1(u_in.data_ptr(), ap, u_out.data_ptr(), reward.data_ptr(), n, h,
                                   ww * 64, ah, aw, rule, status.data_ptr(), slot,
                                   _step_flags(publish_flags, merge_moments, deferred), _stream(u_in))
    if rc != 0:
        _lib.check(rc, "carle_env_step_b1")


# ---------------------------------------------------------------------------------------------
# uint8 format (CARLE_FMT_U8): tensors are uint8 (N,1,H,W), non-zero = alive; the action may be
# fp32 (what the reference's agents emit) or uint8
# ---------------------------------------------------------------------------------------------
FMT_F32, FMT_U8, FMT_B1 = 0, 1, 2


def _action_fmt(action):
    if action.dtype == torch.float32:
        return FMT_F32
    if action.dtype == torch.uint8:
        return FMT_U8
    raise ValueError(f"action must be float32 or uint8 for uint8 grids, got {action.dtype}")


def ca_step_u8(u_in, action, u_out, rule, status=None, slot=0):
    """One generation on uint8 grids; action fp32 / uint8 (N,1,AH,AW) or None."""
    lib = _lib.load()
    _chk_grid(u_in, "u_in", torch.uint8)
    _chk_grid(u_out, "u_out", torch.uint8)
    n, _, h, w = u_in.shape
    if u_out.shape != u_in.shape:
        raise ValueError("u_out shape differs from u_in")
    ap, fmt, ah, aw = None, FMT_F32, 0, 0
    if action is not None:
        fmt = _action_fmt(action)
        _chk_grid(action, "action", action.dtype)
        if action.shape[0] != n:
            raise ValueError("action batch differs from universe batch")
        ah, aw = action.shape[2], action.shape[3]
        ap = action.data_ptr()
    with _on_device(u_in.device):
        rc = lib.carle_ca_step_u8(u_in.data_ptr(), ap, fmt, u_out.data_ptr(), n, h, w, ah, aw, rule,
                                  _ptr(status), slot, _stream(u_in))
    _lib.check(rc, "carle_ca_step_u8")


def env_step_u8(u_in, action, u_out, reward, rule, status, slot, publish_flags=True, merge_moments=False,
                deferred=True):
    """CARLE.step hot path on uint8 state, one library call (``deferred`` as for env_step_f32)."""
    lib = _lib.load()
    n, _, h, w = u_in.shape
    if action is None:
        ap, fmt, ah, aw = None, FMT_F32, 0, 0
    else:
        ap, fmt, ah, aw = action.data_ptr(), _action_fmt(action), action.shape[2], action.shape[3]
    with _on_device(u_in.device):
        rc = lib.carle_env_step_u8(u_in.data_ptr(), ap, fmt, u_out.data_ptr(), reward.data_ptr(), n, h,
                                   w, ah, aw, rule, status.data_ptr(), slot,
                                   _step_flags(publish_flags, merge_moments, deferred), _stream(u_in))
    if rc != 0:
        _lib.check(rc, "carle_env_step_u8")


def ca_toggle_u8(u_in, action, u_out):
    """u_out <- (u_in != 0) XOR centred (action != 0) on uint8 grids."""
    lib = _lib.load()
    _chk_grid(u_in, "u_in", torch.uint8)
    _chk_grid(u_out, "u_out", torch.uint8)
    _chk_grid(action, "action", action.dtype)
    n, _, h, w = u_in.shape
    with _on_device(u_in.device):
        rc = lib.carle_ca_toggle_u8(u_in.data_ptr(), action.data_ptr(), _action_fmt(action),
                                    u_out.data_ptr(), n, h, w, action.shape[2], action.shape[3],
                                    _stream(u_in))
    _lib.check(rc, "carle_ca_toggle_u8")


def pack_u8_b1(src, status=None, slot=0, out=None):
    """uint8 (N,1,H,W) -> int64 (N,1,H,W/64); optionally publishes the action predicates."""
    lib = _lib.load()
    _chk_grid(src, "src", torch.uint8)
    n, _, h, w = src.shape
    if out is None:
        out = torch.empty(n, 1, h, w // 64, dtype=torch.int64, device=src.device)
    with _on_device(src.device):
        rc = lib.carle_pack_u8_b1(src.data_ptr(), out.data_ptr(), n, h, w, _ptr(status), slot,
                                  _stream(src))
    _lib.check(rc, "carle_pack_u8_b1")
    return out


def unpack_b1_u8(src, out=None):
    """int64 (N,1,H,W/64) -> uint8 (N,1,H,W) of 0 / 1."""
    lib = _lib.load()
    _chk_grid(src, "src", torch.int64)
    n, _, h, ww = src.shape
    if out is None:
        out = torch.empty(n, 1, h, ww * 64, dtype=torch.uint8, device=src.device)
    with _on_device(src.device):
        rc = lib.carle_unpack_b1_u8(src.data_ptr(), out.data_ptr(), n, h, ww * 64, _stream(src))
    _lib.check(rc, "carle_unpack_b1_u8")
    return out


def action_flags_f32(action, status, slot):
    """Sets ANY_NONZERO / NOT_ALL_ONES of `slot` from every element of an fp32 tensor."""
    lib = _lib.load()
    _chk_grid(action, "action", torch.float32)
    with _on_device(action.device):
        rc = lib.carle_action_flags_f32(action.data_ptr(), action.numel(), status.data_ptr(), slot,
                                        _stream(action))
    _lib.check(rc, "carle_action_flags_f32")


# ---------------------------------------------------------------------------------------------
# device-side agents
# ---------------------------------------------------------------------------------------------

def bernoulli_f32(shape, p, seed, offset, device, out=None):
    """Bernoulli(p) 0.0/1.0 tensor drawn on the device (Philox4x32-10, see carle_hip.h)."""
    lib = _lib.load()
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=device)
    with _on_device(out.device):
        rc = lib.carle_bernoulli_f32(out.data_ptr(), out.numel(), float(p), int(seed), int(offset),
                                     _stream(out))
    _lib.check(rc, "carle_bernoulli_f32")
    return out


def agent_rn_forward(obs, conv1_w, conv2_w, lin_w, toggle_rate, workspace=None, want_output=False):
    """RandomNetworkAgent.forward on the device (carle/agents.py:91-102): returns (action (N, OUT) fp32,
    sigmoid output (N, OUT) or None, workspace)."""
    lib = _lib.load()
    _chk_grid(obs, "obs", torch.float32)
    n, _, h, w = obs.shape
    out_nodes, dense = lin_w.shape
    if dense != (h // 4) * (w // 4) or h % 8 or w % 8:
        raise ValueError(f"RandomNetworkAgent: observation {h}x{w} does not match the linear layer ({dense} inputs)")
    for t in (conv1_w, conv2_w, lin_w):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.device != obs.device:
            raise ValueError("RandomNetworkAgent: weights must be contiguous fp32 tensors on the observation's device")
    need = lib.carle_agent_rn_workspace_floats(n, h, w, out_nodes)
    if workspace is None or workspace.numel() < need or workspace.device != obs.device:
        workspace = torch.empty(max(need, 4), dtype=torch.float32, device=obs.device)
    action = torch.empty((n, out_nodes), dtype=torch.float32, device=obs.device)
    output = torch.empty((n, out_nodes), dtype=torch.float32, device=obs.device) if want_output else None
    with _on_device(obs.device):
        rc = lib.carle_agent_rn_forward_f32(obs.data_ptr(), n, h, w, conv1_w.data_ptr(), conv2_w.data_ptr(),
                                            lin_w.data_ptr(), out_nodes, float(toggle_rate), workspace.data_ptr(),
                                            _ptr(output), action.data_ptr(), _stream(obs))
    _lib.check(rc, "carle_agent_rn_forward_f32")
    return action, output, workspace


def env_step_f32_rand(u_in, u_out, reward, rule, status, slot, ah, aw, p, seed, offset, action_out=None,
                      merge_moments=False):
    """CARLE.step with Bernoulli(p) toggles drawn inside the step kernel (64x64 grids)."""
    lib = _lib.load()
    n, _, h, w = u_in.shape
    with _on_device(u_in.device):
        rc = lib.carle_env_step_f32_rand(u_in.data_ptr(), u_out.data_ptr(), reward.data_ptr(), n, h, w,
                                         ah, aw, rule, float(p), int(seed), int(offset),
                                         _ptr(action_out), status.data_ptr(), slot,
                                         _step_flags(False, merge_moments), _stream(u_in))
    if rc != 0:
        _lib.check(rc, "carle_env_step_f32_rand")


def env_step_u8_rand(u_in, u_out, reward, rule, status, slot, ah, aw, p, seed, offset, merge_moments=False):
    """CARLE.step on uint8 state with Bernoulli(p) toggles drawn inside the step kernel."""
    lib = _lib.load()
    n, _, h, w = u_in.shape
    with _on_device(u_in.device):
        rc = lib.carle_env_step_u8_rand(u_in.data_ptr(), u_out.data_ptr(), reward.data_ptr(), n, h, w, ah, aw, rule,
                                        float(p), int(seed), int(offset), status.data_ptr(), slot,
                                        _step_flags(False, merge_moments), _stream(u_in))
    if rc != 0:
        _lib.check(rc, "carle_env_step_u8_rand")


def env_step_b1_rand(u_in, u_out, reward, rule, status, slot, ah, aw, p, seed, offset, merge_moments=False):
    """CARLE.step on bit-packed state with the 64x64 Bernoulli(p) window drawn inside the step kernel."""
    lib = _lib.load()
    n, _, h, ww = u_in.shape
    with _on_device(u_in.device):
        rc = lib.carle_env_step_b1_rand(u_in.data_ptr(), u_out.data_ptr(), reward.data_ptr(), n, h, ww * 64, ah, aw,
                                        rule, float(p), int(seed), int(offset), status.data_ptr(), slot,
                                        _step_flags(False, merge_moments), _stream(u_in))
    if rc != 0:
        _lib.check(rc, "carle_env_step_b1_rand")


def instance_sums_f32(x, w0=None, w1=None, w2=None):
    """(N,4) fp32: per-instance [sum x, sum x*w0, sum x*w1, sum x*w2] over all trailing dims; the
    weight planes (same element count as one instance) are optional."""
    lib = _lib.load()
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()):
        x = x.to(torch.float32).contiguous()
    n = x.shape[0]
    elems = x.numel() // max(n, 1)
    ws = []
    for wt in (w0, w1, w2):
        if wt is not None:
            wt = wt.to(x.device, torch.float32).contiguous()
            if wt.numel() != elems:
                raise ValueError("weight plane size differs from one instance")
        ws.append(wt)
    out = torch.empty(n, 4, dtype=torch.float32, device=x.device)
    with _on_device(x.device):
        rc = lib.carle_instance_sums_f32(x.data_ptr(), n, elems, _ptr(ws[0]), _ptr(ws[1]), _ptr(ws[2]),
                                         out.data_ptr(), _stream(x))
    _lib.check(rc, "carle_instance_sums_f32")
    return out


def adam_update_f32(params, blob, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, zero_grad=True, stream=None):
    """torch.optim.Adam's step on the flat gradient, one launch (carle_adam_update_f32): ``params`` = the contiguous
    fp32 parameter tensors in blob order (updated in place), ``blob`` the flat copy the bonus kernels read."""
    import ctypes
    lib = _lib.load()
    k = len(params)
    ptrs = (ctypes.c_void_p * k)(*[p.data_ptr() for p in params])
    counts = (ctypes.c_int64 * k)(*[p.numel() for p in params])
    with _on_device(grad.device):
        rc = lib.carle_adam_update_f32(ptrs, counts, k, _ptr(blob), grad.data_ptr(), exp_avg.data_ptr(),
                                       exp_avg_sq.data_ptr(), float(lr), float(beta1), float(beta2), float(eps),
                                       int(step), 1 if zero_grad else 0, stream if stream is not None else _stream(grad))
    if rc != 0:
        _lib.check(rc, "carle_adam_update_f32")


def speed_update_f32(sums4, live4, com, state, velocity, speed, live_out, reward):
    """SpeedDetector tail (carle_speed_update_f32): centre of mass, velocity, batch speed, reward += speed."""
    lib = _lib.load()
    n = sums4.shape[0]
    with _on_device(sums4.device):
        rc = lib.carle_speed_update_f32(sums4.data_ptr(), _ptr(live4), n, com.data_ptr(), state.data_ptr(),
                                        velocity.data_ptr(), speed.data_ptr(), live_out.data_ptr(), _ptr(reward),
                                        _stream(sums4))
    if rc != 0:
        _lib.check(rc, "carle_speed_update_f32")


def puffer_update_f32(sums4, any_flag_ptr, ring, ring_state, live_total, reward):
    """PufferDetector tail (carle_puffer_update_f32); any_flag_ptr: raw device address of an int32 or None."""
    lib = _lib.load()
    with _on_device(sums4.device):
        rc = lib.carle_puffer_update_f32(sums4.data_ptr(), sums4.shape[0], any_flag_ptr, ring.data_ptr(),
                                         ring_state.data_ptr(), ring.numel(), live_total.data_ptr(), _ptr(reward),
                                         _stream(sums4))
    if rc != 0:
        _lib.check(rc, "carle_puffer_update_f32")


def wrapper_scalars_local(sums4, live4, com, state, velocity, live_out, any_flag_ptr, exch, what):
    """Per-shard part of SpeedDetector (what & 1) / PufferDetector (what & 2) under batch sharding: leaves this shard's
    four terms in ``exch`` (fp64[4]) for the all-reduce (carle_wrapper_scalars_local_f32)."""
    lib = _lib.load()
    with _on_device(sums4.device):
        rc = lib.carle_wrapper_scalars_local_f32(sums4.data_ptr(), _ptr(live4), sums4.shape[0], _ptr(com), _ptr(state),
                                                 _ptr(velocity), _ptr(live_out), any_flag_ptr, exch.data_ptr(), int(what),
                                                 _stream(sums4))
    if rc != 0:
        _lib.check(rc, "carle_wrapper_scalars_local_f32")


def wrapper_scalars_apply(exch, what, speed, ring, ring_state, live_total, reward):
    """The updates finished with the all-reduced ``exch`` (carle_wrapper_scalars_apply_f32): reward += global speed,
    PufferDetector's history on the global live-cell total."""
    lib = _lib.load()
    with _on_device(exch.device):
        rc = lib.carle_wrapper_scalars_apply_f32(exch.data_ptr(), int(what), _ptr(speed), _ptr(ring), _ptr(ring_state),
                                                 ring.numel() if ring is not None else 0, _ptr(live_total),
                                                 reward.data_ptr(), reward.numel(), _stream(exch))
    if rc != 0:
        _lib.check(rc, "carle_wrapper_scalars_apply_f32")


def morpho_match_f32(universe, action, patterns, reward=None, reward_scale=1.0, partial=None):
    """MorphoBonus (carle_morpho_match_f32, mcl.py:170-183): (N,2) = [max, min] over all patterns and positions of
    conv2d(|universe - action|, patterns); with ``reward`` (N,1): reward += reward_scale * (max + min).
    universe / action: fp32 (N,1,H,W) of the same shape; patterns: fp32 (K,1,8,8).  Returns (out, partial)."""
    lib = _lib.load()
    _chk_grid(universe, "universe", torch.float32)
    _chk_grid(action, "action", torch.float32)
    if action.shape != universe.shape or action.device != universe.device:
        raise ValueError(f"morpho_match_f32: the action {tuple(action.shape)} must have the universe's shape "
                         f"{tuple(universe.shape)} (mcl.py:172 subtracts them)")
    if not (patterns.is_cuda and patterns.dtype == torch.float32 and patterns.is_contiguous() and patterns.dim() == 4
            and patterns.shape[1:] == (1, 8, 8)):
        raise ValueError("morpho_match_f32: patterns must be a contiguous fp32 (K,1,8,8) device tensor")
    n, _, h, w = universe.shape
    need = lib.carle_morpho_partial_floats(n, h)
    if partial is None or partial.numel() < need or partial.device != universe.device:
        partial = torch.empty(max(need, 1), dtype=torch.float32, device=universe.device)
    out = torch.empty(n, 2, dtype=torch.float32, device=universe.device)
    with _on_device(universe.device):
        rc = lib.carle_morpho_match_f32(universe.data_ptr(), action.data_ptr(), patterns.data_ptr(), patterns.shape[0],
                                        n, h, w, partial.data_ptr(), out.data_ptr(), _ptr(reward), float(reward_scale),
                                        _stream(universe))
    if rc != 0:
        _lib.check(rc, "carle_morpho_match_f32")
    return out, partial


def bonus_nonempty_f32(reward, err, sums4, offset, sign, scale, live_total=None):
    """reward += scale * (offset + sign * err), zero for instances without a live cell (carle_bonus_nonempty_f32:
    PredictionBonus / SurpriseBonus.step, mcl.py:686-694, 715-723); live_total: optional 1-element fp32 output."""
    lib = _lib.load()
    n = err.shape[0]
    if not (reward.is_contiguous() and err.is_contiguous() and sums4.is_contiguous() and reward.numel() == n
            and sums4.shape[0] == n and sums4.shape[1] == 4):
        raise ValueError("bonus_nonempty_f32: contiguous reward (n,1), err (n,), sums4 (n,4) expected")
    with _on_device(reward.device):
        rc = lib.carle_bonus_nonempty_f32(reward.data_ptr(), err.data_ptr(), sums4.data_ptr(), float(offset), float(sign),
                                          float(scale), n, _ptr(live_total), _stream(reward))
    if rc != 0:
        _lib.check(rc, "carle_bonus_nonempty_f32")


def speed_puffer_f32(obs, w_h, w_w, sums4, counter, com, state, velocity, speed, live_out, any_flag_ptr, ring, ring_state,
                     live_total, reward):
    """SpeedDetector's reduction and update and PufferDetector's update in one launch (carle_speed_puffer_f32)."""
    lib = _lib.load()
    n = obs.shape[0]
    with _on_device(obs.device):
        rc = lib.carle_speed_puffer_f32(obs.data_ptr(), n, obs.numel() // max(n, 1), w_h.data_ptr(), w_w.data_ptr(),
                                        sums4.data_ptr(), counter.data_ptr(), com.data_ptr(), state.data_ptr(),
                                        velocity.data_ptr(), speed.data_ptr(), live_out.data_ptr(), any_flag_ptr,
                                        ring.data_ptr(), ring_state.data_ptr(), ring.numel(), live_total.data_ptr(),
                                        _ptr(reward), _stream(obs))
    if rc != 0:
        _lib.check(rc, "carle_speed_puffer_f32")


def env_step_f32_rnd(u_in, action, u_out, reward, rule, status, slot, publish_flags, pred_params,
                     rand_params, workspace, bonus, reward_scale, tables_valid,
                     event_handle, stream_env, stream_reward, merge_moments=False, deferred=False, partials=None,
                     bits_out=None, bits_in=None):
    """CARLE.step + RND2D bonus of the fresh observation in one library call (see carle_hip.h).  deferred: ONE
    launch (no epilogue kernel); with merge_moments it needs ``partials`` (fp64 scratch of
    carle_rnd_step_partials_doubles(n) elements that outlives the call until the next one on the status block).
    bits_in (int64 (N,1,64,1), with bits_out): the input state is read from this bit-packed copy, ``u_in`` is not touched
    (carle_env_step_f32_rnd_packed)."""
    lib = _lib.load()
    n, _, h, w = u_out.shape
    with _on_device(u_out.device):
        if bits_in is not None:
            rc = lib.carle_env_step_f32_rnd_packed(bits_in.data_ptr(), action.data_ptr(), u_out.data_ptr(),
                                                   reward.data_ptr(), n, h, w, action.shape[2], action.shape[3],
                                                   rule, status.data_ptr(), slot,
                                                   _step_flags(publish_flags, merge_moments, True), pred_params.data_ptr(),
                                                   rand_params.data_ptr(), workspace.data_ptr(), bonus.data_ptr(),
                                                   float(reward_scale), 2 if tables_valid else 0, _ptr(partials), event_handle,
                                                   stream_env, stream_reward, bits_out.data_ptr())
            if rc != 0:
                _lib.check(rc, "carle_env_step_f32_rnd_packed")
            return True
        if bits_out is None:
            rc = lib.carle_env_step_f32_rnd(u_in.data_ptr(), action.data_ptr(), u_out.data_ptr(),
                                            reward.data_ptr(), n, h, w, action.shape[2], action.shape[3],
                                            rule, status.data_ptr(), slot,
                                            _step_flags(publish_flags, merge_moments, deferred), pred_params.data_ptr(),
                                            rand_params.data_ptr(), workspace.data_ptr(), bonus.data_ptr(), float(reward_scale),
                                            2 if tables_valid else 0, _ptr(partials), event_handle,
                                            stream_env, stream_reward)
        else:
            # bits_out (int64 (N,1,64,1)): where the step is one launch it also leaves a bit-packed copy of the observation
            rc = lib.carle_env_step_f32_rnd_bits(u_in.data_ptr(), action.data_ptr(), u_out.data_ptr(),
                                                 reward.data_ptr(), n, h, w, action.shape[2], action.shape[3],
                                                 rule, status.data_ptr(), slot,
                                                 _step_flags(publish_flags, merge_moments, deferred), pred_params.data_ptr(),
                                                 rand_params.data_ptr(), workspace.data_ptr(), bonus.data_ptr(),
                                                 float(reward_scale), 2 if tables_valid else 0, _ptr(partials), event_handle,
                                                 stream_env, stream_reward, bits_out.data_ptr(), ctypes.byref(_bits_written))
    if rc != 0:
        _lib.check(rc, "carle_env_step_f32_rnd")
    return bits_out is not None and _bits_written.value != 0


def ca_step_f32_rnd(u_in, action, u_out, rule, status, slot, pred_params, rand_params, workspace, bonus,
                    tables_valid):
    """The fused 64x64 CA + RND2D kernel alone (no epilogue); bonus has N + 1 elements."""
    lib = _lib.load()
    n, _, h, w = u_in.shape
    ap, ah, aw = (None, 0, 0) if action is None else (action.data_ptr(), action.shape[2], action.shape[3])
    with _on_device(u_in.device):
        rc = lib.carle_ca_step_f32_rnd(u_in.data_ptr(), ap, u_out.data_ptr(), n, h, w, ah, aw, rule,
                                       _ptr(status), slot, pred_params.data_ptr(), rand_params.data_ptr(),
                                       workspace.data_ptr(), bonus.data_ptr(), 2 if tables_valid else 0,
                                       _stream(u_in))
    if rc != 0:
        _lib.check(rc, "carle_ca_step_f32_rnd")


def rnd2d_train(obs, pred_params, rand_params, workspace, partial, grad, bonus, reward, reward_scale,
                grad_scale, tables_valid, stream=None, binary=True):
    """RND2D bonus + accumulated predictor gradient in one pass: fused kernel for 64x64 binary obs
    (fp32 or packed), staged forward + backward kernels for any other square fp32 obs (`binary`
    False selects them at 64x64 too)."""
    lib = _lib.load()
    packed = obs.dtype == torch.int64
    n = obs.shape[0]
    h = obs.shape[2]
    w = 64 * obs.shape[3] if packed else obs.shape[3]
    flags = (1 if binary else 0) | (2 if tables_valid else 0)
    with _on_device(obs.device):
        fn = lib.carle_rnd2d_train_b1 if packed else lib.carle_rnd2d_train_f32
        rc = fn(obs.data_ptr(), n, h, w, pred_params.data_ptr(), rand_params.data_ptr(),
                workspace.data_ptr(), partial.data_ptr(), grad.data_ptr(), bonus.data_ptr(),
                _ptr(reward), float(reward_scale), float(grad_scale), flags,
                stream if stream is not None else _stream(obs))
    if rc != 0:
        _lib.check(rc, "carle_rnd2d_train")


def ae2d_train(obs, params, workspace, partial, grad, bonus, reward, reward_scale, grad_scale,
               tables_valid, stream=None, binary=True):
    """AE2D bonus + accumulated predictor gradient in one pass (paths as for rnd2d_train)."""
    lib = _lib.load()
    packed = obs.dtype == torch.int64
    n = obs.shape[0]
    h = obs.shape[2]
    w = 64 * obs.shape[3] if packed else obs.shape[3]
    flags = (1 if binary else 0) | (2 if tables_valid else 0)
    with _on_device(obs.device):
        fn = lib.carle_ae2d_train_b1 if packed else lib.carle_ae2d_train_f32
        rc = fn(obs.data_ptr(), n, h, w, params.data_ptr(), workspace.data_ptr(), partial.data_ptr(),
                grad.data_ptr(), bonus.data_ptr(), _ptr(reward), float(reward_scale), float(grad_scale),
                flags, stream if stream is not None else _stream(obs))
    if rc != 0:
        _lib.check(rc, "carle_ae2d_train")

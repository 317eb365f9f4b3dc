"""oracle/ca.py -- CPU checker for the CA step.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Two independent restatements of carle/env.py:188-242 (reference = riveSunder/carle):
  * ``c_*``   : ctypes bindings of oracle/ca_oracle.c (per-cell integer loops in C)
  * ``np_*``  : vectorised numpy restatement (np.roll on the torus), used to cross-check
                the C file and to handle tiny cases without a compiler

Parity status: PINNED against tests/golden/ca_cases.npz + ca_known.json (see
tests/test_oracle.py); the fixtures were produced by tools/make_golden.py from the genuine
reference imported in the build container.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def rule_mask(birth, survive):
    """(birth, survive) lists -> 18-bit mask: bit k = survive on k, bit 9+k = birth on k.

    Membership semantics of carle/env.py:221-224 (``elem == neighbourhood`` OR-ed over the
    list).  Elements outside 0..8 can never match a neighbour count and are dropped.
    """
    m = 0
    for k in survive:
        if 0 <= int(k) <= 8 and int(k) == k:
            m |= 1 << int(k)
    for k in birth:
        if 0 <= int(k) <= 8 and int(k) == k:
            m |= 1 << (9 + int(k))
    return m


def build(force=False):
    so = os.path.join(_HERE, "libca_oracle.so")
    src = os.path.join(_HERE, "ca_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libca_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int32)
        L.oracle_action_flags.argtypes = [fp, ctypes.c_int64, ip]
        L.oracle_action_flags.restype = None
        L.oracle_apply_action.argtypes = [fp, fp, fp, ctypes.c_int64] + [ctypes.c_int] * 4
        L.oracle_apply_action.restype = None
        L.oracle_ca_update.argtypes = [fp, fp, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_uint32]
        L.oracle_ca_update.restype = None
        L.oracle_env_step.argtypes = [fp, fp, fp, fp, ctypes.c_int64] + [ctypes.c_int] * 4 + \
                                     [ctypes.c_uint32, ip]
        L.oracle_env_step.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def c_env_step(universe, action, rule):
    """universe (N,1,H,W) f32, action (N,1,AH,AW) f32 or None -> (next universe, reset, flags)."""
    u = np.ascontiguousarray(universe, dtype=np.float32)
    n, _, h, w = u.shape
    out = np.empty_like(u)
    scratch = np.empty_like(u)
    flags = np.zeros(2, np.int32)
    if action is None:
        ap, ah, aw = None, 0, 0
    else:
        a = np.ascontiguousarray(action, dtype=np.float32)
        ah, aw = a.shape[2], a.shape[3]
        ap = _fp(a)
    r = lib().oracle_env_step(_fp(u), ap, _fp(out), _fp(scratch), n, h, w, ah, aw, int(rule),
                              flags.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    return out, bool(r), flags


# ---------------------------------------------------------------------------------------------
# numpy restatement (independent of the C file)
# ---------------------------------------------------------------------------------------------

def np_apply_action(universe, action):
    """carle/env.py:150-182: centre-pad the action window, toggle where action != 0."""
    alive = np.asarray(universe) != 0
    if action is None:
        return alive
    a = np.asarray(action) != 0          # NaN != 0 -> True, -0.0 -> False (torch.logical_xor)
    n, _, h, w = alive.shape
    ah, aw = a.shape[2], a.shape[3]
    oy, ox = (h - ah) // 2, (w - aw) // 2
    out = alive.copy()
    out[:, :, oy:oy + ah, ox:ox + aw] ^= a
    return out


def np_ca_update(alive, rule):
    """carle/env.py:219-229 on boolean grids (N,1,H,W)."""
    a = alive.astype(np.uint8)
    cnt = np.zeros_like(a)
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            if dy or dx:
                cnt += np.roll(np.roll(a, dy, axis=2), dx, axis=3)
    surv = ((rule >> cnt.astype(np.uint32)) & 1).astype(bool)
    birth = ((rule >> (9 + cnt.astype(np.uint32))) & 1).astype(bool)
    return np.where(alive, surv, birth)


def np_env_step(universe, action, rule):
    """carle/env.py:188-242 -> (next universe f32, reset fired)."""
    toggled = np_apply_action(universe, action)
    if action is not None:
        a = np.asarray(action, dtype=np.float64)
        if a.size and a.sum() == a.size:          # torch.mean(action) == 1.0, env.py:208
            return np.zeros(toggled.shape, np.float32), True
    return np_ca_update(toggled, rule).astype(np.float32), False


# ---------------------------------------------------------------------------------------------
# fixture helpers shared by the tests (unpack tests/golden/ca_cases.npz)
# ---------------------------------------------------------------------------------------------

def unpack(bits, shape):
    n = int(np.prod(shape))
    return np.unpackbits(bits, count=n, bitorder="little").reshape(shape).astype(np.float32)


def crop_oversize(action, h, w, ah, aw):
    """carle/env.py:164-167: a full-grid action is centre-cropped to the action window."""
    if action.shape[3] > aw:
        oy, ox = (w - aw) // 2, (h - ah) // 2
        return action[:, :, oy:action.shape[2] - oy, ox:action.shape[3] - ox]
    return action

"""ctypes loader for libcarle_hip.so (the C ABI declared in include/carle_hip.h).

Fails loudly: a missing / stale library is an error, never a reason to fall back to PyTorch ops.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcarle_hip.so")
ABI_VERSION = 2

_c_i64, _c_int, _c_u32, _c_vp = ctypes.c_int64, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/carle_hip.h one to one
SIGNATURES = {
    "carle_abi_version": (_c_int, []),
    "carle_last_error": (ctypes.c_char_p, []),
    "carle_build_arch": (ctypes.c_char_p, []),
    "carle_reload_knobs": (None, []),
    "carle_env_settle": (_c_int, [_c_vp, _c_vp]),
    "carle_ca_step_f32": (_c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_int, _c_int, _c_int, _c_int,
                                   _c_u32, _c_vp, _c_int, _c_vp]),
    "carle_ca_step_b1": (_c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_int, _c_int, _c_int, _c_int,
                                  _c_u32, _c_vp, _c_int, _c_vp]),
    "carle_ca_step_u8": (_c_int, [_c_vp, _c_vp, _c_int, _c_vp, _c_i64, _c_int, _c_int, _c_int, _c_int,
                                  _c_u32, _c_vp, _c_int, _c_vp]),
    "carle_env_step_u8": (_c_int, [_c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_i64, _c_int, _c_int, _c_int,
                                   _c_int, _c_u32, _c_vp, _c_int, _c_int, _c_vp]),
    "carle_ca_toggle_u8": (_c_int, [_c_vp, _c_vp, _c_int, _c_vp, _c_i64, _c_int, _c_int, _c_int, _c_int,
                                    _c_vp]),
    "carle_pack_u8_b1": (_c_int, [_c_vp, _c_vp, _c_i64, _c_int, _c_int, _c_vp, _c_int, _c_vp]),
    "carle_unpack_b1_u8": (_c_int, [_c_vp, _c_vp, _c_i64, _c_int, _c_int, _c_vp]),
    "carle_action_flags_f32": (_c_int, [_c_vp, _c_i64, _c_vp, _c_int, _c_vp]),
    "carle_ca_multistep_b1": (_c_int, [_c_vp, _c_vp, _c_i64, _c_int, _c_int, _c_u32, _c_int, _c_vp, _c_vp]),
    "carle_pack_f32_b1": (_c_int, [_c_vp, _c_vp, _c_i64, _c_int, _c_int, _c_vp, _c_int, _c_vp]),
    "carle_unpack_b1_f32": (_c_int, [_c_vp, _c_vp, _c_i64, _c_int, _c_int, _c_vp]),
    "carle_ca_toggle_f32": (_c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_int, _c_int, _c_int, _c_int,
                                     _c_vp]),
    "carle_rnd2d_workspace_floats": (_c_i64, [_c_i64, _c_int, _c_int]),
    "carle_rnd2d_bonus_f32": (_c_int, [_c_vp, _c_i64, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp,
                                       _c_vp, _c_vp, _c_vp, ctypes.c_float, _c_int, _c_vp]),
    "carle_rnd2d_bonus_b1": (_c_int, [_c_vp, _c_i64, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp,
                                      _c_vp, _c_vp, _c_vp, ctypes.c_float, _c_int, _c_vp]),
    "carle_rnd_ae_bonus": (_c_int, [_c_vp, _c_int, _c_i64, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, ctypes.c_float, _c_int,
                                    _c_vp, _c_vp, _c_vp, ctypes.c_float, _c_int, _c_vp, _c_vp, _c_vp, _c_vp]),
    "carle_rnd_ae_partials_doubles": (_c_i64, [_c_i64, _c_int, _c_int]),
    "carle_rnd_ae_train": (_c_int, [_c_vp, _c_int, _c_i64, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp,
                                    ctypes.c_float, ctypes.c_float, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp,
                                    ctypes.c_float, ctypes.c_float, _c_int, _c_vp, _c_vp, _c_vp, _c_vp]),
    "carle_stream_fork": (_c_int, [_c_vp, _c_vp, _c_vp]),
    "carle_env_step_f32_rnd": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_int, _c_int, _c_int,
                                        _c_int, _c_u32, _c_vp, _c_int, _c_int, _c_vp, _c_vp, _c_vp,
                                        _c_vp, ctypes.c_float, _c_int, _c_vp, _c_vp, _c_vp, _c_vp]),
    "carle_env_step_f32_rnd_bits": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_int, _c_int, _c_int,
                                             _c_int, _c_u32, _c_vp, _c_int, _c_int, _c_vp, _c_vp, _c_vp,
                                             _c_vp, ctypes.c_float, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp,
                                             ctypes.POINTER(ctypes.c_int)]),
    "carle_env_step_f32_rnd_packed": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_int, _c_int, _c_int,
                                               _c_int, _c_u32, _c_vp, _c_int, _c_int, _c_vp, _c_vp, _c_vp,
                                               _c_vp, ctypes.c_float, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp]),
    "carle_rnd_step_partials_doubles": (_c_i64, [_c_i64]),
    "carle_ca_step_f32_rnd": (_c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_int, _c_int, _c_int, _c_int, _c_u32,
                                       _c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_int, _c_vp]),
    "carle_rnd2d_train_partial_floats": (_c_i64, [_c_i64]),
    "carle_rnd2d_train_scratch_floats": (_c_i64, [_c_i64, _c_int, _c_int]),
    "carle_ae2d_train_scratch_floats": (_c_i64, [_c_i64, _c_int, _c_int]),
    "carle_rnd2d_train_f32": (_c_int, [_c_vp, _c_i64, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp,
                                       _c_vp, _c_vp, ctypes.c_float, ctypes.c_float, _c_int, _c_vp]),
    "carle_rnd2d_train_b1": (_c_int, [_c_vp, _c_i64, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp,
                                      _c_vp, _c_vp, ctypes.c_float, ctypes.c_float, _c_int, _c_vp]),
    "carle_ae2d_train_partial_floats": (_c_i64, [_c_i64]),
    "carle_ae2d_train_f32": (_c_int, [_c_vp, _c_i64, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp,
                                      _c_vp, ctypes.c_float, ctypes.c_float, _c_int, _c_vp]),
    "carle_ae2d_train_b1": (_c_int, [_c_vp, _c_i64, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp,
                                     _c_vp, ctypes.c_float, ctypes.c_float, _c_int, _c_vp]),
    "carle_adam_update_f32": (_c_int, [_c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, ctypes.c_double,
                                       ctypes.c_double, ctypes.c_double, ctypes.c_double, _c_i64, _c_int, _c_vp]),
    "carle_ae2d_workspace_floats": (_c_i64, [_c_i64, _c_int, _c_int]),
    "carle_ae2d_bonus_f32": (_c_int, [_c_vp, _c_i64, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp,
                                      _c_vp, ctypes.c_float, _c_int, _c_vp]),
    "carle_ae2d_bonus_b1": (_c_int, [_c_vp, _c_i64, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp,
                                     _c_vp, ctypes.c_float, _c_int, _c_vp]),
    "carle_ae2d_stats_partials_doubles": (_c_i64, [_c_i64]),
    "carle_ae2d_bonus_b1_stats": (_c_int, [_c_vp, _c_i64, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, ctypes.c_float, _c_int,
                                           _c_vp, _c_vp, _c_vp]),
    "carle_morpho_partial_floats": (_c_i64, [_c_i64, _c_int]),
    "carle_morpho_match_f32": (_c_int, [_c_vp, _c_vp, _c_vp, _c_int, _c_i64, _c_int, _c_int, _c_vp, _c_vp, _c_vp,
                                        ctypes.c_float, _c_vp]),
    "carle_bonus_nonempty_f32": (_c_int, [_c_vp, _c_vp, _c_vp, ctypes.c_float, ctypes.c_float, ctypes.c_float, _c_i64,
                                          _c_vp, _c_vp]),
    "carle_ae2d_predict_bonus_f32": (_c_int, [_c_vp, _c_vp, _c_i64, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp]),
    "carle_ae2d_predict_bonus_b1": (_c_int, [_c_vp, _c_vp, _c_i64, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_int, _c_vp]),
    "carle_ae2d_predict_train_b1": (_c_int, [_c_vp, _c_vp, _c_i64, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp,
                                             ctypes.c_float, _c_int, _c_vp]),
    "carle_ae2d_predict_train_f32": (_c_int, [_c_vp, _c_vp, _c_i64, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp,
                                              ctypes.c_float, _c_vp]),
    "carle_batch_moments_f32": (_c_int, [_c_vp, _c_i64, _c_vp, _c_vp, _c_vp]),
    "carle_merge_moments": (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_vp]),
    "carle_normalize_f32": (_c_int, [_c_vp, _c_vp, _c_i64, _c_vp, ctypes.c_double, _c_vp]),
    "carle_env_step_f32": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_int, _c_int, _c_int,
                                    _c_int, _c_u32, _c_vp, _c_int, _c_int, _c_vp]),
    "carle_env_step_f32_bits": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_int, _c_int, _c_int,
                                    _c_int, _c_u32, _c_vp, _c_int, _c_int, _c_vp, ctypes.POINTER(ctypes.c_int), _c_vp]),
    "carle_env_step_f32_packed": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_int, _c_int, _c_int,
                                           _c_int, _c_u32, _c_vp, _c_int, _c_int, _c_vp, _c_vp]),
    "carle_env_run_f32": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_int, _c_int, _c_int, _c_int, _c_u32,
                                   ctypes.c_float, ctypes.c_uint64, ctypes.c_uint64, _c_int, _c_vp, _c_int, _c_int,
                                   _c_vp]),
    "carle_env_step_b1": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_int, _c_int, _c_int,
                                   _c_int, _c_u32, _c_vp, _c_int, _c_int, _c_vp]),
    "carle_bernoulli_f32": (_c_int, [_c_vp, _c_i64, ctypes.c_float, ctypes.c_uint64, ctypes.c_uint64,
                                     _c_vp]),
    "carle_agent_rn_workspace_floats": (_c_i64, [_c_i64, _c_int, _c_int, _c_int]),
    "carle_agent_rn_forward_f32": (_c_int, [_c_vp, _c_i64, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_int,
                                            ctypes.c_float, _c_vp, _c_vp, _c_vp, _c_vp]),
    "carle_ca_step_f32_rand": (_c_int, [_c_vp, _c_vp, _c_i64, _c_int, _c_int, _c_int, _c_int, _c_u32,
                                        ctypes.c_float, ctypes.c_uint64, ctypes.c_uint64, _c_vp, _c_vp,
                                        _c_int, _c_vp]),
    "carle_env_step_f32_rand": (_c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_int, _c_int, _c_int, _c_int,
                                         _c_u32, ctypes.c_float, ctypes.c_uint64, ctypes.c_uint64,
                                         _c_vp, _c_vp, _c_int, _c_int, _c_vp]),
    "carle_ca_step_u8_rand": (_c_int, [_c_vp, _c_vp, _c_i64, _c_int, _c_int, _c_int, _c_int, _c_u32, ctypes.c_float,
                                       ctypes.c_uint64, ctypes.c_uint64, _c_vp, _c_vp, _c_int, _c_vp]),
    "carle_ca_step_b1_rand": (_c_int, [_c_vp, _c_vp, _c_i64, _c_int, _c_int, _c_int, _c_int, _c_u32, ctypes.c_float,
                                       ctypes.c_uint64, ctypes.c_uint64, _c_vp, _c_int, _c_vp]),
    "carle_env_step_u8_rand": (_c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_int, _c_int, _c_int, _c_int, _c_u32,
                                        ctypes.c_float, ctypes.c_uint64, ctypes.c_uint64, _c_vp, _c_int, _c_int,
                                        _c_vp]),
    "carle_env_step_b1_rand": (_c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_int, _c_int, _c_int, _c_int, _c_u32,
                                        ctypes.c_float, ctypes.c_uint64, ctypes.c_uint64, _c_vp, _c_int, _c_int,
                                        _c_vp]),
    "carle_instance_sums_f32": (_c_int, [_c_vp, _c_i64, _c_i64, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp]),
    "carle_speed_update_f32": (_c_int, [_c_vp, _c_vp, _c_i64, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp]),
    "carle_puffer_update_f32": (_c_int, [_c_vp, _c_i64, _c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_vp]),
    "carle_speed_puffer_f32": (_c_int, [_c_vp, _c_i64, _c_i64, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp,
                                        _c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_vp]),
    "carle_wrapper_scalars_local_f32": (_c_int, [_c_vp, _c_vp, _c_i64, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_int,
                                                 _c_vp]),
    "carle_wrapper_scalars_apply_f32": (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_i64, _c_vp]),
    "carle_rle_encode_body": (_c_i64, [_c_vp, _c_int, _c_int, _c_vp, _c_i64]),
    "carle_rle_decode_body": (_c_int, [ctypes.c_char_p, _c_i64, _c_vp, _c_int, _c_int]),
    "carle_ca_finish": (_c_int, [_c_vp, _c_i64, _c_vp, _c_i64, _c_vp, _c_int, _c_int, _c_vp]),
}

_lib = None


def source_hash():
    """sha256 over the kernel sources the library is built from (carle_amd/csrc/*.hip, *.hpp and
    include/carle_hip.h, in name order).  Profiles record it so that a PMC figure is only quoted
    for the kernels it was measured on (bench.py, tools/pmc_summary.py)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.hpp")))
    files.append(os.path.join(os.path.dirname(_HERE), "include", "carle_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


class CarleHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle with typed entry points."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CarleHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (or `make -C carle_amd/csrc`).  carle_amd has no CPU/PyTorch fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise CarleHipError(f"{LIB_PATH} does not export {name}; rebuild the library") from e
        fn.restype = res
        fn.argtypes = args
    got = lib.carle_abi_version()
    if got != ABI_VERSION:
        raise CarleHipError(f"libcarle_hip.so ABI {got} != expected {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().carle_last_error().decode(errors="replace")
        raise CarleHipError(f"{what} failed (code {rc}): {msg}")

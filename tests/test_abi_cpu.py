"""CPU-only: the C-ABI library loads and exports every symbol include/carle_hip.h declares,
and the host-side mirror of the reference surface behaves (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest
import torch

import golden_util as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "carle_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(carle_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from carle_amd import _lib
    lib = _lib.load()
    names = _declared_symbols()
    assert "carle_ca_step_f32" in names and "carle_ca_finish" in names
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for nm in names:
        assert hasattr(raw, nm), f"{nm} declared in carle_hip.h but not exported"
        assert nm in _lib.SIGNATURES, f"{nm} has no ctypes signature in carle_amd/_lib.py"
    assert set(_lib.SIGNATURES) == set(names)
    assert lib.carle_abi_version() == _lib.ABI_VERSION == 2
    assert lib.carle_build_arch() == b"gfx950"


def test_argument_errors_do_not_touch_the_gpu():
    """Argument validation happens before any HIP call: exercised here with NULL / bad dims."""
    from carle_amd import _lib
    lib = _lib.load()
    assert lib.carle_ca_step_f32(None, None, None, 1, 64, 64, 64, 64, 0, None, 0, None) == 1
    assert b"NULL" in lib.carle_last_error()
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    q = ctypes.c_void_p(p.value + 4096 * 64)
    assert lib.carle_ca_step_f32(p, None, q, 1, 64, 32, 0, 0, 0, None, 0, None) == 2     # H != W
    assert lib.carle_ca_step_f32(p, None, q, 1, 63, 63, 0, 0, 0, None, 0, None) == 2     # odd
    assert lib.carle_ca_step_f32(p, p, q, 1, 64, 64, 63, 63, 0, None, 0, None) == 2      # (H-AH) odd
    assert lib.carle_ca_step_f32(p, None, q, 1, 64, 64, 0, 0, 1 << 18, None, 0, None) == 3
    assert lib.carle_ca_step_f32(p, None, p, 1, 64, 64, 0, 0, 0, None, 0, None) == 5     # alias
    assert lib.carle_ca_finish(None, 0, None, 0, None, 0, 0, None) == 1
    # uint8 format, multi-generation kernel, stand-alone predicates, conversions
    b8 = (ctypes.c_uint8 * 64)()
    p8 = ctypes.cast(b8, ctypes.c_void_p)
    q8 = ctypes.c_void_p(p8.value + 4096 * 16)
    assert lib.carle_ca_step_u8(None, None, 0, None, 1, 64, 64, 0, 0, 0, None, 0, None) == 1
    assert lib.carle_ca_step_u8(p8, p8, 7, q8, 1, 64, 64, 64, 64, 0, None, 0, None) == 6      # action_fmt
    assert b"action_fmt" in lib.carle_last_error()
    assert lib.carle_ca_step_u8(p8, None, 0, p8, 1, 64, 64, 0, 0, 0, None, 0, None) == 5      # alias
    assert lib.carle_ca_step_u8(p8, None, 0, q8, 1, 64, 64, 0, 0, 0, None, 2, None) == 2      # slot
    assert lib.carle_env_step_u8(p8, None, 0, q8, None, 1, 64, 64, 0, 0, 0, None, 0, 1, None) == 1
    assert lib.carle_ca_toggle_u8(p8, None, 0, q8, 1, 64, 64, 64, 64, None) == 1
    assert lib.carle_ca_multistep_b1(p8, q8, 1, 96, 96, 0, 3, None, None) == 2               # W not packed-able
    assert lib.carle_ca_multistep_b1(p8, q8, 1, 64, 64, 0, -1, None, None) == 2              # k < 0
    assert lib.carle_ca_multistep_b1(p8, q8, 1, 64, 64, 1 << 18, 1, None, None) == 3
    assert lib.carle_ca_multistep_b1(p8, q8, 0, 64, 64, 0, 1, None, None) == 0               # empty batch
    assert lib.carle_ca_multistep_b1(p8, ctypes.c_void_p(p8.value + 256), 1, 64, 64, 0, 1, None, None) == 5
    assert lib.carle_action_flags_f32(p, 10, None, 0, None) == 1
    assert lib.carle_action_flags_f32(None, 0, p, 0, None) == 0                              # nothing to scan
    assert lib.carle_pack_u8_b1(p8, q8, 1, 64, 65, None, 0, None) == 2
    assert lib.carle_unpack_b1_u8(p8, ctypes.c_void_p(q8.value + 4), 1, 64, 64, None) == 4    # alignment
    assert lib.carle_env_step_f32_rnd(p, None, q, p, 1, 64, 64, 0, 0, 0, None, 0, 1, p, p, p, p, 1.0, 0,
                                      None, None, None, None) == 1                                 # status NULL
    # RandomNetworkAgent forward
    assert lib.carle_agent_rn_forward_f32(None, 1, 64, 64, p, p, p, 64, 0.1, p, None, p, None) == 1
    assert lib.carle_agent_rn_forward_f32(p, 1, 60, 64, p, p, p, 64, 0.1, p, None, p, None) == 2   # H % 8
    assert lib.carle_agent_rn_forward_f32(p, 0, 64, 64, p, p, p, 64, 0.1, p, None, p, None) == 0   # empty batch
    assert lib.carle_agent_rn_workspace_floats(16, 256, 256, 4096) == 16 * 4096 + 16 * 16 * 4096   # features + 16 K splits
    assert lib.carle_agent_rn_workspace_floats(4096, 64, 64, 4096) == 4096 * 256                   # no split


def test_header_is_plain_c_and_links(tmp_path):
    """include/carle_hip.h compiled by a C compiler (gcc -std=c99) and linked against the library:
    the boundary really is a C ABI."""
    import shutil
    import subprocess
    from carle_amd import _lib
    _lib.load()
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    exe = tmp_path / "abi_errors"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c_abi", "abi_errors.c"), "-o", str(exe),
                    "-L", libdir, "-lcarle_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"],
                   check=True, capture_output=True, text=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "abi ok" in out.stdout, out.stdout + out.stderr


def test_env_surface_without_gpu():
    from carle_amd.env import CARLE
    k = G.known()
    env = CARLE()
    d = k["defaults"]
    assert (env.width, env.height, env.action_width, env.action_height, env.instances) == \
        (d["width"], d["height"], d["action_width"], d["action_height"], d["instances"])
    assert env.birth == d["birth"] and env.survive == d["survive"] and env.inner_env is None
    assert list(env.state_dict().keys()) == d["state_dict_keys"]
    assert env.state_dict()["neighborhood.weight"].reshape(-1).tolist() == d["weight"]
    assert tuple(env.state_dict()["neighborhood.weight"].shape) == (1, 1, 3, 3)
    env.birth_rule_from_string("asdfasdfB0357*!@#!@$%")
    env.survive_rule_from_string("S2468")
    assert env.birth == k["parse_b"] and env.survive == k["parse_s"]
    env.rules_from_string("B368/S245")
    assert [env.birth, env.survive] == k["parse_rules_b368s245"]
    env.rules_from_string("b3/s23")
    assert [env.birth, env.survive] == k["parse_rules_lower"]
    env.rules_from_string("B9a33/S")
    assert [env.birth, env.survive] == k["parse_rules_dup_empty"]
    pad = env.action_padding(torch.ones(1, 1, 64, 64))
    nz = pad[0, 0].nonzero()
    assert [int(nz[:, 0].min()), int(nz[:, 0].max()), int(nz[:, 1].min()), int(nz[:, 1].max())] \
        == k["default_action_window"]
    env.strict_reference_errors = True
    with pytest.raises(TypeError):
        env.rule_mask()


def test_non_moore_state_dict_is_rejected():
    from carle_amd.env import CARLE
    env = CARLE(device="cpu")
    sd = env.state_dict()
    env.load_state_dict(sd)                       # the Moore kernel loads fine
    bad = {"neighborhood.weight": torch.ones(1, 1, 3, 3)}
    with pytest.raises(NotImplementedError):
        env.load_state_dict(bad)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_compute_fails_loudly_without_gpu():
    from carle_amd.env import CARLE
    from carle_amd._lib import CarleHipError
    env = CARLE(width=64, height=64)
    with pytest.raises(CarleHipError):
        env.reset()


def test_reference_import_paths_resolve_to_this_build():
    import carle.env, carle.mcl, carle.agents, carle.train_mcl
    import evaluation.eval, evaluation.submission
    import carle_amd.env, carle_amd.mcl
    assert carle.env.CARLE is carle_amd.env.CARLE
    assert carle.mcl.RND2D is carle_amd.mcl.RND2D and carle.mcl.AE2D is carle_amd.mcl.AE2D
    assert callable(carle.train_mcl.train) and callable(evaluation.eval.evaluate)
    assert issubclass(evaluation.submission.SubmissionAgent, carle.agents.RandomAgent)

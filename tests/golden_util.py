"""Helpers to read tests/golden/*.npz (written by tools/make_golden.py from the reference)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def unpack(bits, shape):
    n = int(np.prod(shape))
    return np.unpackbits(bits, count=n, bitorder="little").reshape(shape).astype(np.float32)


class CACase:
    def __init__(self, z, name):
        self.name = name
        self.meta = json.loads(bytes(z[f"{name}/meta"]).decode())
        m = self.meta
        n, s = m["n"], m["size"]
        self.n, self.size, self.asize = n, s, m["action_size"]
        self.birth, self.survive = m["birth"], m["survive"]
        self.steps = m["steps"]
        self.record = m["record"]
        self.init = unpack(z[f"{name}/init"], (n, 1, s, s))
        ashape = (m["steps"],) + tuple(m["action_shape"])
        if m["actions_binary"]:
            self.actions = unpack(z[f"{name}/actions"], ashape)
        else:
            self.actions = z[f"{name}/actions"].astype(np.float32)
        self.obs = unpack(z[f"{name}/obs"], (len(self.record), n, 1, s, s))
        self.step_number = z[f"{name}/step_number"]
        self.steps_since_action = z[f"{name}/steps_since_action"]


_CACHE = {}


def ca_cases():
    if "ca" not in _CACHE:
        z = np.load(os.path.join(GOLDEN, "ca_cases.npz"))
        names = sorted(k[:-5] for k in z.files if k.endswith("/meta"))
        _CACHE["ca"] = [CACase(z, nm) for nm in names]
    return _CACHE["ca"]


def ca_case_names():
    z = np.load(os.path.join(GOLDEN, "ca_cases.npz"))
    return sorted(k[:-5] for k in z.files if k.endswith("/meta"))


def ca_case(name):
    for c in ca_cases():
        if c.name == name:
            return c
    raise KeyError(name)


def predicate_case_names():
    z = np.load(os.path.join(GOLDEN, "ca_predicates.npz"))
    return sorted(k[:-5] for k in z.files if k.endswith("/meta"))


def predicate_case(name):
    """Cases of ca_predicates.npz: actions whose values cancel / average to exactly 1.0."""
    if "pred" not in _CACHE:
        z = np.load(os.path.join(GOLDEN, "ca_predicates.npz"))
        _CACHE["pred"] = {nm: CACase(z, nm) for nm in predicate_case_names()}
    return _CACHE["pred"][name]


def known():
    with open(os.path.join(GOLDEN, "ca_known.json")) as f:
        return json.load(f)


def load_sd(z, prefix):
    """-> ordered dict name -> numpy array (state_dict stored by tools/make_golden.py)."""
    keys = json.loads(bytes(z[f"{prefix}/keys"]).decode())
    return {k: z[f"{prefix}/t{i}"] for i, k in enumerate(keys)}


def nets(size):
    key = f"nets{size}"
    if key not in _CACHE:
        _CACHE[key] = np.load(os.path.join(GOLDEN, f"nets_{size}.npz"))
    return _CACHE[key]


class PredictionCase:
    """A trajectory of tests/golden/prediction.npz (PredictionBonus / SurpriseBonus driven by the reference)."""

    def __init__(self, z, name):
        self.name = name
        m = self.meta = json.loads(bytes(z[f"{name}/meta"]).decode())
        n, s, t = m["n"], m["size"], m["steps"]
        self.n, self.size, self.steps = n, s, t
        self.batch_size, self.scale, self.cls = m["batch_size"], m["scale"], m["cls"]
        self.init = unpack(z[f"{name}/init"], (n, 1, s, s))
        self.actions = unpack(z[f"{name}/actions"], (t, n, 1, 64, 64))
        self.rewards = z[f"{name}/rewards"]
        self.live_cells = z[f"{name}/live_cells"]
        self.final_obs = unpack(z[f"{name}/final_obs"], (n, 1, s, s))
        self.init_sd = load_sd(z, f"{name}/init_sd")
        self.final_sd = load_sd(z, f"{name}/final_sd")


def prediction_file():
    if "prediction" not in _CACHE:
        _CACHE["prediction"] = np.load(os.path.join(GOLDEN, "prediction.npz"))
    return _CACHE["prediction"]


PREDICTION_CASES = ("pred64", "pred64_scaled", "surprise64", "pred256_glider")


def prediction_case(name):
    return PredictionCase(prediction_file(), name)


class MorphoCase:
    """tests/golden/morpho.npz: MorphoBonus driven by the reference on the pattern files this package ships."""

    def __init__(self, name):
        if "morpho" not in _CACHE:
            _CACHE["morpho"] = np.load(os.path.join(GOLDEN, "morpho.npz"))
        z = _CACHE["morpho"]
        m = self.meta = json.loads(bytes(z[f"{name}/meta"]).decode())
        self.n, self.steps, self.scale = m["n"], m["steps"], m["scale"]
        self.birth, self.survive, self.extra_rle = m["birth"], m["survive"], m["extra_rle"]
        self.patterns = z[f"{name}/patterns"]
        self.reset_obs = unpack(z[f"{name}/reset_obs"], (self.n, 1, 64, 64))
        self.actions = unpack(z[f"{name}/actions"], (self.steps, self.n, 1, 64, 64))
        self.rewards = z[f"{name}/rewards"]
        self.final_obs = unpack(z[f"{name}/final_obs"], (self.n, 1, 64, 64))

#!/usr/bin/env python3
"""Differential fuzzing of the reward networks against oracle/nets.py (fp32 torch.nn.functional restatement of
carle/mcl.py:257-298, 445-481) and fp64 autograd, run on the GPU box: random square sizes (multiples of 8 up to 264,
the kernel families' borders 56 / 64 / 72 / 120 / 128 / 136 / 192 / 256 included), batches of 1 .. 70 grids, binary and
non-binary observations, fp32 and bit-packed inputs, randomly scaled parameters, through the wrappers' own entry points:
``get_bonus_only`` (forward: fused 64x64 / tiled / staged kernels) within 1e-5 absolute, ``get_bonus_accumulate`` with
batch_size 1 (forward + gradient + the Adam kernel) against one torch.optim.Adam step on the fp64 autograd gradient.
Not part of the test suite (open-ended).      python tests/tools/fuzz_nets.py [seconds] [seed]"""
import copy
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from carle_amd import hip                      # noqa: E402
from carle_amd.env import CARLE                # noqa: E402
import carle_amd.mcl as M                      # noqa: E402
from oracle import nets as ON                  # noqa: E402

SIZES = [8, 16, 24, 40, 56, 64, 64, 64, 72, 96, 120, 128, 128, 136, 192, 200, 256, 256, 264]


def _bonus64(cls, obs, sd):
    """oracle/nets.py's functions on fp64 tensors (its helpers cast parameters to fp32: restated here in fp64)."""
    import torch.nn.functional as F
    if cls == "RND2D":
        def net(prefix, c1, c2, lin):
            x = F.conv2d(obs, sd[prefix + c1 + ".weight"], sd[prefix + c1 + ".bias"], padding=1)
            x = F.max_pool2d(F.max_pool2d(F.relu(x), 2), 2)
            x = F.max_pool2d(F.relu(F.conv2d(x, sd[prefix + c2 + ".weight"], sd[prefix + c2 + ".bias"], padding=1)), 2)
            return torch.tanh(F.linear(x.flatten(1), sd[prefix + lin + ".weight"], sd[prefix + lin + ".bias"]))
        t, q = net("random_network.", "0", "4", "8"), net("predictor.", "0", "5", "11")
        return torch.mean(torch.abs((t - q) ** 2), dim=[1]).numpy()
    p = "predictor."
    x = F.max_pool2d(F.relu(F.conv2d(obs, sd[p + "0.weight"], sd[p + "0.bias"], padding=1)), 2)
    x = F.max_pool2d(F.relu(F.conv2d(x, sd[p + "4.weight"], sd[p + "4.bias"], padding=1)), 2)
    x = F.relu(F.conv_transpose2d(x, sd[p + "8.weight"], sd[p + "8.bias"], stride=2, padding=1))
    x = torch.sigmoid(F.conv_transpose2d(x, sd[p + "11.weight"], sd[p + "11.bias"], stride=2, padding=1))
    return torch.mean(torch.abs((obs - x) ** 2), dim=[1, 2, 3]).numpy()


def one_case(case, rng):
    cls = str(rng.choice(["RND2D", "AE2D"]))
    size = int(rng.choice(SIZES))
    n = int(rng.choice([1, 2, 3, 5, 9, 17, 33, 70])) if size <= 136 else int(rng.integers(1, 6))
    binary = bool(rng.random() < 0.7)
    packed = binary and size % 64 == 0 and bool(rng.random() < 0.4)
    torch.manual_seed(int(rng.integers(0, 2 ** 31 - 1)))
    env = CARLE(width=size, height=size, action_width=min(64, size), action_height=min(64, size), instances=n, device="cuda:0")
    w = getattr(M, cls)(env)
    w.eval(); w.reset()
    w.assume_binary_obs = binary                             # a caller's promise: the tensor holds 0 / 1 only (table kernels)
    with torch.no_grad():                                    # scaled parameters: away from the default init's regime
        for p in w.predictor.parameters():
            p.mul_(float(rng.choice([0.3, 1.0, 1.0, 2.5])))
    dens = float(rng.choice([0.0, 0.05, 0.3, 0.5, 0.95, 1.0]))
    obs = (rng.random((n, 1, size, size)) < dens).astype(np.float32)
    if not binary:
        obs = (obs * rng.random((n, 1, size, size)) * 2.0 - 0.3 * (rng.random((n, 1, size, size)) < 0.1)).astype(np.float32)
    d_obs = torch.from_numpy(obs).cuda()
    given = hip.pack_f32_b1(d_obs) if packed else d_obs
    desc = f"case {case}: {cls} size={size} n={n} binary={binary} packed={packed} density={dens}"
    sd = {k: v.detach().cpu().numpy() for k, v in w.state_dict().items()}
    with torch.no_grad():
        exp = (ON.rnd_bonus if cls == "RND2D" else ON.ae_bonus)(torch.from_numpy(obs), sd).numpy()
        # the same in fp64: scaled parameters push the bonus to O(1) and beyond, where the fp32 oracle's own rounding
        # exceeds the 1e-5 the north star states for bonuses of ~0.02 -- the bound is 1e-5 of the bonus' size plus a few
        # times what the fp32 oracle itself is off by
        sd64 = {k: torch.from_numpy(v).double() for k, v in sd.items()}
        exp64 = _bonus64(cls, torch.from_numpy(obs).double(), sd64)
    tol = 1e-5 * np.maximum(1.0, np.abs(exp64)) + 8.0 * np.abs(exp - exp64)
    got = w.get_bonus_only(given).cpu().numpy()
    err = np.abs(got - exp64)
    if not bool((err <= tol).all()):
        i = int(np.argmax(err - tol))
        raise AssertionError(f"{desc}: bonus {got[i]:.8g} differs from the fp64 value {exp64[i]:.8g} by {err[i]:.3e} "
                             f"(fp32 oracle: {abs(exp[i] - exp64[i]):.3e})")
    # one update on this observation: kernels' gradient + Adam kernel against fp64 autograd + torch Adam
    ref = copy.deepcopy(w.predictor).double().eval()
    o64 = torch.from_numpy(obs).double().cuda()
    if cls == "RND2D":
        with torch.no_grad():
            target = copy.deepcopy(w.random_network).double().eval()(o64)
        loss = torch.mean(torch.abs((target - ref(o64)) ** 2), dim=[1])
    else:
        loss = torch.mean(torch.abs((o64 - ref(o64).reshape(o64.shape)) ** 2), dim=[1, 2, 3])
    opt = torch.optim.Adam(ref.parameters(), lr=w.optimizer.param_groups[0]["lr"])
    torch.mean(loss).backward()
    grads = [p.grad.clone() for p in ref.parameters()]
    opt.step()
    w.batch_size = 1
    bonus2 = w.get_bonus_accumulate(given).cpu().numpy()
    err2 = np.abs(bonus2 - exp64)
    if not bool((err2 <= tol).all()):
        raise AssertionError(f"{desc}: training-path bonus differs from the fp64 value by {float(err2.max()):.3e}")
    lr = w.optimizer.param_groups[0]["lr"]
    for (name, p), q, g in zip(w.predictor.named_parameters(), ref.parameters(), grads):
        # Adam's first step moves a weight by lr * g / (|g| + eps): a gradient component that is off by e (the kernels: up
        # to ~6e-7 of the tensor's largest component, profiles/r03_grad_error_table.txt) moves it by lr * e / |g| more --
        # and sums of many cancelling terms (conv1 weights over 33 x 128 x 128 pixels) are off by eps * sum|terms|, which
        # the largest component does not bound.  Per element: lr * min(2, 1e-4 * max|g| / |g|) -- tight where the gradient
        # is large, void at the noise floor (there the sign of noise decides between +lr and -lr); a wrong tap or a
        # dropped element shows at 1e-2 and more
        ga = g.abs()
        tol_e = lr * torch.clamp(1e-4 * float(ga.max()) / ga.clamp_min(1e-30), max=2.0) + 2e-6
        d = (p.detach().double() - q.detach()).abs()
        if bool((d > tol_e).any()):
            i = int(torch.argmax(d - tol_e))
            raise AssertionError(f"{desc}: parameter {name} after one update differs by {float(d.flatten()[i]):.3e} where "
                                 f"{float(tol_e.flatten()[i]):.3e} is allowed (lr {lr}, |g| {float(ga.flatten()[i]):.3e}, "
                                 f"max|g| {float(ga.max()):.3e})")
    return desc


SEED_MULT = 7919


def main(budget=60.0, seed0=0, cases=None):
    """Runs cases until `budget` seconds have passed (or exactly `cases` of them: tests/test_gpu_fuzz.py)."""
    t_end = time.time() + budget
    case = 0
    while (case < cases) if cases is not None else (time.time() < t_end):
        d = one_case(case, np.random.default_rng(seed0 * SEED_MULT + case))
        case += 1
        if case % 20 == 0 and cases is None:
            print(f"{case} cases ok (last: {d})", flush=True)
    print(f"fuzz_nets: {case} cases, no difference")
    return case


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)

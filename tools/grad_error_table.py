#!/usr/bin/env python3
"""Per-layer error of every predictor-gradient kernel (fused 64x64, tiled, staged; RND2D and AE2D) against fp64 autograd
of the same loss on the same weights, with fp32 autograd beside it (what plain fp32 arithmetic costs).

    python tools/grad_error_table.py > profiles/r03_grad_error_table.txt        (on the GPU box)

Loss: sum_t mean_n loss_n / batch_size with batch_size = 1 (carle/mcl.py:343-386, 504-547), eval-mode semantics.
Per parameter tensor: max |kernel - fp64| / max |fp64|  (error relative to the layer's largest gradient component) and
the relative L2 error; the same two numbers for fp32 autograd (MIOpen / ATen) against fp64."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from carle_amd.env import CARLE           # noqa: E402
import carle_amd.mcl as M                 # noqa: E402


def autograd_grads(w, obs, dtype):
    """Gradient of mean_n(loss_n) through the wrapper's nn.Sequential modules in `dtype`."""
    import copy
    pred = copy.deepcopy(w.predictor).to(dtype).eval()
    o = obs.to(dtype)
    if w.my_name == "RND2D":
        rnd = copy.deepcopy(w.random_network).to(dtype).eval()
        with torch.no_grad():
            target = rnd(o)
        loss = torch.mean(torch.abs((target - pred(o)) ** 2), dim=[1])
    else:
        loss = torch.mean(torch.abs((o - pred(o).reshape(o.shape)) ** 2), dim=[1, 2, 3])
    torch.mean(loss).backward()
    return [p.grad.detach().to(torch.float64) for p in pred.parameters()], loss.detach().to(torch.float64)


def one(cls, size, n, path, density=0.3, seed=0):
    torch.manual_seed(seed)
    env = CARLE(width=size, height=size, action_width=min(64, size), action_height=min(64, size), instances=n, device="cuda:0")
    w = getattr(M, cls)(env)
    w.eval(); w.reset()
    g = torch.Generator(device="cuda").manual_seed(seed + 1)
    obs = (torch.rand(n, 1, size, size, device="cuda", generator=g) < density).float()
    if path == "staged":
        w.assume_binary_obs = False                   # staged forward + staged backward (fp32 convolutions)
        feed = obs.clone()
    else:
        w.assume_binary_obs = True                    # fused 64x64 / tiled kernels (table look-ups)
        feed = obs
    bonus = torch.empty(n, device="cuda")
    assert w._fused_train(feed, None, bonus, None, batch=1)
    got = w._grad.detach().to(torch.float64)
    ref64, loss64 = autograd_grads(w, obs, torch.float64)
    ref32, _ = autograd_grads(w, obs, torch.float32)
    names = [k for k, _ in w.predictor.named_parameters()]
    rows, off = [], 0
    for name, r64, r32 in zip(names, ref64, ref32):
        k = r64.numel()
        mine = got[off:off + k].reshape(r64.shape)
        off += k
        scale = max(r64.abs().max().item(), 1e-300)      # (a dead channel has an all-zero gradient: errors print as absolute)
        l2 = max(r64.norm().item(), 1e-300)
        if r64.abs().max().item() == 0.0:
            scale = l2 = 1.0
        rows.append((name, tuple(r64.shape), scale,
                     (mine - r64).abs().max().item() / scale, (mine - r64).norm().item() / l2,
                     (r32 - r64).abs().max().item() / scale, (r32 - r64).norm().item() / l2))
    berr = (bonus.to(torch.float64) - loss64).abs().max().item()
    return rows, berr


def main():
    cases = [("RND2D", 64, 256, "fused64"), ("AE2D", 64, 256, "fused64"), ("RND2D", 64, 256, "staged"), ("AE2D", 64, 256, "staged"),
             ("RND2D", 256, 16, "tiled"), ("AE2D", 256, 16, "tiled"), ("RND2D", 256, 16, "staged"), ("AE2D", 256, 16, "staged"),
             ("RND2D", 64, 4096, "fused64"), ("AE2D", 64, 4096, "fused64")]
    print("gradient kernels vs fp64 autograd (same weights, same observations); err = max|d| / max|grad fp64| per tensor, "
          "l2 = ||d|| / ||grad fp64||")
    print(f"{'net':6s} {'size':>4s} {'N':>5s} {'path':8s} {'parameter':12s} {'shape':16s} {'max|g|':>9s} "
          f"{'kernel err':>10s} {'kernel l2':>10s} {'fp32 AG err':>11s} {'fp32 AG l2':>10s}")
    worst = {}
    for cls, size, n, path in cases:
        # AE2D under torch.manual_seed(0) has a dead ReLU behind its first transposed convolution (every gradient but the
        # last bias is exactly zero, in every implementation): take the first seed whose network is alive
        for seed in range(32):
            rows, berr = one(cls, size, n, path, seed=seed)
            if all(r[2] != 1.0 or r[3] != 0.0 for r in rows):
                break
        print(f"# {cls} {size}x{size} N={n} {path}: seed {seed}")
        for name, shape, scale, e, l2, e32, l232 in rows:
            print(f"{cls:6s} {size:4d} {n:5d} {path:8s} {name:12s} {str(shape):16s} {scale:9.2e} {e:10.2e} {l2:10.2e} "
                  f"{e32:11.2e} {l232:10.2e}")
            key = (cls, path)
            worst[key] = max(worst.get(key, 0.0), e)
        print(f"{cls:6s} {size:4d} {n:5d} {path:8s} bonus vs fp64 loss: max abs {berr:.2e}")
    print("\nworst kernel err per (net, path):")
    for k, v in worst.items():
        print(f"  {k[0]:6s} {k[1]:8s} {v:.2e}")


if __name__ == "__main__":
    main()

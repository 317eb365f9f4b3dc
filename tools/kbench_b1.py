#!/usr/bin/env python3
"""Times the bit-packed step kernel (carle_ca_step_b1, no epilogue) at N x SxS: no action, packed
64x64 action window with and without the status flags.  HIP events around back-to-back launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd import hip                      # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
rule = (1 << 12) | (1 << 15) | 0xC              # B36/S23
g = torch.Generator(device="cuda").manual_seed(1)
a = torch.randint(-2 ** 63, 2 ** 63 - 1, (n, 1, size, size // 64), device="cuda", generator=g, dtype=torch.int64)
b = torch.empty_like(a)
RING = int(os.environ.get("KB_RING", "6"))         # action tensors in rotation (6 = bench.py's ring)
ring = [hip.pack_f32_b1((torch.rand(n, 1, 64, 64, device="cuda", generator=g) <= 0.1).float()) for _ in range(RING)]
st = hip.new_status(a.device)
bufs = [a, b]
cases = {"noact": lambda i: hip.ca_step_b1(bufs[i & 1], None, bufs[(i & 1) ^ 1], rule, st, 0),
         "act": lambda i: hip.ca_step_b1(bufs[i & 1], ring[i % RING], bufs[(i & 1) ^ 1], rule, st, 0),
         "act_nostatus": lambda i: hip.ca_step_b1(bufs[i & 1], ring[i % RING], bufs[(i & 1) ^ 1], rule, None, 0)}
alg = {"noact": n * size * size // 4, "act": n * (size * size // 4 + 512), "act_nostatus": n * (size * size // 4 + 512)}
blocks = sys.argv[3].split(",") if len(sys.argv) > 3 else ["0"]
gridmaps = sys.argv[4].split(",") if len(sys.argv) > 4 else [os.environ.get("CARLE_B1_DBG", "")]
for rep in range(3 if len(gridmaps) > 1 else 2):
 for gmap in gridmaps:                       # experiment knob of the half-grid-unit kernel, interleaved A/B
  if gmap:
      os.environ["CARLE_B1_DBG"] = gmap
  for blk in blocks:
    if blk == "0":
        os.environ.pop("CARLE_B1_BLOCKS", None)
    else:
        os.environ["CARLE_B1_BLOCKS"] = blk
    hip.reload_knobs()                  # the library reads its knobs once
    for name, fn in cases.items():
        for i in range(6):
            fn(i)
        reps = []
        for _ in range(int(os.environ.get("KB_REPS", "5"))):      # median and minimum of several blocks of 200 launches
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for i in range(200):
                fn(i)
            e1.record()
            torch.cuda.synchronize()
            reps.append(e0.elapsed_time(e1) * 5)
        reps.sort()
        us, us_min = reps[len(reps) // 2], reps[0]
        print(f"N={n} {size}x{size} gridmap={gmap:>5} blocks={blk:>6} {name:13s}: {us:7.2f} us (min {us_min:6.2f})  {alg[name] / us / 1e6:5.2f} TB/s  frac {alg[name] / us / 8e6:.3f}", flush=True)

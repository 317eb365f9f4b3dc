#!/usr/bin/env python3
"""Root-causing the two timing levels of ca_step_b1_kernel at BASELINE configs[4] (N=4096, 512x512 bit-packed, B36/S23,
64x64 window; 2 x 128 MiB of state = exactly the 256 MiB Infinity Cache).  VERDICT r4 item 3.

In ONE process: P independent state buffers (separate hipMalloc segments: different physical pages) and a packed action
ring; every experiment times `reps` back-to-back one-launch steps (carle_env_step_b1) that ping-pong between two of the
buffers, and prints the buffers' device addresses (mod 4 KiB / 2 MiB / 256 MiB) beside the time.  Experiments:
  pairs     every (in, out) pair of distinct buffers, twice           -> does the level belong to the PAIR (placement)?
  offsets   one 1.5 GiB arena: in at 0, out at 128 MiB + delta       -> does a relative offset inside one segment matter?
  ring      the action ring re-allocated / shared record             -> does the action ring's placement matter?
  third     three buffers rotating (a -> b -> c -> a)                -> is it the re-read of step t's output at step t + 1?
The experiment sequence is printed as `SEQ <index> <label> <reps>` lines so that a rocprofv3 --pmc pass of the same
command (FETCH_SIZE / WRITE_SIZE / TCC_* per dispatch, in dispatch order) can be folded per experiment (tools/b1_levels_pmc.py).

    python tools/b1_levels.py [--n 4096] [--size 512] [--buffers 6] [--reps 200] [--what pairs,offsets,ring,third]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd import hip                      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--buffers", type=int, default=6)
ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--what", default="pairs,offsets,ring,third")
ap.add_argument("--no-window", action="store_true")
args = ap.parse_args()
n, size, reps = args.n, args.size, args.reps
rule = (1 << 12) | (1 << 15) | 0xC              # B36/S23
words = n * size * size // 64
nbytes = words * 8
g = torch.Generator(device="cuda").manual_seed(1)
alg = n * (size * size // 4 + (0 if args.no_window else 512))
seq = [0]


def addr(t):
    p = t.data_ptr()
    return f"0x{p:012x} (4K {p % 4096:4d}, 2M {(p % (2 << 20)) >> 12:3d} pg, 256M {(p % (256 << 20)) >> 20:3d} MiB)"


def fresh_state():
    t = torch.empty(words, dtype=torch.int64, device="cuda")
    t.random_(generator=g)
    return t.view(n, 1, size, size // 64)


def fresh_ring(k=6):
    return [hip.pack_f32_b1((torch.rand(n, 1, 64, 64, device="cuda", generator=g) <= 0.1).float()) for _ in range(k)]


st = hip.new_status(torch.device("cuda", 0))
reward = torch.empty(n, 1, device="cuda")


def timed(label, bufs, ring):
    """`reps` steps cycling through `bufs` (2 = ping-pong, 3 = rotation), action ring `ring` (None: no action)."""
    k = len(bufs)

    def call(i):
        hip.env_step_b1(bufs[i % k], None if ring is None else ring[i % len(ring)], bufs[(i + 1) % k], reward, rule, st, i & 1)
    for i in range(2 * k):
        call(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(reps):
        call(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"SEQ {seq[0]} {label} {reps + 2 * k}", flush=True)
    seq[0] += 1
    print(f"    {label:44s} {us:7.2f} us  frac {alg / us / 8e6:.3f}", flush=True)
    return us


what = set(args.what.split(","))
dummy = None
if "dummyfirst" in what:
    dummy = fresh_state()                        # the FIRST 128 MiB allocation of the process is not part of any pair
    print("dummy (first large allocation):", addr(dummy))
ring = None if args.no_window else fresh_ring()
if ring is not None:
    print("action ring:", "; ".join(addr(r) for r in ring[:3]), "...")

if "pairs" in what:
    bufs = [fresh_state() for _ in range(args.buffers)]
    for i, b in enumerate(bufs):
        print(f"buffer {i}: {addr(b)}")
    res = {}
    for rnd in range(2):
        for i in range(len(bufs)):
            for j in range(i + 1, len(bufs)):
                res.setdefault((i, j), []).append(timed(f"pair ({i},{j}) round {rnd}", [bufs[i], bufs[j]], ring))
    spread = [abs(a - b) for a, b in res.values()]
    lo, hi = min(min(v) for v in res.values()), max(max(v) for v in res.values())
    print(f"pairs: fastest {lo:.2f} us, slowest {hi:.2f} us; the same pair repeats within {max(spread):.2f} us "
          f"(median {sorted(spread)[len(spread) // 2]:.2f})")
    # per buffer: mean over the pairs it takes part in -- is a BUFFER slow, or a combination?
    for i in range(len(bufs)):
        mine = [sum(v) / 2 for (a, b), v in res.items() if i in (a, b)]
        print(f"    buffer {i}: mean over its pairs {sum(mine) / len(mine):.2f} us (min {min(mine):.2f}, max {max(mine):.2f})")
    del bufs
    torch.cuda.empty_cache()

if "direction" in what:
    # one direction only (the same in / out every launch: contents do not matter for the time): is it READING a buffer
    # or WRITING it that is fast?
    bufs = ([dummy] if dummy is not None else []) + [fresh_state() for _ in range(4)]
    for i, b in enumerate(bufs):
        print(f"buffer {i}: {addr(b)}")

    def one_way(label, src, dst):
        def call(i):
            hip.env_step_b1(src, None if ring is None else ring[i % len(ring)], dst, reward, rule, st, i & 1)
        for i in range(4):
            call(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(reps):
            call(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        print(f"SEQ {seq[0]} {label} {reps + 4}", flush=True)
        seq[0] += 1
        print(f"    {label:44s} {us:7.2f} us  frac {alg / us / 8e6:.3f}", flush=True)
    for i in range(len(bufs)):
        for j in range(len(bufs)):
            if i != j:
                one_way(f"read {i} -> write {j}", bufs[i], bufs[j])
    del bufs
    torch.cuda.empty_cache()

if "realloc" in what:
    # everything released to the driver and allocated again, three times: does "the first one" stay the fast one?
    for rnd in range(3):
        torch.cuda.empty_cache()
        bufs = [fresh_state() for _ in range(3)]
        print(f"round {rnd}: " + " | ".join(addr(b) for b in bufs))
        for i, j in ((0, 1), (1, 2), (0, 2)):
            timed(f"realloc round {rnd} pair ({i},{j})", [bufs[i], bufs[j]], ring)
        del bufs

if "offsets" in what:
    arena = torch.empty((1536 << 20) // 8, dtype=torch.int64, device="cuda")
    arena.random_(generator=g)
    print("arena:", addr(arena))
    a = arena[:words].view(n, 1, size, size // 64)
    for delta in (0, 4 << 10, 64 << 10, 1 << 20, 2 << 20, 32 << 20, 64 << 20, 128 << 20, 256 << 20, 512 << 20):
        off = (nbytes + delta) // 8
        b = arena[off:off + words].view(n, 1, size, size // 64)
        timed(f"one arena, out = in + 128 MiB + {delta >> 10} KiB", [a, b], ring)
    del arena, a, b
    torch.cuda.empty_cache()

if "ring" in what and ring is not None:
    a, b = fresh_state(), fresh_state()
    print("in/out:", addr(a), "|", addr(b))
    timed("ring of 6 (as allocated first)", [a, b], ring)
    timed("no action at all", [a, b], None)
    for k in range(3):
        r2 = fresh_ring()
        print("   new ring:", addr(r2[0]))
        timed(f"fresh ring #{k}", [a, b], r2)
    timed("ring of 1 (one record, L2-hot)", [a, b], ring[:1])
    del a, b
    torch.cuda.empty_cache()

if "third" in what:
    a, b, c = fresh_state(), fresh_state(), fresh_state()
    print("a/b/c:", addr(a), "|", addr(b), "|", addr(c))
    timed("ping-pong a <-> b", [a, b], ring)
    timed("ping-pong a <-> c", [a, c], ring)
    timed("ping-pong b <-> c", [b, c], ring)
    timed("rotation a -> b -> c (output re-read two steps later)", [a, b, c], ring)
    timed("ping-pong a <-> b again", [a, b], ring)

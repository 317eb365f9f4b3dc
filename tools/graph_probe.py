#!/usr/bin/env python3
"""Feasibility probe: torch.cuda.CUDAGraph capture of CARLE.step (+ reward wrappers) launched through ctypes."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd.env import CARLE
from carle_amd.mcl import RND2D, AE2D

def probe(n, size, wrappers):
    torch.manual_seed(0)
    def make():
        torch.manual_seed(1)
        env = CARLE(instances=n, width=size, height=size, device="cuda:0")
        if wrappers:
            env = AE2D(RND2D(env)); env.eval()
        return env
    acts = [(torch.rand(n, 1, 64, 64, device="cuda") < 0.1).float() for _ in range(8)]
    ref = make(); ref.reset(); outs = []
    for a in acts:
        o, r, d, i = ref.step(a); outs.append((o.clone(), r.clone()))
    env = make(); env.reset()
    static_a = torch.zeros_like(acts[0])
    # warm-up on a side stream
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        env.step(static_a); env.step(static_a)
    torch.cuda.current_stream().wait_stream(s)
    env.reset()
    graphs = []
    for p in range(2):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            res = env.step(static_a)
        graphs.append((g, res))
    env.reset()   # NOTE: allocates a new universe: graph 0 reads the captured one
    print("captured", n, size, wrappers, flush=True)

for cfg in ((3, 64, False), (16, 256, True)):
    try:
        probe(*cfg)
    except Exception as e:
        import traceback; traceback.print_exc()

"""Where do the ~3 us/step between back-to-back library calls and CARLE.step go? (development tool)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd import hip
from carle_amd.env import CARLE
n = 4096
life = hip.rule_mask([3], [2, 3])
acts = [(torch.rand(n, 1, 64, 64, device="cuda") < 0.1).float() for _ in range(6)]
st = hip.new_status("cuda:0")
reward = torch.empty(n, 1, device="cuda")

def timeit(name, body, k=1000):
    for i in range(20): body(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); t0 = time.perf_counter(); e0.record()
    for i in range(k): body(i)
    t1 = time.perf_counter(); e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / k * 1e3:.2f} us per step (host enqueue {1e6 * (t1 - t0) / k:.1f} us)")

state = {"x": (torch.rand(n, 1, 64, 64, device="cuda") < 0.5).float(), "y": torch.empty(n, 1, 64, 64, device="cuda")}
def pingpong(i):
    hip.env_step_f32(state["x"], acts[i % 6], state["y"], reward, life, st, i & 1, True)
    state["x"], state["y"] = state["y"], state["x"]
def fresh(i):
    out = torch.empty_like(state["x"])
    r = torch.empty(n, 1, device="cuda")
    hip.env_step_f32(state["x"], acts[i % 6], out, r, life, st, i & 1, True)
    state["x"] = out
env = CARLE(width=64, height=64, instances=n, device="cuda:0")
env.reset(); env.universe = state["x"].clone()
def envstep(i):
    env.step(acts[i % 6])
for rep in range(2):
    timeit("direct, ping-pong buffers", pingpong)
    timeit("direct, fresh output tensors", fresh)
    timeit("CARLE.step", envstep)
from carle_amd import dist as cdist
stats = cdist.RunningRewardStats("cuda:0", sync_every=16)
def envstep_sep(i):
    obs, reward, d, info = env.step(acts[i % 6])
    stats.update(reward)
env2 = CARLE(width=64, height=64, instances=n, device="cuda:0")
stats2 = cdist.RunningRewardStats("cuda:0", sync_every=16)
stats2.attach(env2)
env2.reset(); env2.universe = state["x"].clone()
def envstep_merged(i):
    obs, reward, d, info = env2.step(acts[i % 6])
    stats2.after_step(env2, reward)
stats3 = cdist.RunningRewardStats("cuda:0", sync_every=10**9)
env3 = CARLE(width=64, height=64, instances=n, device="cuda:0")
stats3.attach(env3)
env3.reset(); env3.universe = state["x"].clone()
def envstep_merged_noflush(i):
    obs, reward, d, info = env3.step(acts[i % 6])
    stats3.after_step(env3, reward)
for rep in range(2):
    timeit("CARLE.step + separate moments kernel", envstep_sep)
    timeit("CARLE.step + merged moments", envstep_merged)
    timeit("CARLE.step + merged moments, no periodic flush", envstep_merged_noflush)

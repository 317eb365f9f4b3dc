import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd.env import CARLE
from carle_amd.mcl import RND2D
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = CARLE(width=64, height=64, instances=n, device="cuda:0")
w = RND2D(env, async_reward=False, train_predictor=False); w.eval(); w.reset()
env.universe = (torch.rand(n, 1, 64, 64, device="cuda") < 0.5).float()
acts = [(torch.rand(n, 1, 64, 64, device="cuda") < 0.1).float() for _ in range(6)]
for i in range(10): w.step(acts[i % 6])
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for i in range(200): w.step(acts[i % 6])
e1.record(); torch.cuda.synchronize()
print(f"dbg={os.environ.get('CARLE_FUSED_DEBUG','0')} N={n}: {e0.elapsed_time(e1) * 5:.1f} us per step (fused kernel + finish)")

import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd.agents import RandomAgent
from carle_amd.mcl import RND2D, AE2D
from carle_amd.env import CARLE
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
env = AE2D(RND2D(CARLE(width=256, height=256, instances=n, device="cuda:0"), train_predictor=False), train_predictor=False)
env.eval(); obs = env.reset()
agent = RandomAgent(action_width=64, action_height=64)
for i in range(20): env.step(agent(obs))
torch.cuda.synchronize(); t0 = time.time()
for i in range(200): env.step(agent(obs))
t1 = time.time(); torch.cuda.synchronize(); t2 = time.time()
print(f"N={n}: host enqueue {1e3 * (t1 - t0) / 200:.3f} ms/step, total {1e3 * (t2 - t0) / 200:.3f} ms/step")

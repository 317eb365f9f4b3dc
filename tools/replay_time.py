import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd.env import CARLE
from carle_amd.mcl import RND2D, AE2D
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
env = CARLE(width=64, height=64, instances=n, device="cuda:0")
rnd = RND2D(env); ae = AE2D(RND2D(CARLE(width=64, height=64, instances=n, device="cuda:0")))
rnd.reset(); ae.reset()
g = (torch.rand(n, 1, 64, 64, device="cuda") < 0.3).float()
for name, w in (("rnd", rnd), ("ae", ae)):
    for mode in ("train", "eval"):
        getattr(w, mode)()
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.time()
            loss = torch.mean(w._torch_loss(g)); loss.backward()
            torch.cuda.synchronize(); t1 = time.time()
            print(f"{name} {mode} replay {rep}: {1e3 * (t1 - t0):.1f} ms")
from torch.profiler import profile, ProfilerActivity
rnd.train()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    loss = torch.mean(rnd._torch_loss(g)); loss.backward(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=60))

import sys, os, torch
sys.path.insert(0, os.getcwd())
from carle_amd.env import CARLE
from carle_amd.mcl import RND2D, AE2D
from carle_amd import hip
torch.manual_seed(0)
for n in (4096, 16384):
    env = CARLE(width=64, height=64, instances=n, device="cuda:0")
    rnd = RND2D(env); ae = AE2D(RND2D(CARLE(width=64, height=64, instances=n, device="cuda:0")))
    rnd.eval(); rnd.reset(); ae.eval(); ae.reset()
    for dens in (0.5, 0.08):
        g = (torch.rand(n, 1, 64, 64, device="cuda") < dens).float()
        gp = hip.pack_f32_b1(g)
        for name, w, obs in (("rnd f32", rnd, g), ("rnd packed", rnd, gp), ("ae f32", ae, g), ("ae packed", ae, gp)):
            w.assume_binary_obs = True
            for _ in range(5): w.get_bonus_only(obs)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(100): w.get_bonus_only(obs)
            e1.record(); torch.cuda.synchronize()
            print(f"N={n} density {dens}: {name}: {e0.elapsed_time(e1) * 10:.1f} us per call")

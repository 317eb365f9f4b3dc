import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd.env import CARLE
from carle_amd.mcl import RND2D, AE2D
from carle_amd import hip
import itertools
for n, cls in itertools.product((1024, 4096), (RND2D, AE2D)):
    env = CARLE(width=64, height=64, instances=n, device="cuda:0")
    rnd = cls(env); rnd.eval(); rnd.reset(); rnd.assume_binary_obs = True; rnd.batch_size = 10**9
    g = (torch.rand(n, 1, 64, 64, device="cuda") < 0.1).float()
    bonus = torch.empty(n, device="cuda")
    for obs, name in ((g, "f32"), (hip.pack_f32_b1(g), "packed")):
        for _ in range(3): rnd._fused_train(obs, None, bonus, None)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): rnd._fused_train(obs, None, bonus, None)
        e1.record(); torch.cuda.synchronize()
        print(f"{cls.__name__} N={n} {name}: fused fwd+bwd {e0.elapsed_time(e1) * 50:.1f} us per call")

import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carle_amd.env import CARLE
from carle_amd.mcl import RND2D, AE2D
size, n, batch = 256, 16, 3
stacks = []
for paired in (True, False):
    torch.manual_seed(23)
    env = CARLE(width=size, height=size, instances=n, device="cuda:0")
    w = AE2D(RND2D(env))
    w.env.reward_scale, w.reward_scale = 0.7, 1.3
    w.env.batch_size, w.batch_size = batch, batch + 1
    if not paired:
        w._can_pair_step = lambda: 0
    w.eval(); w.reset()
    stacks.append((env, w))
g = torch.Generator(device="cuda").manual_seed(10)
u0 = (torch.rand(n, 1, size, size, device="cuda", generator=g) < 0.3).float()
for env, w in stacks:
    env.universe = u0.clone()
a = (torch.rand(n, 1, 64, 64, device="cuda", generator=g) < 0.1).float()
(o0, r0, _, _), (o1, r1, _, _) = [w.step(a) for env, w in stacks]
print("rnd bonus equal", torch.equal(stacks[0][1].env._bonus_buf, stacks[1][1].env._bonus_buf), (stacks[0][1].env._bonus_buf - stacks[1][1].env._bonus_buf).abs().max().item())
print("ae bonus equal", torch.equal(stacks[0][1]._bonus_buf, stacks[1][1]._bonus_buf))
print("reward equal", torch.equal(r0, r1), (r0 - r1).abs().max().item())
p0, p1 = stacks[0][1].env._partial, stacks[1][1].env._partial
print("pred/targ equal", torch.equal(p0[:2 * n * 16], p1[:2 * n * 16]), (p0[:2*n*16]-p1[:2*n*16]).abs().max().item())
rb0, rb1 = stacks[0][1].env._bonus_buf, stacks[1][1].env._bonus_buf
r_rnd0 = 0.7 * rb0
print("reward - 0.7*rnd - 1.3*ae:", (r0.flatten() - (0.7 * rb0 + 1.3 * stacks[0][1]._bonus_buf)).abs().max().item(), (r1.flatten() - (0.7 * rb1 + 1.3 * stacks[1][1]._bonus_buf)).abs().max().item())

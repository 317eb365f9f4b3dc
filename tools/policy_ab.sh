python tools/kbench_64.py 4096 2>&1 | grep -v amdgpu.ids | grep "f32 state" 
python tools/kbench_rnd.py 4096 0 2>&1 | grep -v amdgpu | tail -1
python tools/kbench_rnd.py 16384 0 2>&1 | grep -v amdgpu | tail -1
python - <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from carle_amd import hip
rule = (1 << 12) | 0xC
g = torch.Generator(device="cuda").manual_seed(1)
for fmt, n, size in (("f32", 256, 256), ("f32", 1024, 256), ("f32", 256, 512), ("u8", 1024, 256), ("u8", 4096, 256)):
    dt = torch.float32 if fmt == "f32" else torch.uint8
    a = (torch.rand(n, 1, size, size, device="cuda", generator=g) < 0.5).to(dt); b = torch.empty_like(a)
    ring = [(torch.rand(n, 1, 64, 64, device="cuda", generator=g) <= 0.1).to(dt) for _ in range(6)]
    st = hip.new_status(a.device); bufs = [a, b]
    step = hip.ca_step_f32 if fmt == "f32" else hip.ca_step_u8
    fn = lambda i: step(bufs[i & 1], ring[i % 6], bufs[(i & 1) ^ 1], rule, st, 0)
    for i in range(4): fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(100): fn(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 10
    alg = n * (size * size * 2 + 4096) * a.element_size()
    print(f"wide {fmt} N={n} {size}x{size} ({n*size*size*a.element_size()>>20} MiB/array) act: {us:8.2f} us  frac {alg/us/8e6:.3f}")
PY

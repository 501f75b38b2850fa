"""control experiment: are plain PyTorch kernels bit-reproducible while a second process loads the GPU?  python dbg_torch_race.py detect|load N"""
import sys, time, hashlib
import torch
mode, n = sys.argv[1], int(sys.argv[2])
g = torch.Generator(device='cuda').manual_seed(1)
x = torch.randn(1024, 128, device='cuda', generator=g); w = torch.randn(128, 4, device='cuda', generator=g); e = torch.randn(4, 128, device='cuda', generator=g)
def f():
    p = torch.softmax(x @ w, dim=1)
    o = (x + p @ e) * 0.5
    y = torch.nn.functional.layer_norm(o, (128,))
    return torch.cat([o, y], 1)
if mode == 'load':
    t0 = time.time()
    while time.time() - t0 < n:
        f()
    torch.cuda.synchronize(); sys.exit(0)
ref = hashlib.md5(f().cpu().numpy().tobytes()).hexdigest()
bad = 0
for it in range(n):
    h = hashlib.md5(f().cpu().numpy().tobytes()).hexdigest()
    bad += h != ref
print('torch control: %d of %d differ' % (bad, n))

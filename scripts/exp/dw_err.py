import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hual_amd import lib
dev = torch.device('cuda:0')
for (M, K) in [(9472, 128), (4096, 1024), (1000, 400), (9472, 256)]:
    for mode in ('randn', 'grad_up', 'grad_down', 'grad_spike', 'tiny'):
        g = torch.Generator().manual_seed(M + K)
        A = torch.randn(M, K, generator=g)
        dY = torch.randn(M, 128, generator=g)
        r = torch.arange(M).float() / M
        if mode == 'grad_up': dY = dY * (1e-7 * 10 ** (5 * r))[:, None]          # 1e-7 .. 1e-2 growing along the rows: the scale is lowered again and again
        if mode == 'grad_down': dY = dY * (1e-2 * 10 ** (-5 * r))[:, None]
        if mode == 'grad_spike': dY = dY * 1e-5; dY[M // 2] *= 1e4
        if mode == 'tiny': dY = dY * 1e-12
        A, dY = A.to(dev), dY.to(dev)
        dW = torch.zeros(K, 128, device=dev); db = torch.zeros(128, device=dev)
        lib.linear_dw(A, dY, dW, db, workgroups=0)
        ref = A.double().t() @ dY.double()
        f32 = (A.t() @ dY).double()
        sc = ref.abs().max().item()
        print('M%-5d K%-4d %-10s hip %.2e   torch f32 %.2e   (of max %.2e)  finite %s' % (M, K, mode, (dW.double() - ref).abs().max().item() / sc, (f32 - ref).abs().max().item() / sc, sc, bool(torch.isfinite(dW).all())))

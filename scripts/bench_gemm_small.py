#!/usr/bin/env python3
"""GPU box: warm the clocks, then launch the plain dense kernel at M = 8192 / 9472 / 65536 (for PMC runs)."""
import sys
import torch
sys.path.insert(0, '.')
from hual_amd import lib
dev = torch.device('cuda:0')
K = N = 128
W = torch.randn(K, N, device=dev) / K ** 0.5
b = torch.randn(N, device=dev)
A0 = torch.randn(65536, K, device=dev); Y0 = torch.empty(65536, N, device=dev)
for _ in range(300):
    lib.linear_fwd(A0, W, b, act=1, out=Y0)      # ~15 ms of warm-up work
torch.cuda.synchronize()
for M in (8192, 9472, 65536):
    A = A0[:M]; Y = Y0[:M]
    for _ in range(30):
        lib.linear_fwd(A, W, b, act=1, out=Y)
    torch.cuda.synchronize()

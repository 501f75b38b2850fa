#!/usr/bin/env python3
"""GPU box: launch the plain dense kernel at a ladder of M (run under rocprofv3 --kernel-trace to read durations)."""
import sys
import torch
sys.path.insert(0, '.')
from hual_amd import lib
dev = torch.device('cuda:0')
K = N = 128
W = torch.randn(K, N, device=dev) / K ** 0.5
b = torch.randn(N, device=dev)
for M in (1024, 2048, 4096, 8192, 9472, 12288, 16384, 32768, 65536):
    A = torch.randn(M, K, device=dev)
    Y = torch.empty(M, N, device=dev)
    for _ in range(20):
        lib.linear_fwd(A, W, b, act=1, out=Y)
    torch.cuda.synchronize()
# dW at the same ladder
for M in (1024, 4096, 9472, 32768):
    A = torch.randn(M, K, device=dev); dY = torch.randn(M, N, device=dev)
    dW = torch.zeros(K, N, device=dev); db = torch.zeros(N, device=dev)
    for rpb in (256, 1024):
        for _ in range(10):
            lib.linear_dw(A, dY, dW, db, rows_per_block=rpb)
    torch.cuda.synchronize()

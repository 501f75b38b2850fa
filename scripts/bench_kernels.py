#!/usr/bin/env python3
"""Micro-benchmarks of individual kernels (GPU box).  Prints one line per case; used while tuning."""
import json
import sys
import torch

sys.path.insert(0, '.')
from hual_amd import lib  # noqa: E402


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3   # us


def main():
    dev = torch.device('cuda:0')
    out = []
    for (M, K, N) in [(9472, 128, 128), (9472, 128, 384), (8192, 1024, 128), (9472, 512, 128), (1280, 400, 128)]:
        A = torch.randn(M, K, device=dev)
        W = torch.randn(K, N, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        Y = torch.empty(M, N, device=dev)
        us = timeit(lambda: lib.linear_fwd(A, W, b, act=1, out=Y))
        fl = 2.0 * M * K * N
        out.append(dict(kernel='gemm_fwd', M=M, K=K, N=N, us=round(us, 2), tflops=round(fl / us / 1e6, 2)))
        Wt = torch.randn(N, K, device=dev)
        us = timeit(lambda: lib.linear_fwd(A, Wt.t().contiguous().t() if False else Wt[:, :K], None, trans_w=True) if N == K else None) if N == K else None
        if us:
            out.append(dict(kernel='gemm_dx', M=M, K=K, N=N, us=round(us, 2), tflops=round(fl / us / 1e6, 2)))
        dY = torch.randn(M, N, device=dev)
        dW = torch.zeros(K, N, device=dev)
        db = torch.zeros(N, device=dev)
        for rpw in (32, 128, 512):
            us = timeit(lambda: lib.linear_dw(A, dY, dW, db, rows_per_wave=rpw))
            out.append(dict(kernel='gemm_dw', M=M, K=K, N=N, rpw=rpw, us=round(us, 2), tflops=round(fl / us / 1e6, 2)))
        ref = timeit(lambda: torch.addmm(b, A, W))
        out.append(dict(kernel='torch_addmm(rocblas)', M=M, K=K, N=N, us=round(ref, 2), tflops=round(fl / ref / 1e6, 2)))
    for o in out:
        print(json.dumps(o))


if __name__ == '__main__':
    main()

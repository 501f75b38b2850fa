"""Attention-core microbenchmark on the GPU box: hual_attention_fwd_save / hual_attention_bwd per job shape of the bench
workload (B=64, 8 heads of 16: video self 128x128, video->query cross 128x20, query self 20x20, query->video 20x128),
dropout 0.2, HIP-event timing over --iters launches.

    python scripts/bench_attn.py [--iters 50] [--B 64] [--T 128] [--L 20]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hual_amd import lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=50)
    ap.add_argument('--B', type=int, default=64)
    ap.add_argument('--T', type=int, default=128)
    ap.add_argument('--L', type=int, default=20)
    ap.add_argument('--rate', type=float, default=0.2)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    l = lib.load()
    B = a.B
    rng_state = torch.tensor(np.array([1, 2, 3], dtype=np.uint32).view(np.int32)).to(dev)
    tot_f = tot_b = 0.0
    for name, Tq, Tk in (('v-self', a.T, a.T), ('v->q', a.T, a.L), ('q-self', a.L, a.L), ('q->v', a.L, a.T)):
        g = torch.Generator().manual_seed(1)
        Q = torch.randn(B * Tq, 384, generator=g).to(dev)
        KV = torch.randn(B * Tk, 384, generator=g).to(dev)
        dO = torch.randn(B * Tq, 128, generator=g).to(dev)
        qm = (torch.arange(Tq)[None, :] < torch.randint(Tq // 2 + 1, Tq + 1, (B, 1), generator=g)).float().reshape(-1).to(dev)
        km = (torch.arange(Tk)[None, :] < torch.randint(Tk // 2 + 1, Tk + 1, (B, 1), generator=g)).float().reshape(-1).to(dev)
        ldm = l.hual_attention_keep_row_bytes(Tk)
        keep = torch.zeros(B * Tq * 8, ldm, dtype=torch.uint8, device=dev)
        stats = torch.zeros(2, B * Tq * 8, device=dev)
        O = torch.empty(B * Tq, 128, device=dev)
        dQ = torch.empty(B * Tq, 128, device=dev)
        dKV = torch.empty(B * Tk, 384, device=dev)
        K, V = KV[:, 128:], KV[:, 256:]

        def fwd():
            lib.check(l.hual_attention_fwd_save(lib.ptr(Q), 384, lib.ptr(K), lib.ptr(V), 384, lib.ptr(O), 128, B, Tq, Tk, lib.ptr(qm),
                                                lib.ptr(km), lib.ptr(stats), lib.ptr(keep), ldm, lib.ptr(rng_state), a.rate, 9,
                                                lib.stream_ptr()))

        def bwd():
            lib.check(l.hual_attention_bwd(lib.ptr(Q), 384, lib.ptr(K), lib.ptr(V), 384, lib.ptr(O), 128, lib.ptr(stats), lib.ptr(keep),
                                           ldm, lib.ptr(dO), 128, lib.ptr(dQ), 128, lib.ptr(dKV[:, 128:]), lib.ptr(dKV[:, 256:]), 384, B, Tq,
                                           Tk, lib.ptr(qm), lib.ptr(km), lib.ptr(rng_state), a.rate, 9, lib.stream_ptr()))
        res = []
        for fn in (fwd, bwd):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            # a captured graph of `iters` launches: the ctypes call (~10 us) would otherwise bound the short kernels
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(a.iters):
                    fn()
            gr.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                gr.replay()
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) * 1e3 / (4 * a.iters))
        flops = 4.0 * B * 8 * Tq * Tk * 16
        print('%-7s Tq=%3d Tk=%3d  fwd %7.2f us (%6.1f TF)   bwd %7.2f us (%6.1f TF)' %
              (name, Tq, Tk, res[0], flops / res[0] / 1e6, res[1], 2.5 * flops / res[1] / 1e6))
        tot_f += res[0]
        tot_b += res[1]
    print('sum of the four jobs: fwd %.1f us, bwd %.1f us' % (tot_f, tot_b))


if __name__ == '__main__':
    main()

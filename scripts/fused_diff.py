#!/usr/bin/env python3
"""GPU box: per-tap differences between the fused conv_block kernels (HUAL_FUSE_CB=1) and the launch sequences they replace."""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import parity_util as pu

names = []
for t in ('fe1', 'fe0', 'cb'):
    for i in (3, 2, 1, 0):
        names += ['d.%s.z%d' % (t, i)]
    names += ['d.%s.x0' % t]

def run(case, drop, fuse):
    os.environ['HUAL_FUSE_CB'] = fuse
    cfg, p, wv, b, labels = case
    m = pu.hip_model(cfg, p, wv); m.set_rng(5, 7)
    m.forward(b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy(), drop_rate=drop, labels=tuple(x.numpy() for x in labels))
    m.backward(); torch.cuda.synchronize()
    return {n: m.tap(n).clone() for n in names}, m.grads.clone()

case = pu.make_case(B=3, T=37, L=9, C=4, seed=11, max_vlen=40)
for rep in range(2):
    a, ga = run(case, 0.0, '1'); b, gb = run(case, 0.0, '0')
    a2, ga2 = run(case, 0.0, '0')
    for n in names:
        d = (a[n] - b[n]).abs(); d2 = (a2[n] - b[n]).abs()
        nz = (d > 0).nonzero()
        print('%-12s fused-unfused %.3e (rows differing: %s)   unfused-unfused %.3e  scale %.3e' % (n, float(d.max()), sorted(set(nz[:, 0].tolist()))[:12], float(d2.max()), float(b[n].abs().max())))
    print('grads: fused-unfused %.3e unfused-unfused %.3e' % (float((ga - gb).abs().max()), float((ga2 - gb).abs().max())))

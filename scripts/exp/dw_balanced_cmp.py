"""GPU box: gradients of one eager step with the balanced weight-gradient launch vs the fixed row split, per tensor."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import parity_util as pu
from hual_amd.train import Trainer
B, T, L = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (4, 24, 7))]
cfg, p, wv, b, labels = pu.make_case(B=B, T=T, L=L, C=5, seed=21, max_vlen=max(T, 24))


def run(bal):
    os.environ['HUAL_DW_BALANCED'] = str(bal)
    m = pu.hip_model(cfg, p, wv)
    m.set_rng(3, 5)
    tr = Trainer(m, world=1, use_graph=False)
    tr.set_batch(b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy(), *[x.numpy() for x in labels])
    tr.step(lr=1e-3, drop_rate=0.2)
    torch.cuda.synchronize()
    return m, m.grads.cpu().numpy().copy()


m1, g1 = run(1)
m0, g0 = run(0)
for e in sorted(m0.table.entries, key=lambda e: e['offset']):
    o, sz = e['offset'], e['size']
    d = np.abs(g1[o:o + sz] - g0[o:o + sz]).max()
    r = np.abs(g0[o:o + sz]).max()
    if d > 1e-4 * r + 1e-9:
        print('%-70s diff %.3e ref %.3e' % (e['name'], d, r))
print('done')

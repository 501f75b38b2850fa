import sys
import numpy as np
import torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import parity_util as pu
from hual_amd.train import Trainer
lr, drop, seed, off = 1e-3, 0.2, 99, 5
cfg, p, wv, b, labels = pu.well_conditioned_case(drop_rate=drop, rng_seed=seed, rng_offset=off, B=4, T=24, L=7, C=5, seed=21)


def run(graph):
    m = pu.hip_model(cfg, p, wv)
    m.set_rng(seed, off)
    tr = Trainer(m, world=1, use_graph=graph)
    tr.set_batch(b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy(), *[x.numpy() for x in labels])
    gs, ps = [], []
    for s in range(3):
        tr.step(lr=lr, drop_rate=drop)
        torch.cuda.synchronize()
        gs.append(m.grads.cpu().numpy().copy())
        ps.append(m.params.cpu().numpy().copy())
    return m, gs, ps


mg, gg, pg = run(True)
me, ge, pe = run(False)
for s in range(3):
    d = np.abs(gg[s] - ge[s])
    bad = np.argwhere(d > 1e-3 * (np.abs(ge[s]) + 1e-3)).ravel()
    print('step %d: grads differ at %d of %d elements; params max diff %.3e' % (s, len(bad), d.size, np.abs(pg[s] - pe[s]).max()))
    if len(bad):
        print('   first bad idx', bad[:5], 'last', bad[-5:], 'graph vals', gg[s][bad[:5]], 'eager vals', ge[s][bad[:5]])
        for e in sorted(mg.table.entries, key=lambda e: e['offset']):
            o, sz = e['offset'], e['size']
            k = int(((bad >= o) & (bad < o + sz)).sum())
            if k:
                seg = slice(o, o + sz)
                print('     %-66s %7d / %7d  max|g| graph %.3e eager %.3e' % (e['name'], k, sz, np.abs(gg[s][seg]).max(), np.abs(ge[s][seg]).max()))

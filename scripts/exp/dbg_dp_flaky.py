"""debug: single path vs forced data-parallel path on one rank, two steps; where do the step-2 gradients differ?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import parity_util as pu
from hual_amd.train import Trainer
cfg, p, wv, b, labels = pu.make_case(B=4, T=24, L=7, C=5, seed=21)
feeds = (b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy(), *[x.numpy() for x in labels])
junk = [torch.randn(1 << 24, device='cuda') * 1e3 for _ in range(4)]      # dirty the allocator's memory
del junk
for trial in range(int(os.environ.get('TRIALS', '12'))):
    out = []
    for force_dp in (False, True):
        m = pu.hip_model(cfg, p, wv); m.set_rng(7, 3)
        tr = Trainer(m, world=1, use_graph=False, force_dp=force_dp)
        tr.set_batch(*feeds)
        gs = []
        for _ in range(2):
            tr.step(lr=1e-4, drop_rate=0.2)
            torch.cuda.synchronize()
            gs.append(m.table.unpack(m.grads.detach().cpu().numpy().copy()))
        out.append(gs)
    for s in range(2):
        worst = max(((float(np.abs(out[0][s][k] - out[1][s][k]).max()), k) for k in out[0][s]), key=lambda t: t[0])
        gmax = max(float(np.abs(v).max()) for v in out[0][s].values())
        print('trial %d step %d: worst |dg| %.3e at %s (max |g| %.2f)' % (trial, s, worst[0], worst[1], gmax), flush=True)

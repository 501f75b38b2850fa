"""prints, per step of a free-running HIP trajectory and a free-running oracle trajectory on the same batch, the losses,
the largest logit difference and the largest parameter difference (how fast two fp32 implementations drift apart)"""
import collections, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import parity_util as pu
from oracle import seqpan_ref as R
from hual_amd.train import Trainer
lr = float(os.environ.get('LR', 1e-3)); drop = float(os.environ.get('DROP', 0.2)); steps = int(os.environ.get('STEPS', 30))
seed, off = 31, 11
cfg, p, wv, b, labels = pu.make_case(B=4, T=24, L=7, C=5, seed=21)
m = pu.hip_model(cfg, p, wv); m.set_rng(seed, off)
tr = Trainer(m, world=1, use_graph=True)
tr.set_batch(b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy(), *[x.numpy() for x in labels])
rp = collections.OrderedDict((k, v.clone()) for k, v in p.items())
rm = {k: torch.zeros_like(v) for k, v in p.items()}; rv = {k: torch.zeros_like(v) for k, v in p.items()}
batch = (b['video'], b['lens'], b['word_ids'], b['char_ids'])
for s in range(steps):
    tr.step(lr=lr, drop_rate=drop); torch.cuda.synchronize()
    rp, rm, rv, info = R.train_step(rp, rm, rv, cfg, wv, batch, labels, lr, drop, seed=seed, offset=off + s)
    got = m.state_dict()
    worst = max((float(np.abs(got[k] - v.numpy()).max()), k) for k, v in rp.items())
    dl = float((tr.start_logits.cpu() - info['start_logits']).abs().max())
    print(s, '%.5f %.5f' % (float(tr.last_loss()), float(info['loss'])), 'dlogit %.3e' % dl, 'dparam %.3e %s' % worst,
          tr.start_index.cpu().tolist(), info['start_index'].tolist(), tr.end_index.cpu().tolist(), info['end_index'].tolist())

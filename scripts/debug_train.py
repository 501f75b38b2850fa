import sys, collections
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import parity_util as pu
from oracle import seqpan_ref as R
from hual_amd.train import Trainer
cfg, p, wv, b, labels = pu.make_case(B=4, T=24, L=7, C=5, seed=21)
lr, drop, seed, off = 1e-3, 0.2, 99, 5
pp = collections.OrderedDict((k, v.clone()) for k, v in p.items())
m_ = {k: torch.zeros_like(v) for k, v in p.items()}; v_ = {k: torch.zeros_like(t) for k, t in p.items()}
batch = (b['video'], b['lens'], b['word_ids'], b['char_ids'])
p1, m1, v1, info = R.train_step(pp, m_, v_, cfg, wv, batch, labels, lr, drop, seed=seed, offset=off)
m = pu.hip_model(cfg, p, wv); m.set_rng(seed, off)
tr = Trainer(m, world=1, use_graph=False)
tr.set_batch(b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy(), *[x.numpy() for x in labels])
tr.step(lr=lr, drop_rate=drop); torch.cuda.synchronize()
print('loss', float(tr.last_loss()), float(info['loss']), 'gnorm', float(info['grad_norm']), float(m.sqnorm.sqrt()))
got = m.state_dict(); gg = m.grads_dict()
rows = []
for k in p1:
    d = np.abs(got[k] - p1[k].numpy()); gd = np.abs(gg[k] - info['grads'][k].numpy() * float(info['grad_norm']) / 1.0) if False else None
    rows.append((float(d.max()), k, float(np.abs(p1[k].numpy() - p[k].numpy()).max())))
rows.sort(reverse=True)
for r in rows[:15]: print('%.3e %-80s moved %.3e' % (r[0], r[1], r[2]))
print('---- clipped gradient comparison')
sc = 1.0 / max(float(m.sqnorm.sqrt()), 1.0)
rows = []
for k in p1:
    a = gg[k] * sc; r = info['grads'][k].numpy()
    d = np.abs(a - r)
    i = np.unravel_index(np.argmax(d), d.shape)
    rows.append((float(d.max()), k, float(np.abs(r).max()), float(a[i]), float(r[i])))
rows.sort(reverse=True)
for r in rows[:12]: print('%.3e %-70s refmax %.3e hip %.4e ref %.4e' % r)
k = 'predictor/start_hidden/kernel'
d = np.abs(got[k] - p1[k].numpy()); i = np.unravel_index(np.argmax(d), d.shape)
print('worst elem', i, 'p0', float(p[k].numpy()[i]), 'hip', float(got[k][i]), 'ref', float(p1[k].numpy()[i]), 'g_hip_c', float(gg[k][i]*sc), 'g_ref_c', float(info['grads'][k].numpy()[i]))

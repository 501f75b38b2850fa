"""margins of tests/test_gpu_epoch_loop.py::test_epoch_loop_matches_eager_steps_on_fresh_feeds over repeated runs: share of equal spans
(bar 0.8) and relative difference of the last loss (bar 2e-2) between the eager reference and the graph loop"""
import sys, os
import numpy as np, torch
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, 'tests'))
import test_gpu_epoch_loop as T
from hual_amd.model import SeqPAN
from hual_amd.train import Trainer
cfg, wv, ds = T._setup()
N, bs, lr, drop, epochs = len(ds), 16, 1e-4, 0.2, 3
orders = T._orders(N, epochs)
shares, dl = [], []
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    m0 = SeqPAN(cfg, wv); t0 = Trainer(m0, world=1, use_graph=False)
    ref = []
    for order in orders:
        for lo in range(0, N, bs):
            t0.set_batch_device(ds.assemble(order[lo:lo + bs], out=None, min_chars=4)); t0.step(lr=lr, drop_rate=drop)
            ref.append((t0.start_index.cpu().numpy().copy(), t0.end_index.cpu().numpy().copy()))
    l0 = float(t0.last_loss())
    m1 = SeqPAN(cfg, wv); t1 = Trainer(m1, world=1, use_graph=True)
    got = [t1.run_epoch(ds, order, bs, lr=lr, drop_rate=drop, min_chars=4) for order in orders]
    k = same = total = 0
    for (s, e), order in zip(got, orders):
        for lo in range(0, N, bs):
            n = len(order[lo:lo + bs]); same += int(((s[lo:lo + n] == ref[k][0]) & (e[lo:lo + n] == ref[k][1])).sum()); total += n; k += 1
    shares.append(same / total); dl.append(abs(float(t1.last_loss()) - l0) / abs(l0))
print('equal spans: min %.3f median %.3f (bar 0.8); last-loss relative difference: max %.2e median %.2e (bar 2e-2)' % (min(shares), np.median(shares), max(dl), np.median(dl)))

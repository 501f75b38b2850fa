"""where the HOST time of a replayed epoch-loop step goes (cProfile over one epoch of the ActivityNet length distribution at batch 16)"""
import sys, os, cProfile, pstats, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import al_synth
from hual_amd import al, lib
from hual_amd.dataset import DeviceDataset
from hual_amd.model import SeqPAN
from hual_amd.train import Trainer
N, bs = 4096, 16
recs, vis, data_gt, _ = al_synth.make_trainset_from_lengths('anet', N, 1024, 100, seed=3)
cfg = lib.make_cfg(vdim=1024, max_vlen=100, num_words=1000, num_chars=40)
wv = np.random.default_rng(1).normal(0, 0.4, size=(998, 300)).astype(np.float32)
ds = DeviceDataset(recs, vis)
s0, e0 = al.labels_from_times(data_gt, ds.vlen_h)
ds.set_labels(s0, e0)
m = SeqPAN(cfg, wv)
tr = Trainer(m, world=1, use_graph=True)
rng = np.random.default_rng(0)
orders = [rng.permutation(N).astype(np.int32) for _ in range(2)]
for ep in range(5):
    for o in orders: tr.run_epoch(ds, o, bs, lr=1e-4, drop_rate=0.2)
torch.cuda.synchronize()
h0 = tr.stats['host_enqueue_s']
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
tr.run_epoch(ds, orders[0], bs, lr=1e-4, drop_rate=0.2, want_spans=False)
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('steps %d: host enqueue %.3f ms/step (under the profiler), until the device is done %.3f ms/step; stats %s' % (N // bs, (tr.stats['host_enqueue_s'] - h0) / (N // bs) * 1e3, (t2 - t0) / (N // bs) * 1e3, {k: v for k, v in tr.stats.items() if k != 'host_enqueue_s'}))
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)

"""where the epoch loop's time goes: host time of each part of a step (no device waits inside the loop) against the device time
    python scripts/exp/epoch_host.py [--n 4096] [--batch 64] [--T 128]"""
import argparse, os, sys, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import al_synth
from hual_amd import al, lib
from hual_amd.dataset import DeviceDataset
from hual_amd.model import SeqPAN
from hual_amd.train import Trainer

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=4096); ap.add_argument('--batch', type=int, default=64); ap.add_argument('--T', type=int, default=128)
ap.add_argument('--no-graph', action='store_true')
a = ap.parse_args()
recs, vis, data_gt, _ = al_synth.make_trainset(a.n, 512, 1024, a.T, seed=11, num_words=1000, num_chars=40, max_words=20)
cfg = lib.make_cfg(vdim=1024, max_vlen=a.T, num_words=1000, num_chars=40)
wv = np.random.default_rng(777).normal(0, 0.4, size=(998, 300)).astype(np.float32)
m = SeqPAN(cfg, wv)
ds = DeviceDataset(recs, vis)
s0, e0 = al.labels_from_times(data_gt, ds.vlen_h); ds.set_labels(s0, e0)
tr = Trainer(m, world=1, use_graph=not a.no_graph)
g = np.random.default_rng(0)
for _ in range(2):
    tr.run_epoch(ds, g.permutation(a.n), a.batch, lr=1e-4, drop_rate=0.2)
torch.cuda.synchronize()
print('warm:', tr.stats)
# the loop of Trainer.run_epoch, instrumented
order = g.permutation(a.n).astype(np.int32)
order_dev = torch.from_numpy(order).to(m.device)
feeds = tr._feeds
bank = torch.empty((a.n + a.batch - 1) // a.batch, 2, tr.spans.shape[1], dtype=torch.int64, device=m.device)
t = dict(assemble=0.0, set_batch=0.0, step=0.0, bank=0.0)
torch.cuda.synchronize()
T0 = time.perf_counter()
for i, lo in enumerate(range(0, a.n, a.batch)):
    sel = order[lo:lo + a.batch]
    t0 = time.perf_counter()
    f = ds.assemble(sel, min_chars=4, buffers=feeds, sel_dev=order_dev[lo:lo + len(sel)])
    t1 = time.perf_counter()
    tr.set_batch_device(f)
    t2 = time.perf_counter()
    tr.step(lr=1e-4, drop_rate=0.2)
    t3 = time.perf_counter()
    bank[i].copy_(tr.spans)
    t4 = time.perf_counter()
    t['assemble'] += t1 - t0; t['set_batch'] += t2 - t1; t['step'] += t3 - t2; t['bank'] += t4 - t3
Th = time.perf_counter() - T0
torch.cuda.synchronize()
Td = time.perf_counter() - T0
n = (a.n + a.batch - 1) // a.batch
print('steps %d   host enqueue %.3f ms/step   wall (host + device drain) %.3f ms/step' % (n, Th / n * 1e3, Td / n * 1e3))
print('host parts, us/step:', {k: round(v / n * 1e6, 1) for k, v in t.items()})
print('modes:', tr.stats)

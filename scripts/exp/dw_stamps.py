"""Per-workgroup clock stamps of the balanced weight-gradient launch (debug build: HUAL_STAMPS=1 python -m hual_amd.build)."""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from hual_amd import lib
from hual_amd.model import SeqPAN
from hual_amd.train import Trainer
dev = torch.device('cuda', 0)
cfg = lib.make_cfg(vdim=1024, max_vlen=128, num_words=1000, num_chars=40)
wv = np.random.default_rng(777).normal(0, 0.4, size=(998, 300)).astype(np.float32)
model = SeqPAN(cfg, wv, device=dev, seed=12345, rng_seed=12345)
b = bench.synth_batch(64, 128, 20, 8, 1024, 1000, 40, 12345)
tr = Trainer(model, world=1, use_graph=False)
tr.set_batch(b['video'], b['lens'], b['word_ids'], b['char_ids'], b['y1'], b['y2'], b['match'], b['inner'])
for _ in range(30):
    tr.step(lr=1e-4, drop_rate=0.2)
torch.cuda.synchronize()
l = lib.load()
S = 16
n = 2048 * S
buf = (ctypes.c_ulonglong * n)()
l.hual_debug_dw_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = l.hual_debug_dw_stamps(buf, n)
st = np.frombuffer(buf, dtype=np.uint64).reshape(2048, S).astype(np.int64)
nb = int((st[:, 0] > 0).sum())
st = st[:nb]
np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'gpurun_out', 'dw_stamps.npy'), st)
t0 = st[:, 0].min()
print('rc', rc, 'blocks', nb, 'kernel span (cycles of the 100 MHz counter?)', st[:, 2].max() - t0)
dur = st[:, 2] - st[:, 0]
print('block duration: mean %.0f min %d max %d' % (dur.mean(), dur.min(), dur.max()))
print('start offsets: p50 %d p90 %d max %d' % tuple(np.percentile(st[:, 0] - t0, [50, 90, 100])))
print('search: mean %.0f' % (st[:, 1] - st[:, 0]).mean())
print('tiles/block: min %d max %d; segments: mean %.2f max %d' % (st[:, 3].min(), st[:, 3].max(), st[:, 4].mean(), st[:, 4].max()))
per_tile = dur / np.maximum(st[:, 3], 1)
order = np.argsort(dur)
print('slowest blocks (block, first job, tiles, segs, nonplain segs, duration, per tile):')
for k in order[-12:]:
    print('   %4d job %2d tiles %3d segs %d np %d dur %6d per-tile %.0f' % (k, st[k, 5], st[k, 3], st[k, 4], st[k, 6], dur[k], per_tile[k]))
print('fastest blocks:')
for k in order[:6]:
    print('   %4d job %2d tiles %3d segs %d np %d dur %6d per-tile %.0f' % (k, st[k, 5], st[k, 3], st[k, 4], st[k, 6], dur[k], per_tile[k]))
print('duration by first job:')
for j in np.unique(st[:, 5]):
    m = st[:, 5] == j
    print('   job %2d blocks %3d mean dur %7.0f per-tile %6.0f  end (from kernel start) max %d' % (j, m.sum(), dur[m].mean(), per_tile[m].mean(), (st[m, 2] - t0).max()))

"""Per-phase cycle deltas of the kernel selected by the HUAL_STAMPS=<n> debug build (csrc/tilecore.h): mean over the
workgroups of the LAST launch of that kernel in a training step at the bench shape."""
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
    tr.step(lr=1e-4, drop_rate=float(os.environ.get("DROP", "0.2")))
torch.cuda.synchronize()
l = lib.load()
if os.environ.get('HUAL_STAMPS_FIRST'):      # (build with HUAL_STAMPS_FIRST=1 as well): the first launch of the kernel in one more step
    print('reset', l.hual_debug_stamps_reset())
    tr.step(lr=1e-4, drop_rate=float(os.environ.get("DROP", "0.2")))
    torch.cuda.synchronize()
n = 512 * 64
buf = (ctypes.c_ulonglong * n)()
l.hual_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = l.hual_debug_stamps(buf, n)
st = np.frombuffer(buf, dtype=np.uint64).reshape(512, 64).astype(np.int64)
nb = int((st[:, 0] > 0).sum())
st = st[:nb]
used = [k for k in range(st.shape[1]) if (st[:, k] > 0).all()]      # (slot numbers need not be contiguous)
ns = len(used)
print('rc', rc, 'blocks', nb, 'stamps', ns)
sv = st[:, used]
d = np.diff(sv, axis=1)
tot = sv[:, -1] - sv[:, 0]
print('total cycles per workgroup: mean %.0f min %d max %d' % (tot.mean(), tot.min(), tot.max()))
print('kernel span (first start .. last end) cycles:', sv[:, -1].max() - sv[:, 0].min())
for k in range(ns - 1):
    print('%2d -> %2d  mean %7.0f  p10 %7.0f  p90 %7.0f' % (used[k], used[k + 1], d[:, k].mean(), np.percentile(d[:, k], 10), np.percentile(d[:, k], 90)))

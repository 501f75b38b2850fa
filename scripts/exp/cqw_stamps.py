"""Clock stamps of wave 0 around every barrier of the long-clip context-query kernels (csrc/cqwide.hip; a library built with
-DHUAL_STAMPS=1, loaded through HUAL_LIB_PATH): cycles between consecutive stamps, mean over the workgroups of each direction.
   scripts/exp/cqw_stamps.py [B] [T]"""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from hual_amd import lib
from hual_amd.model import SeqPAN
from hual_amd.train import Trainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device('cuda', 0)
cfg = lib.make_cfg(vdim=1024, max_vlen=T, num_words=1000, num_chars=40)
wv = np.random.default_rng(777).normal(0, 0.4, size=(998, 300)).astype(np.float32)
model = SeqPAN(cfg, wv, device=dev, seed=12345, rng_seed=12345)
b = bench.synth_batch(B, T, 20, 8, 1024, 1000, 40, 12345)
tr = Trainer(model, world=1, use_graph=False)
tr.set_batch(b['video'], b['lens'], b['word_ids'], b['char_ids'], b['y1'], b['y2'], b['match'], b['inner'])
for _ in range(30):
    tr.step(lr=1e-4, drop_rate=0.2)
torch.cuda.synchronize()
l = lib.load()
n = 256 * 64
buf = (ctypes.c_ulonglong * n)()
l.hual_debug_cqw_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert l.hual_debug_cqw_stamps(buf, n) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(256, 64).astype(np.int64)
gx = (B + 7) & ~7
for d in (0, 1):
    blk = st[(d * gx) % 256:(d * gx) % 256 + min(B, 64)]
    for base, nm in ((0, 'fwd'), (32, 'bwd')):
        row = blk[:, base:base + 32]
        nst = int((row[0] > 0).sum())
        df = np.diff(row[:, :nst], axis=1).mean(axis=0)
        print('dir %d %s  total %6.0f cycles   phases (compute | barrier wait, ...): %s' % (d, nm, (row[:, nst - 1] - row[:, 0]).mean(), ' '.join('%d%s' % (v, '|' if i % 2 == 0 else ',') for i, v in enumerate(df))))
if hasattr(l, 'hual_debug_cqw_wstamps'):
    l.hual_debug_cqw_wstamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert l.hual_debug_cqw_wstamps(buf, n) == 0
    ws = np.frombuffer(buf, dtype=np.uint64).reshape(256, 4, 16).astype(np.int64)[:min(B, 64)]      # direction 0 workgroups
    nw = int((ws[0, 0] > 0).sum())
    t0 = ws[:, 0, :nw].min(axis=1, keepdims=True)
    for sl, nm in enumerate(('wave start', 'loads issued', 'loads arrived', 'before barrier 1')):
        print('bwd dir 0 %-18s per wave (cycles after the first wave start, mean over workgroups): %s' % (nm, ' '.join('%6d' % v for v in (ws[:, sl, :nw] - t0).mean(axis=0))))

"""Phase clock stamps of the LARGE job (kind 0: 128 x 128 video self attention) of the last attn_bwd_chain_kernel launch of a step
(debug build -DHUAL_STAMPS=1).  Stamps: 0 entry, 1 staging issued + stored, 2 behind the barrier, 3 products done, 4 behind the barrier, 5 end.
    HUAL_LIB_PATH=build_exp/x.so python scripts/exp/attn_chain_stamps.py"""
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
n = 4096 * 8
buf = (ctypes.c_ulonglong * n)()
l.hual_debug_attn_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
l.hual_debug_attn_stamps(buf, n)
G = 1024
st = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)[:G]
bid = np.arange(G); xcd = bid & 7; q = G >> 3
lid = (xcd * q + (bid >> 3)) >> 3
per = G >> 6; cpx = per >> 1
x = lid // per; w = lid - x * per
kind = 1 - w // cpx
s = st[kind == 0]
t0 = st[:, 0].min()
names = ['issue+store', 'barrier', 'products', 'barrier', 'epilogue']
d = np.diff(s[:, 0:6], axis=1)
print('large job (%d workgroups): entry after launch start mean %.0f max %.0f | ' % (len(s), (s[:, 0] - t0).mean(), (s[:, 0] - t0).max()) +
      '  '.join('%s %6.0f' % (names[k], d[:, k].mean()) for k in range(5)) + '  | total %6.0f' % (s[:, 5] - s[:, 0]).mean())
c = st[kind == 1]
print('chain workgroups: last job only (stamps overwritten): entry->end of job 3 %.0f; launch span %.0f cycles' % ((c[:, 5] - c[:, 0]).mean(), st[:, 5].max() - t0))

"""Phase clock stamps of attn_bwd_kernel per workgroup (debug build: HUAL_STAMPS=1): the LAST backward attention launch of a step
(the 4-job launch of dual-attention layer 0).  Stamps: 0 entry, 1 staging issued + stored, 2 behind the barrier, 3 products done, 4 behind
the barrier, 5 end."""
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
st = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)[:2048]
t0 = st[:, 0].min()
# logical id: xcd-aware; job = (lid >> 3) % 4 - recover it from blockIdx as the kernel does
bid = np.arange(2048); xcd = bid & 7; q = 2048 >> 3
lid = xcd * q + (bid >> 3); job = (lid >> 3) % 4
names = ['issue+store', 'barrier', 'products', 'barrier', 'epilogue']
print('launch span: first entry -> last end %d cycles' % (st[:, 5].max() - t0))
for jb, nm in enumerate(('v-self 128x128', 'v->q 128x20', 'q-self 20x20', 'q->v 20x128')):
    s = st[job == jb]
    d = np.diff(s[:, 0:6], axis=1)
    print('%-16s entry (after launch start) mean %7.0f max %7.0f | ' % (nm, (s[:, 0] - t0).mean(), (s[:, 0] - t0).max()) +
          '  '.join('%s %6.0f' % (names[k], d[:, k].mean()) for k in range(5)) + '  | total %6.0f' % (s[:, 5] - s[:, 0]).mean())

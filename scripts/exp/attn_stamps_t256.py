"""attn_stamps.py at another shape: phase stamps of the last four-job attn_bwd_kernel launch of a step (B, T from the command line;
a -DHUAL_STAMPS=1 build through HUAL_LIB_PATH).  Jobs are numbered by cost as the kernel sorts them (0 = the T x T self-attention)."""
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
n = 4096 * 8
buf = (ctypes.c_ulonglong * n)()
l.hual_debug_attn_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
l.hual_debug_attn_stamps(buf, n)
G = B * 8 * 4
st = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)[:G]
t0 = st[:, 0].min()
bid = np.arange(G); xcd = bid & 7; q = G >> 3
lid = (xcd * q + (bid >> 3)) >> 3
per = G >> 6
cpx = per // 4
w = lid % per
job = w // cpx
names = ['issue+store', 'barrier', 'products', 'barrier', 'epilogue']
print('launch span: first entry -> last end %d cycles' % (st[:, 5].max() - t0))
for jb in range(4):
    s = st[job == jb]
    d = np.diff(s[:, 0:6], axis=1)
    print('job %d: entry (after launch start) mean %7.0f max %7.0f | ' % (jb, (s[:, 0] - t0).mean(), (s[:, 0] - t0).max()) +
          '  '.join('%s %6.0f' % (names[k], d[:, k].mean()) for k in range(5)) + '  | total %6.0f  end mean %7.0f max %7.0f' % ((s[:, 5] - s[:, 0]).mean(), (s[:, 5] - t0).mean(), (s[:, 5] - t0).max()))

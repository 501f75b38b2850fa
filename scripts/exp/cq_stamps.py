"""Phase clock stamps of cq_fwd_kernel / cq_bwd_kernel (debug build: HUAL_STAMPS=1 python -m hual_amd.build)."""
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
n = 256 * 32
buf = (ctypes.c_ulonglong * n)()
l.hual_debug_cq_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = l.hual_debug_cq_stamps(buf, n)
st = np.frombuffer(buf, dtype=np.uint64).reshape(256, 32).astype(np.int64)[:128]
names_f = ['score', 'row softmax', 'col softmax + sync', 'save Sr/Sc', 'c2q', 'M2 + sync', 'q2c']
names_b = ['load S', 'dSr', 'dM2,dXb + sync', 'dSc,dXa + sync', 'softmax bwd rows', 'softmax bwd cols', 'dS0 rows', 'dD1W,dD2']
for d, nm in ((0, 'dir 0 (x1 = video)'), (1, 'dir 1 (x1 = query)')):
    blk = st[64 * d:64 * d + 64]
    print(nm)
    df = np.diff(blk[:, 0:8], axis=1)
    for k, n_ in enumerate(names_f):
        print('   fwd %-20s mean %7.0f  max %7.0f' % (n_, df[:, k].mean(), df[:, k].max()))
    print('   fwd total %.0f' % (blk[:, 7] - blk[:, 0]).mean())
    print('   fwd softmax phase of wave 0: X image stores %.0f, row softmax %.0f, column softmax %.0f, barrier %.0f' % ((blk[:, 8] - blk[:, 2]).mean(), (blk[:, 9] - blk[:, 8]).mean(), (blk[:, 10] - blk[:, 9]).mean(), (blk[:, 3] - blk[:, 10]).mean()))
    db = np.diff(blk[:, 16:24], axis=1)
    for k, n_ in enumerate(names_b[:7]):
        print('   bwd %-20s mean %7.0f  max %7.0f' % (n_, db[:, k].mean(), db[:, k].max()))
    print('   bwd total %.0f' % (blk[:, 23] - blk[:, 16]).mean())

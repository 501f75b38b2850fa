import os, sys, ctypes, numpy as np, torch
# NOTE: needs the temporary STAMP() instrumentation of gemm_bf16_body + hual_debug_stamps() (see git history of this file's commit);
# kept as the record of how the in-kernel phase split quoted in DESIGN.md 6 was measured.
sys.path.insert(0, os.getcwd())
import bench
from hual_amd import lib
from hual_amd.model import SeqPAN
cfg = lib.make_cfg(vdim=1024, max_vlen=128, num_words=1000, num_chars=40)
wv = np.random.default_rng(777).normal(0, 0.4, size=(998, 300)).astype(np.float32)
m = SeqPAN(cfg, wv)
b = bench.synth_batch(64, 128, 20, 8, 1024, 1000, 40, 12345)
labels = (b['y1'], b['y2'], b['match'], b['inner'])
for _ in range(20):
    m.forward(b['video'], b['lens'], b['word_ids'], b['char_ids'], drop_rate=0.2, labels=labels)
torch.cuda.synchronize()
buf = torch.zeros(4096, dtype=torch.int64, device='cuda')
l = lib.load()
l.hual_debug_stamps.argtypes = [ctypes.c_void_p]
l.hual_debug_stamps(ctypes.c_void_p(buf.data_ptr()))
m.forward(b['video'], b['lens'], b['word_ids'], b['char_ids'], drop_rate=0.2, labels=labels)
torch.cuda.synchronize()
t = buf.cpu().numpy()
n = int((t != 0).sum())
t = t[:n]
print('stamps', n)
# stamps per job (nstages<=2): start, before-wait, after-wait, after-barrier, before-epilogue, end  => 6
d = np.diff(t)
for i in range(0, n, 6):
    seg = t[i:i + 6]
    if len(seg) < 6: break
    print(i // 6, 'issue %5d  wait %5d  barrier %5d  compute %5d  epilogue %5d  | gap-to-next %s' % (
        seg[1] - seg[0], seg[2] - seg[1], seg[3] - seg[2], seg[4] - seg[3], seg[5] - seg[4], (t[i + 6] - seg[5]) if i + 6 < n else '-'))

"""device-fed vs host-fed Runner: parameter agreement after k steps of the first epoch, against the device-fed runner run twice"""
import sys, os
import numpy as np, torch
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, 'tests'))
import test_gpu_runner as T
from hual_amd.runner import Runner
vdim = 64
vis = T._videos(24, vdim, 0); train = T._task(192, vis, 1); test = T._task(64, vis, 2)
cfg = dict(task='synth', train=dict(batch_size=32, droprate=0.0, lr=2e-3, epochs=8, clip_norm=1.0),
           model=dict(vdim=vdim, dim=128, num_heads=8, word_dim=300, char_dim=50, max_vlen=32, attn_layer=2),
           loss=dict(match_lambda=1.0, tau=0.3, no_gumbel=True), num_chars=10)
wv = np.random.default_rng(0).normal(0, 0.4, size=(40, 300)).astype(np.float32)
class L:
    def info(self, s): pass
lr = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-4
rs = [Runner(cfg, wv, train, test, vis, ckpt_dir='/tmp/ck%d' % i, logger=L(), feed=f) for i, f in enumerate(('device', 'device', 'host'))]
for ep in range(2):
    ms = [r.train_epoch(lr) for r in rs]
    P = [r.model.params.detach().cpu().numpy() for r in rs]
    for a, b, nm in ((0, 1, 'device vs device'), (0, 2, 'device vs host  ')):
        d = np.abs(P[a] - P[b])
        print('epoch %d %s: max %.2e  frac <= 0.1 lr %.4f  frac == %.4f   metrics %s | %s' % (ep, nm, d.max(), np.mean(d <= 0.1 * lr), np.mean(d == 0), np.round(ms[a], 2), np.round(ms[b], 2)))

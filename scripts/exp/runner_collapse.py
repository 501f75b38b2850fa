"""Find a collapsing run of the runner test's training task and say what it looks like (NaN parameters? which epoch?)."""
import os, sys, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import test_gpu_runner as tr
from hual_amd.runner import Runner
for i in range(400):
    vdim = 64
    vis = tr._videos(24, vdim, 0); train = tr._task(192, vis, 1); test = tr._task(64, vis, 2)
    cfg = dict(task='synth', train=dict(batch_size=32, droprate=0.1, lr=2e-3, epochs=8, clip_norm=1.0),
               model=dict(vdim=vdim, dim=128, num_heads=8, word_dim=300, char_dim=50, max_vlen=32, attn_layer=2),
               loss=dict(match_lambda=1.0, tau=0.3, no_gumbel=True), num_chars=10)
    wv = np.random.default_rng(0).normal(0, 0.4, size=(40, 300)).astype(np.float32)
    lines = []
    class L:
        def info(self, s): lines.append(str(s))
    with tempfile.TemporaryDirectory() as d:
        r = Runner(cfg, wv, train, test, vis, ckpt_dir=str(pathlib.Path(d) / 'ckpt'), logger=L())
        before = r.test_epoch(); r.train(); after = r.test_epoch()
    if after[3] < 5.0:
        p = r.model.params
        print('run', i, 'after', after, 'NaN params', int(torch.isnan(p).sum()), 'Inf', int(torch.isinf(p).sum()), 'max|p| %.3e' % float(p[~torch.isnan(p)].abs().max()))
        for l in lines: print('   ', l[:200])
        tbl = r.model.table.unpack(p.cpu().numpy())
        worst = sorted(((float(np.nanmax(np.abs(v))) if v.size else 0.0, k) for k, v in tbl.items()), reverse=True)[:6]
        print('largest parameters:', worst)
        nan_t = [k for k, v in tbl.items() if np.isnan(v).any()]
        print('tensors with NaN:', len(nan_t), nan_t[:8])
        break
else:
    print('no collapse in 400 runs')

"""The runner test's training task N times in ONE process: how often does the short, atomics-noisy training run of
tests/test_gpu_runner.py miss its +3 mIoU bar?   scripts/exp/runner_repeat.py [N]"""
import os, sys, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import test_gpu_runner as tr
from hual_amd.runner import Runner
n = int(sys.argv[1]) if len(sys.argv) > 1 else 15
feed = sys.argv[2] if len(sys.argv) > 2 else 'device'
bad = 0
for i in range(n):
    vdim = 64
    vis = tr._videos(24, vdim, 0); train = tr._task(192, vis, 1); test = tr._task(64, vis, 2)
    cfg = dict(task='synth', train=dict(batch_size=32, droprate=0.1, lr=2e-3, epochs=8, clip_norm=1.0),
               model=dict(vdim=vdim, dim=128, num_heads=8, word_dim=300, char_dim=50, max_vlen=32, attn_layer=2),
               loss=dict(match_lambda=1.0, tau=0.3, no_gumbel=True), num_chars=10)
    wv = np.random.default_rng(0).normal(0, 0.4, size=(40, 300)).astype(np.float32)
    tests_ = []
    class L:
        def info(self, s):
            for l in str(s).splitlines():
                if l.startswith('TEST:'): tests_.append(float(l.split('\t')[4]))
    with tempfile.TemporaryDirectory() as d:
        r = Runner(cfg, wv, train, test, vis, ckpt_dir=str(pathlib.Path(d) / 'ckpt'), logger=L(), feed=feed)
        before = r.test_epoch(); r.train(); after = r.test_epoch()
    ok = after[3] > before[3] + 3.0
    best_gain = max(tests_) - before[3]
    gains = globals().setdefault('gains', []); gains.append(best_gain)
    bad += 0 if ok else 1
    print(i, 'before mIoU %.2f after %.2f %s' % (before[3], after[3], '' if ok else '  <<<< MISS'), flush=True)
print('feed', feed, 'misses', bad, 'of', n, '| best-epoch test mIoU gain: min %.1f p1 %.1f median %.1f' % (min(gains), np.percentile(gains, 1), np.median(gains)))

"""two identical runs of the synchronous feed, step by step: where do their parameters part?"""
import sys, os
import numpy as np, torch
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, 'tests'))
import parity_util as pu
import test_gpu_feeder as T
from hual_amd.train import Trainer
lr = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-4
cfg, p, wv, _, _ = pu.make_case(B=2, T=12, L=4, C=4, max_vlen=24, vdim=64)
batches = T._batches(cfg)
ms = [pu.hip_model(cfg, p, wv) for _ in range(2)]
trs = [Trainer(m, world=1, use_graph=False) for m in ms]
names = None
for ep in range(2):
    for i, b in enumerate(batches):
        gs = []
        for m, tr in zip(ms, trs):
            tr.set_batch(b['video'], b['video_seq_len'], b['word_ids'], b['char_ids'], b['y1'], b['y2'], b['match_labels'], b['inner_labels'])
            tr.step(lr=lr, drop_rate=0.0)
            torch.cuda.synchronize()
        pa, pb = (m.params.detach().cpu().numpy() for m in ms)
        d = np.abs(pa - pb)
        da, db = ms[0].state_dict(), ms[1].state_dict()
        worst = sorted(((float(np.abs(da[k] - db[k]).max()), k) for k in da), reverse=True)[:3]
        print('epoch %d step %2d shape %s: max |dp| %.2e (%.1f lr)  frac > 0.1 lr %.4f  loss %.6f / %.6f  worst %s' % (
            ep, i, T.SHAPES[i], d.max(), d.max() / lr, np.mean(d > 0.1 * lr), float(trs[0].last_loss()), float(trs[1].last_loss()),
            [(('%.1e' % w), k[-40:]) for w, k in worst]))

"""Synthetic active-learning data shared by the AL tests and scripts/bench_al_round.py (no reference data blobs exist
in this environment)."""
import numpy as np


def make_trainset(N, n_videos, vdim, max_vlen, seed, num_words=200, num_chars=30, max_words=10):
    """records (as utils/data_gen.py:98-125 leaves them), visual features, ground-truth and initial train lists"""
    g = np.random.default_rng(seed)
    vis = {'v%d' % i: g.standard_normal((int(g.integers(max(4, max_vlen // 2), max_vlen + 1)), vdim)).astype(np.float32)
           for i in range(n_videos)}
    durs = {v: float(np.round(g.uniform(10.0, 120.0), 2)) for v in vis}
    recs, data_gt, data_old = [], [], []
    for i in range(N):
        vid = 'v%d' % int(g.integers(0, n_videos))
        n, dur = vis[vid].shape[0], durs[vid]
        s = g.uniform(0, dur * 0.7)
        gt = [float(np.round(s, 2)), float(np.round(g.uniform(s + 0.05 * dur, dur), 2))]
        glance = g.uniform(gt[0], gt[1])                         # ViGA-style glance -> a short initial pseudo span
        old = [float(np.round(max(0.0, glance - 0.05 * dur), 2)), float(np.round(min(dur, glance + 0.05 * dur), 2))]
        nw = int(g.integers(3, max_words + 1))
        words = ['w%d' % int(x) for x in g.integers(2, num_words, size=nw)]
        recs.append(dict(vid=vid, duration=dur, v_len=n, words=words,
                         w_ids=[int(w[1:]) for w in words],
                         c_ids=[[int(x) for x in g.integers(1, num_chars, size=int(g.integers(1, 9)))] for _ in range(nw)]))
        data_gt.append([vid, dur, gt, ' '.join(words)])
        data_old.append([vid, dur, old, ' '.join(words)])
    return recs, vis, data_gt, data_old

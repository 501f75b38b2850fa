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


# feature frames per second behind v_len = min(max_vlen, n_features) (data_gen.py:175-178).  feature_shapes.json is among the blobs
# this checkout lacks, so v_len is derived from the annotation's duration: clips of 16 frames at 25 fps for ActivityNet,
# of 8 frames at 24 fps for Charades (the usual C3D / I3D extraction strides) - an assumption, stated wherever a number rests on it
FEATURE_RATE = {'anet': 25.0 / 16.0, 'charades': 24.0 / 8.0}


def load_lengths(task):
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'lengths_%s.npz' % task))


def make_trainset_from_lengths(task, N, vdim, max_vlen, seed, num_words=1000, num_chars=40, feats=True):
    """a training set whose (v_len, words per query, longest word) follow the reference's OWN annotations
    (tests/golden/lengths_<task>.npz, scripts/gen_lengths.py): N queries drawn without replacement (all of them if N is None),
    every query keeps its video, so queries of one video share v_len and features.  Word / char ids and features are random.
    Returns (records, visual features or {vid: n_frames} with feats=False, ground-truth list, initial train list)."""
    a = load_lengths(task)
    g = np.random.default_rng(seed)
    n_all = len(a['nwords'])
    pick = np.arange(n_all) if N is None or N >= n_all else np.sort(g.choice(n_all, size=N, replace=False))
    vids = np.unique(a['vid'][pick])
    vdur = {}
    for i in pick:
        vdur[int(a['vid'][i])] = float(a['duration'][i])
    vlen = {v: int(min(max_vlen, max(4, round(vdur[v] * FEATURE_RATE[task])))) for v in vdur}
    vis = {}
    for v in vids:
        name = 'v%d' % int(v)
        vis[name] = g.standard_normal((vlen[int(v)], vdim), dtype=np.float32) if feats else vlen[int(v)]
    recs, data_gt, data_old = [], [], []
    for i in pick:
        v = int(a['vid'][i])
        name, n, dur = 'v%d' % v, vlen[v], float(np.round(vdur[v], 2))
        s = g.uniform(0, dur * 0.7)
        gt = [float(np.round(s, 2)), float(np.round(g.uniform(s + 0.05 * dur, dur), 2))]
        glance = g.uniform(gt[0], gt[1])
        old = [float(np.round(max(0.0, glance - 0.05 * dur), 2)), float(np.round(min(dur, glance + 0.05 * dur), 2))]
        nw, mc = int(a['nwords'][i]), int(a['maxchars'][i])
        w_ids = [int(x) for x in g.integers(2, num_words, size=nw)]
        clens = g.integers(1, mc + 1, size=nw)
        clens[int(g.integers(0, nw))] = mc                       # the query's longest word
        recs.append(dict(vid=name, duration=dur, v_len=n, words=['w%d' % x for x in w_ids], w_ids=w_ids,
                         c_ids=[[int(x) for x in g.integers(1, num_chars, size=int(c))] for c in clens]))
        data_gt.append([name, dur, gt, ' '.join(recs[-1]['words'])])
        data_old.append([name, dur, old, ' '.join(recs[-1]['words'])])
    return recs, vis, data_gt, data_old

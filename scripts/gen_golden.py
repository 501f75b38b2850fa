#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING the reference's importable host modules.

Runs only in the build container (needs /root/reference).  The outputs are data
(inputs + expected outputs); no reference source travels.  Re-run:  python scripts/gen_golden.py

Pins:
  * labels.npz   - utils/data_loader.py:30-98 TrainLoader.process_batch (soft start/end labels, match labels,
                   inner labels, padding) on seeded synthetic records, incl. the edge cases st=0, et=vlen-1,
                   overlapping +-2 extensions, vlen < max_len
  * timeidx.npz  - utils/data_utils.py:110-128 time_to_index / index_to_time and :70-85 visual_feature_sampling
  * uncert.npz   - utils/utils_hual.py:144-170 get_uncert_model / infer_idx(start_prob, end_prob) (the active-learning scoring)
"""
import os
import sys
import types

import numpy as np

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')


def main():
    sys.path.insert(0, REF)
    # stubs for modules that are absent from this image and unused by the functions we call
    om = types.ModuleType('omegaconf')
    om.OmegaConf = object
    sys.modules.setdefault('omegaconf', om)
    from utils import data_loader, data_utils   # noqa: E402
    os.makedirs(OUT, exist_ok=True)

    # ---------------- labels -------------------------------------------------------------
    class C:  # configs.train.batch_size
        class train:
            batch_size = 8
    g = np.random.default_rng(2024)
    cases = []
    V = 8
    vis = {}
    recs = []
    spans = [(0, 5, 12), (3, 11, 12), (0, 0, 7), (6, 6, 7), (2, 3, 16), (0, 15, 16), (5, 9, 10), (1, 2, 3),
             (7, 12, 16), (4, 6, 9), (0, 1, 2), (10, 11, 12)]
    for i, (s, e, vlen) in enumerate(spans):
        vid = 'v%d' % i
        vis[vid] = g.standard_normal((vlen, V)).astype(np.float32)
        nw = int(g.integers(1, 6))
        w_ids = [int(x) for x in g.integers(1, 50, size=nw)]
        c_ids = [[int(x) for x in g.integers(1, 30, size=int(g.integers(1, 7)))] for _ in range(nw)]
        recs.append(dict(vid=vid, w_ids=w_ids, c_ids=c_ids, s_ind=s, e_ind=e, v_len=vlen))
    save = {}
    for bi, lo in enumerate(range(0, len(recs), 4)):
        batch = recs[lo:lo + 4]
        loader = data_loader.TrainLoader(list(batch), vis, C)
        vfeats, vlens, word_ids, char_ids, s_l, e_l, m_l, i_l = loader.process_batch(batch)
        save['b%d_s_ind' % bi] = np.array([r['s_ind'] for r in batch])
        save['b%d_e_ind' % bi] = np.array([r['e_ind'] for r in batch])
        save['b%d_vlens' % bi] = vlens
        save['b%d_vfeats' % bi] = vfeats
        save['b%d_word_ids' % bi] = word_ids
        save['b%d_char_ids' % bi] = char_ids
        save['b%d_s_labels' % bi] = s_l
        save['b%d_e_labels' % bi] = e_l
        save['b%d_match_labels' % bi] = m_l
        save['b%d_inner_labels' % bi] = i_l
        # ragged inputs needed to rebuild the batch on the other side
        save['b%d_w_lens' % bi] = np.array([len(r['w_ids']) for r in batch])
        wflat = np.concatenate([np.array(r['w_ids']) for r in batch])
        save['b%d_w_flat' % bi] = wflat
        save['b%d_c_lens' % bi] = np.concatenate([np.array([len(c) for c in r['c_ids']]) for r in batch])
        save['b%d_c_flat' % bi] = np.concatenate([np.array(c) for r in batch for c in r['c_ids']])
    save['n_batches'] = np.array(len(range(0, len(recs), 4)))
    np.savez_compressed(os.path.join(OUT, 'labels.npz'), **save)

    # ---------------- time/index + sampling ----------------------------------------------
    t = {}
    q = []
    for (st, et, n, dur) in [(0.0, 5.0, 64, 30.0), (2.4, 17.9, 64, 31.2), (10.0, 10.5, 20, 12.0), (0.0, 12.0, 20, 12.0),
                             (3.3, 8.8, 100, 117.5), (29.0, 30.0, 64, 30.0)]:
        si, ei = data_utils.time_to_index(st, et, n, dur)
        s2, e2 = data_utils.index_to_time([si, ei], n, dur)
        q.append([st, et, n, dur, si, ei, s2, e2])
    t['time_index'] = np.array(q, dtype=np.float64)
    for k, (n, m) in enumerate([(200, 64), (64, 64), (65, 64), (130, 100), (10, 64)]):
        f = g.standard_normal((n, 6)).astype(np.float32)
        t['samp%d_in' % k] = f
        t['samp%d_max' % k] = np.array(m)
        t['samp%d_out' % k] = np.asarray(data_utils.visual_feature_sampling(f, m))
    np.savez_compressed(os.path.join(OUT, 'timeidx.npz'), **t)

    # ---------------- AL uncertainty scoring ---------------------------------------------
    from utils import utils_hual   # noqa: E402
    u = {}
    for k, vlen in enumerate([12, 7, 30]):
        lg = [g.standard_normal((2, 32)).astype(np.float32) for _ in range(3)]
        u['u%d_vlen' % k] = np.array(vlen)
        u['u%d_logits' % k] = np.stack(lg)
        u['u%d_uncert' % k] = np.asarray(utils_hual.get_uncert_model(lg[1], lg[2], vlen), dtype=np.float64)
        sp = 1.0 / (1.0 + np.exp(-lg[0][0].astype(np.float64)))
        ep = 1.0 / (1.0 + np.exp(-lg[0][1].astype(np.float64)))
        u['u%d_sprob' % k], u['u%d_eprob' % k] = sp, ep
        u['u%d_idx' % k] = np.asarray(utils_hual.infer_idx(sp, ep), dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, 'uncert.npz'), **u)
    print('wrote', sorted(os.listdir(OUT)))


if __name__ == '__main__':
    main()

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
  * al.npz       - update_label.py: renew_label :85-123 on seeded cases (both branches), get_uncert_rank :125-169 and two
                   consecutive main() rounds :173-208 on a synthetic 24-sample train set (files in a temp dir), with the
                   F_renew coefficient tables :11-37 of both tasks; utils_hual.get_distance_score / center_width_gauss
"""
import os
import sys
import types

import numpy as np

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')


class _EasyDict(dict):
    """stand-in for easydict.EasyDict (absent from this image): nested attribute access, nothing else is used"""
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = _EasyDict(v) if isinstance(v, dict) else v
    __getattr__ = dict.__getitem__


def _rand_ap(g, vlen, gt, npts):
    """active points the way update_label.append_AP builds them: each observed frame is pos iff inside the GT span"""
    pos, neg = [], []
    for p in g.choice(vlen, size=min(npts, vlen), replace=False):
        (pos if gt[0] <= int(p) <= gt[1] else neg).append(int(p))
    return pos, neg


def _pad_lists(lists, width):
    out = np.full((len(lists), width), -1, dtype=np.int64)
    for i, l in enumerate(lists):
        out[i, :len(l)] = l
    return out


def gen_al(g, utils_hual):
    import json
    import pickle
    import tempfile
    ed = types.ModuleType('easydict')
    ed.EasyDict = _EasyDict
    sys.modules.setdefault('easydict', ed)
    import update_label as U   # noqa: E402
    a = {}
    # ---------------- center_width_gauss / get_distance_score -----------------------------
    rows = []
    for k, (c, w, vlen, mv) in enumerate([(3.0, 7, 20, 32), (10.5, 4, 12, 12), (0.0, 1, 5, 16), (30.0, 19.2, 64, 64)]):
        a['cwg%d_in' % k] = np.array([c, w, vlen, mv], dtype=np.float64)
        a['cwg%d_out' % k] = np.asarray(utils_hual.center_width_gauss(c, w, vlen, mv))
    for k, (pos, neg, vlen, mv) in enumerate([([], [], 10, 16), ([4, 6], [1, 9], 12, 16), ([], [3, 8], 12, 12),
                                              ([0], [], 7, 8), ([5], [6, 2], 9, 9)]):
        a['dist%d_pos' % k] = np.array(pos, dtype=np.int64)
        a['dist%d_neg' % k] = np.array(neg, dtype=np.int64)
        a['dist%d_dims' % k] = np.array([vlen, mv])
        a['dist%d_out' % k] = np.asarray(utils_hual.get_distance_score(list(pos), list(neg), vlen, mv), dtype=np.float64)
        s_, e_ = utils_hual.get_distance_score_shift(list(pos), list(neg), vlen, mv, -0.3 if pos else 0.9)
        a['dist%d_shift_s' % k], a['dist%d_shift_e' % k] = np.asarray(s_, np.float64), np.asarray(e_, np.float64)
    a['n_dist'] = np.array(5)
    # ---------------- renew_label ---------------------------------------------------------
    K, TM = 40, 48
    rn = dict(old=[], pos=[], neg=[], sp=np.zeros((K, TM), np.float32), ep=np.zeros((K, TM), np.float32), dims=[],
              coff=[], out=[])
    for k in range(K):
        mv = int(g.integers(8, TM + 1))
        vlen = int(g.integers(4, mv + 1))
        gs = int(g.integers(0, vlen - 1)); ge = int(g.integers(gs, vlen))
        pos, neg = _rand_ap(g, vlen, (gs, ge), int(g.integers(1, 5)))
        if k % 3 == 0:
            pos = []                      # force the negative-only branch regularly
            if not neg:
                neg = [int(g.integers(0, vlen))]
        os_ = int(g.integers(0, vlen)); oe = int(g.integers(os_, vlen))
        sp = U.sigmoid(g.standard_normal(mv).astype(np.float32) * 2)
        ep = U.sigmoid(g.standard_normal(mv).astype(np.float32) * 2)
        task, I = ('charades', 'anet')[k % 2], 1 + (k // 2) % 3
        coff = U.get_coff(U.F_renew, task, I)
        res = U.renew_label([os_, oe], {'pos_idx': list(pos), 'neg_idx': list(neg)}, sp.copy(), ep.copy(), vlen, mv, coff)
        rn['old'].append([os_, oe]); rn['pos'].append(pos); rn['neg'].append(neg)
        rn['sp'][k, :mv] = sp; rn['ep'][k, :mv] = ep
        rn['dims'].append([vlen, mv])
        rn['coff'].append([coff.pos.distance, coff.pos.model, coff.pos.old, coff.neg.distance, coff.neg.model, coff.neg.old,
                           coff.uncert])
        rn['out'].append([int(res[0]), int(res[1])])
    a['renew_old'] = np.array(rn['old']); a['renew_pos'] = _pad_lists(rn['pos'], 8); a['renew_neg'] = _pad_lists(rn['neg'], 8)
    a['renew_sprob'], a['renew_eprob'] = rn['sp'], rn['ep']
    a['renew_dims'] = np.array(rn['dims']); a['renew_coff'] = np.array(rn['coff'], dtype=np.float64)
    a['renew_out'] = np.array(rn['out'])
    # ---------------- get_uncert_rank + main(): two consecutive rounds ---------------------
    N, TMAX = 24, 40
    dur = np.round(g.uniform(8.0, 60.0, size=N), 2)
    vlen = g.integers(6, 33, size=N)
    tm = np.array([int(g.integers(v, TMAX + 1)) for v in vlen])      # padded length of the batch the sample was in
    gt = np.zeros((N, 2)); old = np.zeros((N, 2))
    for i in range(N):
        s_ = g.uniform(0, dur[i] * 0.7); gt[i] = [round(s_, 2), round(g.uniform(s_ + 0.5, dur[i]), 2)]
        s_ = g.uniform(0, dur[i] * 0.7); old[i] = [round(s_, 2), round(g.uniform(s_ + 0.5, dur[i]), 2)]
    a['al_dur'], a['al_vlen'], a['al_tm'], a['al_gt'], a['al_old0'] = dur, vlen, tm, gt, old
    tmp = tempfile.mkdtemp()
    data_gt = [['v%d' % i, float(dur[i]), [float(gt[i, 0]), float(gt[i, 1])], 'q %d' % i] for i in range(N)]
    data_old = [['v%d' % i, float(dur[i]), [float(old[i, 0]), float(old[i, 1])], 'q %d' % i] for i in range(N)]
    gt_path = os.path.join(tmp, 'gt.json')
    json.dump(data_gt, open(gt_path, 'w'))
    U.GT_PATH = gt_path
    cur_path = os.path.join(tmp, 're0.json')
    json.dump(data_old, open(cur_path, 'w'))
    for rnd, task in ((1, 'charades'), (2, 'anet')):
        logits = np.zeros((N, 3, 2, TMAX), np.float32)
        prop = []
        for i in range(N):
            lg = g.standard_normal((3, 2, tm[i])).astype(np.float32) * 1.5
            lg[1] = lg[0] + 0.3 * g.standard_normal((2, tm[i])).astype(np.float32)   # MC-dropout passes near the base pass
            lg[2] = lg[0] + 0.3 * g.standard_normal((2, tm[i])).astype(np.float32)
            logits[i, :, :, :tm[i]] = lg
            prop.append({'vid': 'v%d' % i, 'v_len': int(vlen[i]), 'prop_logits': [lg[0, 0], lg[0, 1]],
                         'prop_logits1': [lg[1, 0], lg[1, 1]], 'prop_logits2': [lg[2, 0], lg[2, 1]]})
        a['al_r%d_logits' % rnd] = logits
        coff = U.get_coff(U.F_renew, task, rnd)
        a['al_r%d_coff' % rnd] = np.array([coff.pos.distance, coff.pos.model, coff.pos.old, coff.neg.distance,
                                            coff.neg.model, coff.neg.old, coff.uncert], dtype=np.float64)
        prop_path = os.path.join(tmp, 'prop%d.pkl' % rnd)
        pickle.dump(prop, open(prop_path, 'wb'))
        # ranking as main() sees it (a fresh load: main() mutates its own copy)
        d0 = json.load(open(cur_path))
        if len(d0[0]) == 4:
            for r in d0:
                r.append({'pos_idx': [], 'neg_idx': []})
        rank = U.get_uncert_rank(d0, data_gt, prop, coff)
        a['al_r%d_rank_idx' % rnd] = np.array([r['idx'] for r in rank])
        uf = np.zeros((N, TMAX), np.float64)
        for r in rank:
            uf[r['idx'], :r['max_vlen']] = r['uncert_frame']
        a['al_r%d_uncert_frame' % rnd] = uf
        uv = np.zeros(N, np.float64)
        for r in rank:
            uv[r['idx']] = r['uncert_video']
        a['al_r%d_uncert_video' % rnd] = uv
        a['al_r%d_gt_idx' % rnd] = np.array([r['gt_idx'] for r in sorted(rank, key=lambda r: r['idx'])])
        a['al_r%d_old_idx' % rnd] = np.array([r['old_idx'] for r in sorted(rank, key=lambda r: r['idx'])])
        new_path = os.path.join(tmp, 're%d.json' % rnd)
        U.main(cur_path, new_path, prop_path, coff)
        new = json.load(open(new_path))
        a['al_r%d_new_time' % rnd] = np.array([r[2] for r in new], dtype=np.float64)
        a['al_r%d_new_pos' % rnd] = _pad_lists([r[4]['pos_idx'] for r in new], 8)
        a['al_r%d_new_neg' % rnd] = _pad_lists([r[4]['neg_idx'] for r in new], 8)
        cur_path = new_path
    np.savez_compressed(os.path.join(OUT, 'al.npz'), **a)


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
    gen_al(g, utils_hual)
    print('wrote', sorted(os.listdir(OUT)))


if __name__ == '__main__':
    main()

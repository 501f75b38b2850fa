"""The active-learning oracle (oracle/al_ref.py) against fixtures produced by the reference's own update_label.py /
utils_hual.py (scripts/gen_golden.py -> tests/golden/al.npz, uncert.npz)."""
import os

import numpy as np

from oracle import al_ref as A


def _lst(row):
    return [int(x) for x in row if x >= 0]


def al_dataset(g, rnd):
    """rebuild the python structures update_label.main() sees in round `rnd` of the fixture"""
    N = len(g['al_dur'])
    dur, vlen, tm = g['al_dur'], g['al_vlen'], g['al_tm']
    data_gt = [['v%d' % i, float(dur[i]), [float(x) for x in g['al_gt'][i]], 'q %d' % i] for i in range(N)]
    if rnd == 1:
        data_old = [['v%d' % i, float(dur[i]), [float(x) for x in g['al_old0'][i]], 'q %d' % i] for i in range(N)]
    else:
        p = rnd - 1
        data_old = [['v%d' % i, float(dur[i]), [float(x) for x in g['al_r%d_new_time' % p][i]], 'q %d' % i,
                     {'pos_idx': _lst(g['al_r%d_new_pos' % p][i]), 'neg_idx': _lst(g['al_r%d_new_neg' % p][i])}]
                    for i in range(N)]
    lg = g['al_r%d_logits' % rnd]
    prop = [{'vid': 'v%d' % i, 'v_len': int(vlen[i]), 'prop_logits': [lg[i, 0, 0, :tm[i]], lg[i, 0, 1, :tm[i]]],
             'prop_logits1': [lg[i, 1, 0, :tm[i]], lg[i, 1, 1, :tm[i]]],
             'prop_logits2': [lg[i, 2, 0, :tm[i]], lg[i, 2, 1, :tm[i]]]} for i in range(N)]
    return data_old, data_gt, prop, tuple(float(x) for x in g['al_r%d_coff' % rnd])


def test_gauss_and_distance_scores(golden_dir):
    g = np.load(os.path.join(golden_dir, 'al.npz'))
    for k in range(4):
        c, w, vlen, mv = g['cwg%d_in' % k]
        c = int(c) if float(c).is_integer() else float(c)
        w = int(w) if float(w).is_integer() else float(w)
        out = A.center_width_gauss(c, w, int(vlen), int(mv))
        assert out.dtype == np.float32
        np.testing.assert_array_equal(out, g['cwg%d_out' % k])
    for k in range(int(g['n_dist'])):
        pos, neg = _lst(g['dist%d_pos' % k]), _lst(g['dist%d_neg' % k])
        vlen, mv = [int(x) for x in g['dist%d_dims' % k]]
        np.testing.assert_array_equal(A.get_distance_score(pos, neg, vlen, mv), g['dist%d_out' % k])
        s, e = A.get_distance_score_shift(pos, neg, vlen, mv, -0.3 if pos else 0.9)
        np.testing.assert_array_equal(s, g['dist%d_shift_s' % k])
        np.testing.assert_array_equal(e, g['dist%d_shift_e' % k])


def test_uncert_model_and_infer_idx(golden_dir):
    g = np.load(os.path.join(golden_dir, 'uncert.npz'))
    for k in range(3):
        lg, vlen = g['u%d_logits' % k], int(g['u%d_vlen' % k])
        np.testing.assert_array_equal(A.get_uncert_model(lg[1], lg[2], vlen), g['u%d_uncert' % k].astype(np.float32))


def test_renew_label_cases(golden_dir):
    g = np.load(os.path.join(golden_dir, 'al.npz'))
    K = len(g['renew_out'])
    branches = set()
    for k in range(K):
        vlen, mv = [int(x) for x in g['renew_dims'][k]]
        ap = {'pos_idx': _lst(g['renew_pos'][k]), 'neg_idx': _lst(g['renew_neg'][k])}
        branches.add(bool(ap['pos_idx']))
        out = A.renew_label([int(x) for x in g['renew_old'][k]], ap, g['renew_sprob'][k, :mv].copy(),
                            g['renew_eprob'][k, :mv].copy(), vlen, mv, tuple(g['renew_coff'][k]))
        assert out == [int(x) for x in g['renew_out'][k]], k
    assert branches == {True, False}


def test_two_update_rounds(golden_dir):
    g = np.load(os.path.join(golden_dir, 'al.npz'))
    for rnd in (1, 2):
        data_old, data_gt, prop, coff = al_dataset(g, rnd)
        if rnd == 1:
            probe = [r + [{'pos_idx': [], 'neg_idx': []}] for r in data_old]
        else:
            probe = [r[:4] + [{'pos_idx': list(r[4]['pos_idx']), 'neg_idx': list(r[4]['neg_idx'])}] for r in data_old]
        rank = A.get_uncert_rank(probe, data_gt, prop, coff)
        np.testing.assert_array_equal([r['idx'] for r in rank], g['al_r%d_rank_idx' % rnd])
        for r in rank:
            np.testing.assert_array_equal(r['uncert_frame'], g['al_r%d_uncert_frame' % rnd][r['idx'], :r['max_vlen']])
            assert float(r['uncert_video']) == g['al_r%d_uncert_video' % rnd][r['idx']]
            assert r['gt_idx'] == [int(x) for x in g['al_r%d_gt_idx' % rnd][r['idx']]]
            assert r['old_idx'] == [int(x) for x in g['al_r%d_old_idx' % rnd][r['idx']]]
        new = A.update_labels(data_old, data_gt, prop, coff)
        np.testing.assert_array_equal(np.array([r[2] for r in new]), g['al_r%d_new_time' % rnd])
        for i, r in enumerate(new):
            assert r[4]['pos_idx'] == _lst(g['al_r%d_new_pos' % rnd][i])
            assert r[4]['neg_idx'] == _lst(g['al_r%d_new_neg' % rnd][i])


def test_coefficient_tables():
    assert A.get_coff('charades', 1) == (4.0, 0.8, 1.0, 2.0, 2.4, 1.0, 0.25)
    assert A.get_coff('anet', 3) == (1.6, 2.0, 1.0, 1.6, 2.0, 1.0, 0.25)

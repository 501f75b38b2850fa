"""GPU: the active-learning label update kernels (csrc/al.hip through hual_al_score / hual_al_renew) against the
reference-generated fixtures (tests/golden/al.npz) and against the CPU oracle (oracle/al_ref.py) on larger seeded sets.
Indices (frame to annotate, ranking, new pseudo spans, new times) must be equal; float fields within 1e-6."""
import copy
import os

import numpy as np
import pytest

from oracle import al_ref as A
from test_al_golden import al_dataset, _lst

pytestmark = pytest.mark.gpu


def test_two_update_rounds_match_reference_fixture(golden_dir):
    from hual_amd import al
    g = np.load(os.path.join(golden_dir, 'al.npz'))
    for rnd in (1, 2):
        data_old, data_gt, prop, coff = al_dataset(g, rnd)
        new, dbg = al.update_labels(data_old, data_gt, prop, coff, return_debug=True)
        np.testing.assert_array_equal(dbg['order'], g['al_r%d_rank_idx' % rnd])
        np.testing.assert_allclose(dbg['uncert_video'], g['al_r%d_uncert_video' % rnd], rtol=1e-6, atol=1e-6)
        tm = g['al_tm']
        for i in range(len(tm)):
            np.testing.assert_allclose(dbg['uncert_frame'][i, :tm[i]], g['al_r%d_uncert_frame' % rnd][i, :tm[i]],
                                       rtol=1e-6, atol=1e-6)
        np.testing.assert_array_equal(dbg['gt_idx'], g['al_r%d_gt_idx' % rnd])
        np.testing.assert_array_equal(dbg['old_idx'], g['al_r%d_old_idx' % rnd])
        np.testing.assert_array_equal(np.array([r[2] for r in new]), g['al_r%d_new_time' % rnd])
        for i, r in enumerate(new):
            assert r[4]['pos_idx'] == _lst(g['al_r%d_new_pos' % rnd][i])
            assert r[4]['neg_idx'] == _lst(g['al_r%d_new_neg' % rnd][i])


def test_renew_label_fixture_cases(golden_dir):
    from hual_amd import al
    g = np.load(os.path.join(golden_dir, 'al.npz'))
    K = len(g['renew_out'])
    # hual_al_renew takes probabilities, so feed logits whose sigmoid is the fixture's probability: run the kernel on the
    # probabilities directly by building the updater and overwriting its sprob / eprob
    import torch
    for k in range(K):
        vlen, mv = [int(x) for x in g['renew_dims'][k]]
        prop = [{'vid': 'v', 'v_len': vlen, 'prop_logits': [np.zeros(mv, np.float32)] * 2,
                 'prop_logits1': [np.zeros(mv, np.float32)] * 2, 'prop_logits2': [np.zeros(mv, np.float32)] * 2}]
        aps = [[(f, True) for f in _lst(g['renew_pos'][k])] + [(f, False) for f in _lst(g['renew_neg'][k])]]
        up = al.LabelUpdater(prop, aps)
        up.sprob[0, :mv] = torch.from_numpy(g['renew_sprob'][k, :mv]).to(up.dev)
        up.eprob[0, :mv] = torch.from_numpy(g['renew_eprob'][k, :mv]).to(up.dev)
        out = up.renew(np.array([0]), g['renew_old'][k][None, :], tuple(g['renew_coff'][k]))
        assert [int(x) for x in out[0]] == [int(x) for x in g['renew_out'][k]], k


def _synthetic_round(N, tmax, seed, with_aps):
    g = np.random.default_rng(seed)
    dur = np.round(g.uniform(8.0, 200.0, size=N), 2)
    vlen = g.integers(6, tmax + 1, size=N)
    tm = np.array([int(g.integers(v, tmax + 1)) for v in vlen])
    data_gt, data_old, prop = [], [], []
    for i in range(N):
        s = g.uniform(0, dur[i] * 0.7); gt = [round(s, 2), round(g.uniform(s + 0.5, dur[i]), 2)]
        s = g.uniform(0, dur[i] * 0.7); old = [round(s, 2), round(g.uniform(s + 0.5, dur[i]), 2)]
        data_gt.append(['v%d' % i, float(dur[i]), gt, 'q'])
        rec = ['v%d' % i, float(dur[i]), old, 'q']
        if with_aps:
            gi = A.time_to_index_v2(gt, float(dur[i]), int(vlen[i]))
            pos, neg = [], []
            for p in g.choice(int(vlen[i]), size=int(g.integers(0, 4)), replace=False):
                (pos if gi[0] <= int(p) <= gi[1] else neg).append(int(p))
            rec.append({'pos_idx': pos, 'neg_idx': neg})
        data_old.append(rec)
        lg = g.standard_normal((3, 2, tm[i])).astype(np.float32) * 1.5
        lg[1] = lg[0] + 0.3 * g.standard_normal((2, tm[i])).astype(np.float32)
        lg[2] = lg[0] + 0.3 * g.standard_normal((2, tm[i])).astype(np.float32)
        prop.append({'vid': 'v%d' % i, 'v_len': int(vlen[i]), 'prop_logits': [lg[0, 0], lg[0, 1]],
                     'prop_logits1': [lg[1, 0], lg[1, 1]], 'prop_logits2': [lg[2, 0], lg[2, 1]]})
    return data_old, data_gt, prop


@pytest.mark.parametrize('N,tmax,task,I,with_aps', [(301, 64, 'charades', 1, False), (257, 100, 'anet', 2, True),
                                                     (64, 300, 'anet', 1, True)])
def test_update_round_matches_oracle(N, tmax, task, I, with_aps):
    from hual_amd import al
    data_old, data_gt, prop = _synthetic_round(N, tmax, 1000 + N, with_aps)
    coff = al.get_coff(task, I)
    assert coff == A.get_coff(task, I)
    ref = A.update_labels(copy.deepcopy(data_old), data_gt, prop, coff)
    new, dbg = al.update_labels(copy.deepcopy(data_old), data_gt, prop, coff, return_debug=True)
    mism = [i for i in range(N) if ref[i][2] != new[i][2] or ref[i][4] != new[i][4]]
    # float32 exp / sum-order differences may flip an argmax between two frames whose scores agree to ~1e-7; none is
    # expected on these seeds, a handful would be tolerated on other data
    assert len(mism) == 0, mism[:10]


def test_as_written_mode_has_zero_model_uncertainty():
    """identical logits in all three passes -> uncert_video == 0 for every sample -> ranking = sample order"""
    from hual_amd import al
    data_old, data_gt, prop = _synthetic_round(50, 64, 7, False)
    for p in prop:
        p['prop_logits1'] = p['prop_logits']
        p['prop_logits2'] = p['prop_logits']
    new, dbg = al.update_labels(data_old, data_gt, prop, al.get_coff('charades', 1), return_debug=True)
    assert float(np.abs(dbg['uncert_video']).max()) == 0.0
    np.testing.assert_array_equal(dbg['order'], np.arange(50))

"""Host batch assembly (hual_amd/data.py) against fixtures produced by the reference's own loader code
(scripts/gen_golden.py imports /root/reference/utils/data_loader.py + data_utils.py)."""
import os

import numpy as np

from hual_amd import data


def _rebuild(g, bi):
    w_lens = g['b%d_w_lens' % bi]
    w_flat = g['b%d_w_flat' % bi]
    c_lens = g['b%d_c_lens' % bi]
    c_flat = g['b%d_c_flat' % bi]
    w_ids, c_ids, wo, co, ci = [], [], 0, 0, 0
    for n in w_lens:
        w_ids.append([int(x) for x in w_flat[wo:wo + n]])
        words = []
        for _ in range(n):
            k = int(c_lens[ci]); ci += 1
            words.append([int(x) for x in c_flat[co:co + k]]); co += k
        c_ids.append(words)
        wo += n
    return w_ids, c_ids


def test_label_synthesis_matches_reference_loader(golden_dir):
    g = np.load(os.path.join(golden_dir, 'labels.npz'))
    for bi in range(int(g['n_batches'])):
        lens = g['b%d_vlens' % bi]
        y1, y2, m, i = data.make_labels(g['b%d_s_ind' % bi], g['b%d_e_ind' % bi], lens)
        np.testing.assert_array_equal(y1, g['b%d_s_labels' % bi])
        np.testing.assert_array_equal(y2, g['b%d_e_labels' % bi])
        np.testing.assert_array_equal(m, g['b%d_match_labels' % bi])
        np.testing.assert_array_equal(i, g['b%d_inner_labels' % bi])
        assert y1.dtype == np.float32 and m.dtype == np.int32


def test_padding_matches_reference_loader(golden_dir):
    g = np.load(os.path.join(golden_dir, 'labels.npz'))
    for bi in range(int(g['n_batches'])):
        w_ids, c_ids = _rebuild(g, bi)
        np.testing.assert_array_equal(data.pad_word_ids(w_ids), g['b%d_word_ids' % bi])
        np.testing.assert_array_equal(data.pad_char_ids(c_ids), g['b%d_char_ids' % bi])
        vf = g['b%d_vfeats' % bi]
        lens = g['b%d_vlens' % bi]
        v, l = data.pad_video([vf[b, :lens[b]] for b in range(len(lens))])
        np.testing.assert_array_equal(v, vf)
        np.testing.assert_array_equal(l, lens)


def test_time_index_and_sampling(golden_dir):
    g = np.load(os.path.join(golden_dir, 'timeidx.npz'))
    for st, et, n, dur, si, ei, s2, e2 in g['time_index']:
        a, b = data.time_to_index(st, et, int(n), dur)
        assert (a, b) == (int(si), int(ei))
        ts, te = data.index_to_time([a, b], int(n), dur)
        assert float(ts) == s2 and float(te) == e2
    k = 0
    while 'samp%d_in' % k in g:
        out = data.visual_feature_sampling(g['samp%d_in' % k], int(g['samp%d_max' % k]))
        np.testing.assert_array_equal(out, g['samp%d_out' % k])
        k += 1
    assert k == 5


def test_batched_iou_bookkeeping_equals_the_scalar_functions():
    """al.ious_of_spans (one numpy pass per batch / epoch) against the per-sample path it replaced: data.index_to_time (pinned above to
    data_utils.py:121-128 by the fixture) + calculate_iou (runner_utils.py:34-38), value for value"""
    from hual_amd import al
    g = np.random.default_rng(11)
    recs, ps, pe = [], [], []
    for _ in range(3000):
        n = int(g.integers(2, 257))
        dur = float(np.round(g.uniform(1.0, 400.0), int(g.integers(0, 7))))
        s = int(g.integers(0, n)); e = int(g.integers(s, n))
        recs.append(dict(v_len=n, duration=dur, s_ind=s, e_ind=e))
        a = int(g.integers(0, n)); b = int(g.integers(0, n))            # predictions may have end < start (argmax of an outer product does not, but the formula must agree anyway)
        ps.append(a); pe.append(b)
    want = []
    for r, a, b in zip(recs, ps, pe):
        st, et = data.index_to_time([a, b], r['v_len'], r['duration'])
        gs, ge = data.index_to_time([r['s_ind'], r['e_ind']], r['v_len'], r['duration'])
        want.append(float(al.calculate_iou([st, et], [gs, ge])))
    got = al.ious_of_spans(recs, np.array(ps), np.array(pe))
    assert got == want
    assert al.iou_metrics(got) == al.iou_metrics(want)
    assert al.ious_of_spans([], [], []) == []

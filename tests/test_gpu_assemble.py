"""GPU: device-side batch assembly (hual_assemble_batch) against the outputs of the reference's own
TrainLoader.process_batch stored in tests/golden/labels.npz - every feed bit exact."""
import os

import numpy as np
import pytest

from test_data_golden import _rebuild

pytestmark = pytest.mark.gpu


def _fixture_dataset(g):
    recs, vis, batches = [], {}, []
    for bi in range(int(g['n_batches'])):
        w_ids, c_ids = _rebuild(g, bi)
        lens, vf = g['b%d_vlens' % bi], g['b%d_vfeats' % bi]
        ids = []
        for b in range(len(lens)):
            vid = 'b%dv%d' % (bi, b)
            vis[vid] = vf[b, :lens[b]].copy()
            ids.append(len(recs))
            recs.append(dict(vid=vid, w_ids=w_ids[b], c_ids=c_ids[b], s_ind=int(g['b%d_s_ind' % bi][b]),
                             e_ind=int(g['b%d_e_ind' % bi][b]), v_len=int(lens[b])))
        batches.append(ids)
    return recs, vis, batches


def test_assembled_batches_equal_reference_loader(golden_dir):
    from hual_amd.dataset import DeviceDataset
    g = np.load(os.path.join(golden_dir, 'labels.npz'))
    recs, vis, batches = _fixture_dataset(g)
    ds = DeviceDataset(recs, vis)
    for bi, ids in enumerate(batches):
        o = ds.assemble(ids)
        np.testing.assert_array_equal(o['video'].cpu().numpy(), g['b%d_vfeats' % bi])
        np.testing.assert_array_equal(o['video_seq_len'].cpu().numpy(), g['b%d_vlens' % bi])
        np.testing.assert_array_equal(o['word_ids'].cpu().numpy(), g['b%d_word_ids' % bi])
        np.testing.assert_array_equal(o['char_ids'].cpu().numpy(), g['b%d_char_ids' % bi])
        np.testing.assert_array_equal(o['y1'].cpu().numpy(), g['b%d_s_labels' % bi])
        np.testing.assert_array_equal(o['y2'].cpu().numpy(), g['b%d_e_labels' % bi])
        np.testing.assert_array_equal(o['match_labels'].cpu().numpy(), g['b%d_match_labels' % bi])
        np.testing.assert_array_equal(o['inner_labels'].cpu().numpy(), g['b%d_inner_labels' % bi].astype(np.float32))


def test_assembly_equals_host_loader_on_a_large_shuffled_batch():
    """B=64 drawn from 300 samples with shared videos, against hual_amd/data.py (itself fixture-pinned)."""
    from hual_amd import data
    from hual_amd.dataset import DeviceDataset
    g = np.random.default_rng(5)
    V = 64
    vis = {'v%d' % i: g.standard_normal((int(g.integers(1, 65)), V)).astype(np.float32) for i in range(40)}
    recs = []
    for i in range(300):
        vid = 'v%d' % int(g.integers(0, 40))
        n = vis[vid].shape[0]
        s = int(g.integers(0, n)); e = int(g.integers(s, n))
        nw = int(g.integers(1, 12))
        recs.append(dict(vid=vid, w_ids=[int(x) for x in g.integers(1, 99, size=nw)],
                         c_ids=[[int(x) for x in g.integers(1, 30, size=int(g.integers(1, 9)))] for _ in range(nw)],
                         s_ind=s, e_ind=e))
    ds = DeviceDataset(recs, vis)
    out = None
    for rep in range(3):
        sel = g.permutation(300)[:64]
        out = ds.assemble(sel, out=out)
        ref = data.process_train_batch([recs[i] for i in sel], vis)
        for k in ('video', 'video_seq_len', 'word_ids', 'char_ids', 'y1', 'y2', 'match_labels'):
            np.testing.assert_array_equal(out[k].cpu().numpy(), ref[k], err_msg=k)
        np.testing.assert_array_equal(out['inner_labels'].cpu().numpy(), ref['inner_labels'].astype(np.float32))
    t = ds.assemble(sel[:5], labels=False)
    assert 'y1' not in t

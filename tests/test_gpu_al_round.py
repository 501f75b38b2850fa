"""GPU: one whole active-learning round (update_label -> train -> infer_trainset, run_charades.py:9-41) on a small
synthetic training set that lives in HBM."""
import copy

import numpy as np
import pytest
import torch

import al_synth
from oracle import al_ref as A

pytestmark = pytest.mark.gpu


def test_one_round_end_to_end():
    from hual_amd import al, lib
    from hual_amd.dataset import DeviceDataset
    from hual_amd.model import SeqPAN
    N, vdim, max_vlen = 40, 64, 24
    recs, vis, data_gt, data_old = al_synth.make_trainset(N, 12, vdim, max_vlen, seed=3)
    cfg = lib.make_cfg(vdim=vdim, max_vlen=max_vlen, num_words=200, num_chars=30)
    wv = np.random.default_rng(1).normal(0, 0.4, size=(198, 300)).astype(np.float32)
    model = SeqPAN(cfg, wv)
    ds = DeviceDataset(recs, vis)
    s0, e0 = al.labels_from_times(data_old, ds.vlen_h)
    ds.set_labels(s0, e0)
    for r, a, b in zip(recs, s0, e0):
        r['s_ind'], r['e_ind'] = int(a), int(b)

    def batches():
        for lo in range(0, N, 16):
            sel = np.arange(lo, min(N, lo + 16))
            f = ds.assemble(sel, labels=False, min_chars=4)
            yield [recs[i] for i in sel], f['video'], f['video_seq_len'], f['word_ids'], f['char_ids']
    prop0, ious0 = al.infer_trainset(model, batches(), mc_dropout=0.5)
    assert len(prop0) == N and set(prop0[0]) == {'vid', 'duration', 'psuedo_idx', 'sentence', 'v_len', 'prop_idx',
                                                  'prop_logits', 'prop_logits1', 'prop_logits2', 'm_score'}
    T0 = max(recs[i]['v_len'] for i in range(16))
    assert prop0[0]['prop_logits'][0].shape == (T0,) and prop0[0]['m_score'].shape == (T0, 4)
    # the two stochastic passes differ from each other and from the deterministic one
    assert np.abs(prop0[0]['prop_logits1'][0] - prop0[0]['prop_logits2'][0]).max() > 0
    # as-written mode: all three identical
    p_w, _ = al.infer_trainset(model, batches(), mc_dropout=None)
    np.testing.assert_array_equal(p_w[3]['prop_logits1'][0], p_w[3]['prop_logits'][0])
    np.testing.assert_array_equal(p_w[3]['prop_logits'][0], prop0[3]['prop_logits'][0])

    ref_new = A.update_labels(copy.deepcopy(data_old), data_gt, prop0, A.get_coff('charades', 1))
    p_before = model.params.clone()
    new_data, prop1, m = al.run_round(model, ds, copy.deepcopy(data_old), data_gt, prop0, 'charades', 1, epochs=2,
                                      batch_size=16, lr=1e-3, drop_rate=0.2)
    assert [r[2] for r in new_data] == [r[2] for r in ref_new]
    assert [r[4] for r in new_data] == [r[4] for r in ref_new]
    assert m['train_steps'] == 2 * 3 and len(prop1) == N
    assert float((model.params - p_before).abs().max()) > 0
    assert torch.isfinite(model.params).all()
    assert 0.0 <= m['miou'] <= 100.0

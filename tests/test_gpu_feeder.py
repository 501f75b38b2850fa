"""GPU: host-fed training (hual_amd/feeder.py = the reference's feed path, /root/reference/utils/runner_utils.py:139-159, as a pinned-memory
upload pipeline).  Every step of the pipeline must see exactly the batch the synchronous path (Trainer.set_batch: blocking
copies into static buffers) would have fed - slot reuse, the compact slot layout of a changing padded shape, the bfloat16 feed."""
import numpy as np
import pytest
import torch

import parity_util as pu
from oracle import seqpan_ref as R

pytestmark = pytest.mark.gpu
SHAPES = [(6, 24, 7, 5), (6, 18, 6, 4), (4, 24, 9, 6), (6, 24, 7, 5), (5, 21, 6, 5), (6, 18, 6, 4), (6, 24, 7, 5), (2, 12, 4, 4),
          (6, 24, 7, 5), (6, 18, 6, 4)]


def _batches(cfg):
    from hual_amd import data
    out = []
    for i, (B, T, L, C) in enumerate(SHAPES):
        b = R.synthetic_batch(cfg, B, T, L, C, seed=50 + i)
        lens = b['lens'].numpy()
        s = np.minimum(np.asarray(b['s_ind']), lens - 2).clip(0)
        e = np.maximum(np.minimum(np.asarray(b['e_ind']), lens - 1), s)
        y1, y2, mm, ii = data.make_labels(s, e, lens, max_len=T)
        # (float64 / int64 on purpose: what numpy code upstream of a feed_dict typically holds)
        out.append(dict(video=b['video'].numpy(), video_seq_len=lens.astype(np.int64), word_ids=b['word_ids'].numpy().astype(np.int64),
                        char_ids=b['char_ids'].numpy(), y1=np.asarray(y1, dtype=np.float64), y2=y2, match_labels=mm, inner_labels=ii))
    return out


def _sync_path(m, batches, lr, vdt=torch.float32):
    """the synchronous feed: one Trainer.set_batch + step per batch; per-step fetches"""
    from hual_amd.train import Trainer
    tr = Trainer(m, world=1, use_graph=False)
    res = []
    for b in batches:
        tr.set_batch(b['video'], b['video_seq_len'], b['word_ids'], b['char_ids'], b['y1'], b['y2'], b['match_labels'], b['inner_labels'],
                     video_dtype=vdt)
        tr.step(lr=lr, drop_rate=0.0)
        torch.cuda.synchronize()
        res.append((tr.start_index.cpu().numpy().copy(), tr.end_index.cpu().numpy().copy(), tr.loss_terms.cpu().numpy().copy(),
                    tr.start_logits.cpu().numpy().copy()))
    return res


@pytest.mark.parametrize('vdt', [torch.float32, torch.bfloat16])
def test_host_fed_steps_see_the_batches_of_the_synchronous_feed(vdt):
    from hual_amd.feeder import HostFeeder
    from hual_amd.train import Trainer
    cfg, p, wv, _, _ = pu.make_case(B=2, T=12, L=4, C=4, max_vlen=24, vdim=64)
    batches = _batches(cfg)
    ref = _sync_path(pu.hip_model(cfg, p, wv), batches, 0.0, vdt)
    m = pu.hip_model(cfg, p, wv)
    tr = Trainer(m, world=1, use_graph=True)
    fd = HostFeeder(tr, capacity=(6, 24, 9, 6), vdim=64, video_dtype=vdt)
    logits, losses = [], []
    for b in batches:                       # lr 0: every step is a function of its batch alone - the per-step fetches must be the same bits
        fd.feed(b, 0.0, 0.0)
        logits.append(tr.start_logits.clone())
        losses.append(tr.loss_terms.clone())
    spans = fd.collect()
    assert len(spans) == len(batches) and fd.stats['batches'] == len(batches)
    for k, (s, e, lt, sl) in enumerate(ref):
        np.testing.assert_array_equal(spans[k][0], s)
        np.testing.assert_array_equal(spans[k][1], e)
        np.testing.assert_array_equal(logits[k].cpu().numpy(), sl)
        np.testing.assert_allclose(losses[k].cpu().numpy(), lt, rtol=2e-6)      # (loss sums through float atomics)
    # the repeated shapes ran as graphs after their first sightings, nothing was allocated per step
    assert tr.stats['replayed'] + tr.stats['captured'] >= 1 and tr.stats['capture_failed'] == 0


def test_host_fed_epoch_trains_like_the_synchronous_feed():
    """one epoch at lr 1e-4 through the pipeline (two slots, eager first sightings then step graphs) against set_batch + eager steps: the
    same parameters up to the float-atomics noise of the gradient bucket.  (Only ten steps: two runs of the SAME path part ways after
    ~13 steps of this toy problem - a ReLU unit of conv layer 2 flips on 1e-7 of noise, scripts/exp/traj_repro.py.)"""
    from hual_amd.feeder import HostFeeder
    from hual_amd.train import Trainer
    lr = 1e-4
    cfg, p, wv, _, _ = pu.make_case(B=2, T=12, L=4, C=4, max_vlen=24, vdim=64)
    batches = _batches(cfg)
    ma = pu.hip_model(cfg, p, wv)
    ref = _sync_path(ma, batches, lr)
    mb = pu.hip_model(cfg, p, wv)
    fd = HostFeeder(Trainer(mb, world=1, use_graph=True), capacity=(6, 24, 9, 6), vdim=64)
    spans = fd.run_epoch(batches, lr, 0.0)
    d = np.abs(ma.params.detach().cpu().numpy() - mb.params.detach().cpu().numpy())
    assert d.max() <= 2.0 * 3.2 * lr * len(batches), d.max()          # a noise-level gradient decides the direction of an AdamWD update
    assert np.mean(d <= 0.1 * lr) >= 0.99, float(np.mean(d <= 0.1 * lr))
    for k in range(len(batches)):
        np.testing.assert_array_equal(spans[k][0], ref[k][0])
        np.testing.assert_array_equal(spans[k][1], ref[k][1])


def test_producer_writes_into_the_pinned_slot_and_errors_are_loud():
    from hual_amd.feeder import HostFeeder
    from hual_amd.train import Trainer
    cfg, p, wv, _, _ = pu.make_case(B=2, T=12, L=4, C=4, max_vlen=24, vdim=64)
    batches = _batches(cfg)[:4]
    ref = _sync_path(pu.hip_model(cfg, p, wv), batches, 0.0)
    tr = Trainer(pu.hip_model(cfg, p, wv), world=1, use_graph=True)
    fd = HostFeeder(tr, capacity=(6, 24, 9, 6), vdim=64)
    for b in batches:
        B, T, V = b['video'].shape
        v = fd.stage_views(B, T, b['word_ids'].shape[1], b['char_ids'].shape[2])
        for k in v:
            v[k][...] = b[k]                 # the loader's own writes, straight into pinned memory
        fd.submit(0.0, 0.0)
    spans = fd.collect()
    for k, (s, e, _, _) in enumerate(ref):
        np.testing.assert_array_equal(spans[k][0], s)
        np.testing.assert_array_equal(spans[k][1], e)
    # the tuple train_loader.batch_iter() yields (data_loader.py:23-28): records first, then the eight arrays
    b0 = batches[0]
    fd.feed((['record'] * len(b0['video']), b0['video'], b0['video_seq_len'], b0['word_ids'], b0['char_ids'], b0['y1'], b0['y2'],
             b0['match_labels'], b0['inner_labels']), 0.0, 0.0)
    (s0, e0), = fd.collect()
    np.testing.assert_array_equal(s0, ref[0][0])
    np.testing.assert_array_equal(e0, ref[0][1])
    with pytest.raises(AssertionError, match='capacity'):
        fd.stage_views(7, 24, 9, 6)
    bad = dict(batches[0]); bad['video'] = bad['video'][:, :, :32]
    with pytest.raises(ValueError, match='feature width'):
        fd.feed(bad, 0.0, 0.0)
    short = dict(batches[0]); short['video_seq_len'] = short['video_seq_len'] - 1
    with pytest.raises(ValueError, match='max\\(video_seq_len\\)'):
        fd.feed(short, 0.0, 0.0)


def test_records_padded_straight_into_the_pinned_slot_equal_the_loader_batch():
    """HostFeeder.feed_records (the loader's padding written into pinned memory by the staging threads) against
    feed(process_train_batch(...)) - the restated loader, pinned to the reference's by tests/golden/labels.npz: same spans, same logits"""
    import al_synth
    from hual_amd import data, lib
    from hual_amd.feeder import HostFeeder
    from hual_amd.model import SeqPAN
    from hual_amd.train import Trainer
    recs, vis, data_gt, _ = al_synth.make_trainset(40, 10, 64, 24, seed=8)
    for r, g in zip(recs, data_gt):
        r['s_ind'], r['e_ind'] = (int(x) for x in data.time_to_index(g[2][0], g[2][1], r['v_len'], r['duration']))
    cfg = lib.make_cfg(vdim=64, max_vlen=24, num_words=200, num_chars=30)
    wv = np.random.default_rng(1).normal(0, 0.4, size=(198, 300)).astype(np.float32)
    cap = (8, 24, max(len(r['w_ids']) for r in recs), 8)
    out = []
    for mode in ('batch', 'records'):
        m = SeqPAN(cfg, wv)
        tr = Trainer(m, world=1, use_graph=True)
        fd = HostFeeder(tr, capacity=cap, vdim=64)
        logits = []
        for lo in range(0, 40, 8):
            if mode == 'batch':
                b = data.process_train_batch(recs[lo:lo + 8], vis)
                if b['char_ids'].shape[2] < 4:
                    b = data.pad_batch_to(b, b['video'].shape[1], b['word_ids'].shape[1], 4)
                fd.feed(b, 0.0, 0.0)
            else:
                fd.feed_records(recs[lo:lo + 8], vis, 0.0, 0.0)
            logits.append(tr.start_logits.clone())
        out.append((fd.collect(), [x.cpu().numpy() for x in logits]))
    for (sa, ea), (sb, eb) in zip(out[0][0], out[1][0]):
        np.testing.assert_array_equal(sa, sb)
        np.testing.assert_array_equal(ea, eb)
    for a, b in zip(out[0][1], out[1][1]):
        np.testing.assert_array_equal(a, b)

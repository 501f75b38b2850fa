"""GPU: the device-fed epoch loop (Trainer.run_epoch = runner_utils.train_epoch, /root/reference/utils/runner_utils.py:139-159, on a
DeviceDataset): padded shapes that change from batch to batch run in ONE workspace, through per-shape step graphs, with the spans
fetched once per epoch - and give the same losses, spans and parameters as plain eager launches on freshly allocated feeds."""
import numpy as np
import pytest
import torch

import al_synth

pytestmark = pytest.mark.gpu


def _setup(N=96, vdim=64, max_vlen=24, seed=5):
    from hual_amd import al, lib
    from hual_amd.dataset import DeviceDataset
    from hual_amd.model import SeqPAN
    recs, vis, data_gt, data_old = al_synth.make_trainset(N, 16, vdim, max_vlen, seed=seed)
    cfg = lib.make_cfg(vdim=vdim, max_vlen=max_vlen, num_words=200, num_chars=30)
    wv = np.random.default_rng(1).normal(0, 0.4, size=(198, 300)).astype(np.float32)
    ds = DeviceDataset(recs, vis)
    s0, e0 = al.labels_from_times(data_gt, ds.vlen_h)
    ds.set_labels(s0, e0)
    return cfg, wv, ds


def _orders(N, epochs, seed=0):
    g = np.random.default_rng(seed)
    return [g.permutation(N).astype(np.int32) for _ in range(epochs)]


def test_epoch_loop_matches_eager_steps_on_fresh_feeds():
    from hual_amd.model import SeqPAN
    from hual_amd.train import Trainer
    cfg, wv, ds = _setup()
    N, bs, lr, drop, epochs = len(ds), 16, 1e-4, 0.2, 3
    orders = _orders(N, epochs)
    # reference run: every batch on freshly allocated feeds and a workspace of its own shape, eager launches, spans fetched per step
    m0 = SeqPAN(cfg, wv)
    m0.ws_poison = 0xFF
    t0 = Trainer(m0, world=1, use_graph=False)
    ref_spans, ref_loss = [], []
    for order in orders:
        for lo in range(0, N, bs):
            sel = order[lo:lo + bs]
            t0.set_batch_device(ds.assemble(sel, out=None, min_chars=4))
            t0.step(lr=lr, drop_rate=drop)
            ref_spans.append((t0.start_index.cpu().numpy().copy(), t0.end_index.cpu().numpy().copy()))
            ref_loss.append(float(t0.last_loss()))
    # the loop under test
    m1 = SeqPAN(cfg, wv)
    m1.ws_poison = 0xFF                                           # whatever the workspace holds must not matter
    t1 = Trainer(m1, world=1, use_graph=True)
    got = [t1.run_epoch(ds, order, bs, lr=lr, drop_rate=drop, min_chars=4) for order in orders]
    torch.cuda.synchronize()
    nsteps = epochs * ((N + bs - 1) // bs)
    st = t1.stats
    assert st['eager'] + st['captured'] + st['replayed'] == nsteps
    shapes = {ds.batch_shape(order[lo:lo + bs]) for order in orders for lo in range(0, N, bs)}
    assert st['eager'] == len(shapes) and st['replayed'] > 0, (st, len(shapes))
    # the workspace was allocated once, for the largest shape
    assert m1._ws.numel() >= max(m1._ws_need.values())
    k = same = total = 0
    for (s, e), order in zip(got, orders):
        assert len(s) == N and len(e) == N
        for lo in range(0, N, bs):
            n = len(order[lo:lo + bs])
            # float atomics in the weight-gradient launch: steps are not bit-reproducible, and AdamWeightDecay has no bias correction
            # (every element moves ~lr per step whatever its gradient): only the first step is comparable span for span
            if k == 0:
                np.testing.assert_array_equal(s[lo:lo + n], ref_spans[k][0])
                np.testing.assert_array_equal(e[lo:lo + n], ref_spans[k][1])
            same += int(((s[lo:lo + n] == ref_spans[k][0]) & (e[lo:lo + n] == ref_spans[k][1])).sum())
            total += n
            k += 1
    assert same >= 0.8 * total, (same, total)
    assert abs(float(t1.last_loss()) - ref_loss[-1]) <= 2e-2 * abs(ref_loss[-1])
    d = float((m1.params - m0.params).abs().max())
    assert d <= 2.0 * lr * nsteps * 3.2, d                        # both moved, by at most Adam's first-step size per step
    assert torch.isfinite(m1.params).all()


def test_first_step_of_a_shape_is_bit_identical_whatever_ran_before():
    """a step's forward is bit-reproducible: the same batch gives the same loss and logits on a fresh model and on a
    model whose shared workspace has just been used by other shapes and whose step runs as a replayed graph"""
    from hual_amd.model import SeqPAN
    from hual_amd.train import Trainer
    cfg, wv, ds = _setup(N=64)
    sel_a, sel_b = np.arange(0, 16, dtype=np.int32), np.arange(16, 40, dtype=np.int32)
    # fresh model, one eager forward of batch a
    m0 = SeqPAN(cfg, wv)
    t0 = Trainer(m0, world=1, use_graph=False)
    t0.set_batch_device(ds.assemble(sel_a, out=None, min_chars=4))
    opts = t0._opts(0.2, 0, defer_loss=False)                     # a forward on its own: it closes the loss itself
    t0._forward(opts)
    want = (t0.loss_terms.cpu().numpy().copy(), t0.start_logits.cpu().numpy().copy())
    # loop model: lr 0 steps keep the parameters (AdamWD with lr 0 leaves p untouched), the Philox offset is reset for the comparison
    m1 = SeqPAN(cfg, wv)
    m1.ws_poison = 0xFF
    t1 = Trainer(m1, world=1, use_graph=True)
    bufs = ds.feed_buffers(24, min_chars=4)
    t1.reserve(*bufs['shape'])                                    # workspace + fetch tensors for the largest batch: no growth, stable addresses
    for sel in (sel_a, sel_b, sel_a, sel_b):                      # eager a, eager b, capture a, capture b
        t1.set_batch_device(ds.assemble(sel, min_chars=4, buffers=bufs))
        t1.step(lr=0.0, drop_rate=0.2)
    assert (t1.stats['eager'], t1.stats['captured'], t1.stats['replayed']) == (2, 2, 0)
    m1.set_rng(12345, 0)
    t1.set_batch_device(ds.assemble(sel_a, min_chars=4, buffers=bufs))
    t1.step(lr=0.0, drop_rate=0.2)                                # replay of a's graph
    torch.cuda.synchronize()
    assert t1.stats['replayed'] == 1
    np.testing.assert_array_equal(t1.loss_terms.cpu().numpy(), want[0])
    np.testing.assert_array_equal(t1.start_logits.cpu().numpy(), want[1])


def test_an_epoch_of_more_than_64_padded_shapes_replays_bit_identical_to_eager():
    """The reference's own ActivityNet annotations (tests/golden/lengths_anet.npz) give several hundred distinct (L, C) padded shapes
    per epoch (data_loader.py:23-28 + pad_seq / pad_char_seq): far more than the 64-entry cache of round 5.  Every shape keeps its
    own step graph and its own job table; a replayed step gives bit for bit the loss terms, logits and spans of the eager launch."""
    import al_synth
    from hual_amd import al, lib
    from hual_amd.dataset import DeviceDataset
    from hual_amd.model import SeqPAN
    from hual_amd.train import Trainer
    recs, vis, data_gt, _ = al_synth.make_trainset_from_lengths('anet', 480, 64, 24, seed=4, num_words=200, num_chars=30)
    Lm = max(len(r['w_ids']) for r in recs)
    cfg = lib.make_cfg(vdim=64, max_vlen=max(24, Lm), num_words=200, num_chars=30)
    wv = np.random.default_rng(1).normal(0, 0.4, size=(198, 300)).astype(np.float32)
    ds = DeviceDataset(recs, vis)
    s0, e0 = al.labels_from_times(data_gt, ds.vlen_h)
    ds.set_labels(s0, e0)
    bs, N = 4, len(ds)
    batches = [np.arange(lo, min(N, lo + bs), dtype=np.int32) for lo in range(0, N, bs)]
    shapes = {(len(s),) + ds.batch_shape(s) for s in batches}
    assert len(shapes) > 64, len(shapes)
    m = SeqPAN(cfg, wv)
    m.ws_poison = 0xFF
    tr = Trainer(m, world=1, use_graph=True)
    bufs = ds.feed_buffers(bs, min_chars=4)
    tr.reserve(*bufs['shape'])
    p0 = m.params.clone()
    runs = []
    for pas in range(3):                                          # eager / captured / replayed (repeated shapes are ahead by one)
        before = dict(tr.stats)
        out = []
        for k, sel in enumerate(batches):
            m.set_rng(777, k)                                     # the same dropout stream for batch k in every pass
            tr.set_batch_device(ds.assemble(sel, min_chars=4, buffers=bufs))
            tr.step(lr=0.0, drop_rate=0.2)                        # lr 0: AdamWD leaves the parameters where they are
            B, T = tr.shape[0], tr.shape[1]
            out.append((tr.loss_terms.clone(), tr.start_logits.clone(), tr.end_logits.clone(), tr.spans[:, :B].clone()))
        runs.append((out, {k: tr.stats[k] - before[k] for k in before}))
    torch.cuda.synchronize()
    assert runs[0][1]['eager'] == len(shapes) and runs[2][1]['replayed'] == len(batches), [r[1] for r in runs]
    assert tr.stats['evicted'] == 0 and tr.stats['capture_failed'] == 0 and len(tr._cache) == len(shapes)
    assert torch.equal(m.params, p0)
    for a, b in zip(runs[0][0], runs[2][0]):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    assert all(torch.isfinite(o[0]).all() for o in runs[2][0])

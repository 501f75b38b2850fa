"""GPU: the data-parallel EPOCH LOOP (Trainer.run_epoch with world = 2 = the loop of /root/reference/utils/runner_utils.py:139-159 split over
ranks; both ranks on the one GPU of the test box, collectives over gloo staged through the host - RCCL refuses two ranks on one device):
every rank assembles its shard of each global batch on the device, padded to the GLOBAL batch's (T, L, C), with the host-side matching
denominator; step for step the averaged gradient equals the single-process gradient of the global batch, the spans are the
single-process spans, and after two epochs both ranks hold identical parameters next to the single-process run's."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, BS, WORLD, LR = 48, 4, 2, 1e-4


def _setup():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import al_synth
    from hual_amd import al, lib
    from hual_amd.dataset import DeviceDataset
    recs, vis, data_gt, _ = al_synth.make_trainset(N, 12, 64, 24, seed=21)
    cfg = lib.make_cfg(vdim=64, max_vlen=24, num_words=200, num_chars=30)
    wv = np.random.default_rng(1).normal(0, 0.4, size=(198, 300)).astype(np.float32)
    ds = DeviceDataset(recs, vis)
    s0, e0 = al.labels_from_times(data_gt, ds.vlen_h)
    ds.set_labels(s0, e0)
    orders = [np.random.default_rng(3 + k).permutation(N).astype(np.int32) for k in range(2)]
    return cfg, wv, ds, orders


def _run(world, seg=False):
    """per-step averaged gradients + spans at lr 0, then two real epochs; returns what the comparison needs"""
    from hual_amd.model import SeqPAN
    from hual_amd.train import Trainer
    cfg, wv, ds, orders = _setup()
    m = SeqPAN(cfg, wv)
    m.ws_poison = 0xFF
    tr = Trainer(m, world=world, use_graph=(world == 1) or seg)      # seg: three segment graphs per padded shape, collectives eager
    gb = BS * WORLD
    grads, spans = [], []
    for lo in range(0, N, gb):
        st, en = tr.run_epoch(ds, orders[0][lo:lo + gb], BS if world > 1 else gb, lr=0.0, drop_rate=0.0, min_chars=4)
        torch.cuda.synchronize()
        grads.append(m.grads.detach().cpu().numpy() / world)
        spans.append((st.copy(), en.copy()))
    ep_spans = []
    for order in orders:
        ep_spans.append(tr.run_epoch(ds, order, BS if world > 1 else gb, lr=LR, drop_rate=0.0, min_chars=4))
    torch.cuda.synchronize()
    return grads, spans, m.params.detach().cpu().numpy(), ep_spans, dict(tr.stats), tr.last_epoch_ids.copy()


def _worker(rank, world, port, q, seg=False):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    res = _run(world, seg)
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('seg', [False, True])
def test_two_rank_epoch_loop_matches_single_process_global_batches(seg):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29650 + (os.getpid() % 120) + (5 if seg else 0)
    procs = [ctx.Process(target=_worker, args=(r, WORLD, port, q, seg)) for r in range(WORLD)]
    for pr in procs:
        pr.start()
    got = dict(q.get(timeout=600) for _ in range(WORLD))
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    g1, s1, p1, e1, st1, ids1 = _run(1)
    cfg, wv, ds, orders = _setup()
    shapes = {ds.batch_shape(orders[0][lo:lo + BS * WORLD]) for lo in range(0, N, BS * WORLD)}
    assert len(shapes) >= 3                                        # the global batches have different padded shapes
    # some shard lacks its global batch's longest clip: it was padded to the GLOBAL T
    assert any(ds.vlen_h[orders[0][lo:lo + BS]].max() != ds.vlen_h[orders[0][lo:lo + BS * WORLD]].max() or
               ds.vlen_h[orders[0][lo + BS:lo + 2 * BS]].max() != ds.vlen_h[orders[0][lo:lo + BS * WORLD]].max()
               for lo in range(0, N, BS * WORLD))
    for r in range(WORLD):
        g, s, p, e, st, ids = got[r]
        nst = len(g1) + 2 * (N // (BS * WORLD))
        assert st['eager'] + st['captured'] + st['replayed'] == nst and st.get('dropped', 0) == 0
        assert (st['eager'] == nst) if not seg else (st['replayed'] > 0 and st['capture_failed'] == 0), st
        assert np.array_equal(ids, ids1)
        for k in range(len(g1)):
            scale = max(1.0, float(np.abs(g1[k]).max()))
            assert np.abs(g[k] - g1[k]).max() <= 2e-4 * scale, (r, k)
            # spans of the WHOLE global batch on every rank, in the single-process order
            np.testing.assert_array_equal(s[k][0], s1[k][0])
            np.testing.assert_array_equal(s[k][1], s1[k][1])
        assert len(e[1][0]) == N
    assert np.array_equal(got[0][2], got[1][2])                    # replicas stay identical
    nsteps = 2 * (N // (BS * WORLD))
    d = np.abs(got[0][2] - p1)
    # AdamWeightDecay has no bias correction: an element whose gradient is rounding noise moves ~3.2 lr per step in a direction the
    # noise decides, so single elements may differ by that much; the bulk agrees far below one step
    assert d.max() <= 2.0 * 3.2 * LR * nsteps, d.max()
    assert np.mean(d <= 0.1 * LR) >= 0.9, float(np.mean(d <= 0.1 * LR))
    # first epoch's spans: the same predictions (parameters agree to ~1e-5 by then; allow a few near-tie flips)
    same = np.mean((got[0][3][0][0] == e1[0][0]) & (got[0][3][0][1] == e1[0][1]))
    assert same >= 0.9, same


# ---------------------------------------------------------------------------------------------------------------------
# infer_trainset and one whole active-learning round, data parallel (runner_utils.py:69-110, run_charades.py:9-41 over two ranks)
def _al_setup():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import al_synth
    from hual_amd import al, lib
    from hual_amd.dataset import DeviceDataset
    recs, vis, data_gt, data_old = al_synth.make_trainset(40, 10, 64, 24, seed=8)
    cfg = lib.make_cfg(vdim=64, max_vlen=24, num_words=200, num_chars=30)
    wv = np.random.default_rng(1).normal(0, 0.4, size=(198, 300)).astype(np.float32)
    ds = DeviceDataset(recs, vis)
    s0, e0 = al.labels_from_times(data_old, ds.vlen_h)
    ds.set_labels(s0, e0)
    for r, a, b in zip(recs, s0, e0):
        r['s_ind'], r['e_ind'] = int(a), int(b)
    return cfg, wv, ds, data_gt, data_old


def _al_run(world):
    from hual_amd import al
    from hual_amd.model import SeqPAN
    cfg, wv, ds, data_gt, data_old = _al_setup()
    m = SeqPAN(cfg, wv)
    recs, ious = al.infer_trainset_sharded(m, ds, 6, mc_dropout=0.5)
    rng_after = m.rng_state.cpu().numpy().copy()
    out = None
    if recs is not None:
        out = dict(n=len(recs), idx=np.array([r['prop_idx'] for r in recs]), vids=[r['vid'] for r in recs],
                   l0=[np.stack(r['prop_logits']) for r in recs], l1=[np.stack(r['prop_logits1']) for r in recs],
                   l2=[np.stack(r['prop_logits2']) for r in recs], ious=np.array(ious))
    # one round: rank 0 renews the labels from its records, everybody trains one epoch data parallel, inference is sharded again
    new_data, recs2, met = al.run_round(m, ds, data_old, data_gt, recs, 'charades', 1, epochs=1, batch_size=6 // world if world > 1 else 6,
                                        lr=1e-4, drop_rate=0.2, mc_dropout=0.5)
    torch.cuda.synchronize()
    return out, rng_after, [list(map(float, r[2])) for r in new_data], None if recs2 is None else len(recs2), met, m.params.detach().cpu().numpy()


def _al_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    q.put((rank, _al_run(world)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_infer_trainset_and_al_round():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29520 + (os.getpid() % 100)
    procs = [ctx.Process(target=_al_worker, args=(r, WORLD, port, q)) for r in range(WORLD)]
    for pr in procs:
        pr.start()
    got = dict(q.get(timeout=600) for _ in range(WORLD))
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    o1, rng1, new1, n1, met1, p1 = _al_run(1)
    o0, rng0, new0, n0, met0, p0 = got[0]
    assert got[1][0] is None and got[1][3] is None               # records live on rank 0 only
    # the sharded pass returns the single-process records: same order, same spans, the same BITS in all three pairs of logits
    assert o0['n'] == o1['n'] == 40 and o0['vids'] == o1['vids']
    np.testing.assert_array_equal(o0['idx'], o1['idx'])
    np.testing.assert_array_equal(o0['ious'], o1['ious'])
    for k in ('l0', 'l1', 'l2'):
        for a, b in zip(o0[k], o1[k]):
            np.testing.assert_array_equal(a, b)
    assert not np.array_equal(np.concatenate([x.ravel() for x in o0['l1']]), np.concatenate([x.ravel() for x in o0['l0']]))   # dropout was on
    np.testing.assert_array_equal(rng0, got[1][1])               # both ranks left the pass with their stream at the same place
    np.testing.assert_array_equal(rng0, rng1)
    # the round: the same renewed labels everywhere (computed on rank 0, broadcast), replicas identical, all records back on rank 0
    assert new0 == got[1][2] == new1
    assert n0 == 40 and met0['world'] == 2 and met0['train_steps'] == (40 + 5) // 6
    assert np.array_equal(p0, got[1][5])
    for k in ('r1i3', 'r1i5', 'r1i7', 'miou'):
        assert met0[k] == got[1][4][k]

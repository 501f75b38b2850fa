"""Data-parallel exactness on CPU: 2 gloo processes x B clips reproduce the gradient of the 2B-clip
batch.  The compute engine here is the CPU oracle (allowed in tests); what is under test is the decomposition that
hual_amd/train.py uses on GPUs: flat-bucket all-reduce + 1/world prescale, global matching denominator,
all-gathered alignment features with local gradient rows scaled by world (hual_amd/dist.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    import parity_util as pu
    from hual_amd import dist as hdist
    from oracle import seqpan_ref as R
    cfg, p, wv, b, labels = pu.make_case(B=4, T=14, L=5, C=4, seed=31, max_vlen=16)
    B = 4 // world
    sl = slice(rank * B, (rank + 1) * B)
    # every rank pads to the GLOBAL T here so the shards line up with the single-process reference
    lens = b['lens']
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    denom = hdist.match_denominator(int(lens[sl].sum()), torch.device('cpu'))
    feats = {}

    def align_override(t_hat, v_hat):
        ta, va = hdist.gather_features(t_hat.detach(), v_hat.detach())
        ta = ta.clone(); va = va.clone()
        ta[sl] = t_hat; va[sl] = v_hat                      # keep the graph for the local rows
        feats['n'] = ta.shape[0]
        return R.align_loss_from_pooled(ta, va) * world     # gradient rows of local samples scaled by world
    video = b['video'][sl]
    out = R.forward(pr, cfg, wv, video, lens[sl], b['word_ids'][sl], b['char_ids'][sl], labels=tuple(x[sl] for x in labels),
                    align_override=align_override, match_denom=denom)
    names = list(pr.keys())
    gl = torch.autograd.grad(out['loss'], [pr[k] for k in names], allow_unused=True)
    flat = torch.cat([(g if g is not None else torch.zeros_like(pr[k])).reshape(-1) for k, g in zip(names, gl)])
    hdist.allreduce_sum_(flat)
    flat = flat / world
    if rank == 0:
        q.put((names, [tuple(pr[k].shape) for k in names], flat.numpy(), feats['n']))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_equals_single_process():
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import parity_util as pu
    from oracle import seqpan_ref as R
    # this shard construction needs every shard to contain a full-length clip (T = max len), make_case guarantees
    # only one -> build the case, then force clip 0 and clip 2 to full length
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr_ in procs:
        pr_.start()
    names, shapes, flat, n = q.get(timeout=300)
    for pr_ in procs:
        pr_.join(timeout=120)
        assert pr_.exitcode == 0
    assert n == 4
    cfg, p, wv, b, labels = pu.make_case(B=4, T=14, L=5, C=4, seed=31, max_vlen=16)
    ref_out, ref_grads = pu.oracle_run(cfg, p, wv, b, labels)
    off = 0
    for k, sh in zip(names, shapes):
        sz = int(np.prod(sh))
        g = flat[off:off + sz].reshape(sh)
        off += sz
        r = ref_grads[k].numpy()
        assert np.abs(g - r).max() <= 2e-4 * max(1.0, np.abs(r).max()), k


def _shape_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from hual_amd import dist as hdist
    res = []
    # (longest local clip, padded T) per rank: consistent; rank 1 padded further than rank 0; nobody reaches the padded length
    for longest, T in (((9, 14), (14, 14)), ((14, 14), (12, 16)), ((10, 14), (11, 14))):
        try:
            res.append(('ok', hdist.check_padded_length(longest[rank], T[rank])))
        except ValueError:
            res.append(('raised', None))
    res.append(('min', hdist.global_min(1 if rank == 0 else 0)))
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_shape_check_and_graph_decision_are_collective():
    """hual_amd/dist.py check_padded_length / global_min with 2 gloo ranks: a shape mismatch raises on BOTH ranks (neither is left
    waiting in the next collective), and the graph-or-eager decision of Trainer._step_dp is the minimum over ranks"""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29100 + (os.getpid() % 300)
    procs = [ctx.Process(target=_shape_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr_ in procs:
        pr_.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for pr_ in procs:
        pr_.join(timeout=60)
        assert pr_.exitcode == 0
    for r in range(2):
        assert got[r] == [('ok', 14), ('raised', None), ('raised', None), ('min', 0)], got


# ---------------------------------------------------------------------------------------------------------------------
# the data-parallel EPOCH LOOP (Trainer.run_epoch with world > 1): plan, padding and denominators (hual_amd/dist.py shard_plan)
def test_shard_plan_covers_the_epoch_and_pads_to_the_global_batch():
    from hual_amd import dist as hdist
    g = np.random.default_rng(5)
    N = 203
    vlen, nw, mc = g.integers(4, 65, N), g.integers(3, 40, N), g.integers(1, 20, N)
    order = g.permutation(N).astype(np.int32)
    # one rank: the reference's batching (data_loader.py:23-28), ragged last batch kept, nothing dropped
    steps, dropped = hdist.shard_plan(order, 16, 1, vlen, nw, mc, min_chars=4)
    assert dropped == 0 and len(steps) == (N + 15) // 16 and steps[-1]['B'] == N % 16
    assert np.array_equal(np.concatenate([s['ids'] for s in steps]), order)
    for s in steps:
        ids = s['ids']
        assert s['shape'] == (vlen[ids].max(), nw[ids].max(), max(4, mc[ids].max())) and s['frames'] == vlen[ids].sum()
    # four ranks of 16: global batches of 64, the tail split evenly, < world clips dropped, every shard the same size
    steps, dropped = hdist.shard_plan(order, 16, 4, vlen, nw, mc, min_chars=4)
    tail = N % 64
    assert dropped == tail % 4 and steps[-1]['B'] == tail // 4 and all(s['B'] == 16 for s in steps[:-1])
    kept = np.concatenate([s['ids'] for s in steps])
    assert np.array_equal(kept, order[:N - dropped])
    for s in steps:
        assert len(s['ids']) == 4 * s['B']
        assert s['shape'][0] == vlen[s['ids']].max()          # the GLOBAL batch's longest clip, whichever shard holds it
    # fewer clips left than ranks: no step at all for them
    steps, dropped = hdist.shard_plan(order[:66], 16, 4, vlen, nw, mc)
    assert len(steps) == 1 and dropped == 2


def _epoch_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    from hual_amd import data
    from hual_amd import dist as hdist
    from oracle import seqpan_ref as R
    cfg, p, wv, recs, vis, order = _epoch_case()
    vlen = np.array([r['v_len'] for r in recs]); nw = np.array([len(r['w_ids']) for r in recs])
    mc = np.array([max(len(c) for c in r['c_ids']) for r in recs])
    steps, _ = hdist.shard_plan(order, 3, world, vlen, nw, mc, min_chars=4)
    res = []
    for st in steps:
        B = st['B']
        sel = st['ids'][rank * B:(rank + 1) * B]
        T, L, C = st['shape']
        b = data.pad_batch_to(data.process_train_batch([recs[i] for i in sel], vis), T, L, C)       # shard padded to the GLOBAL shape
        pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        denom = (st['frames'] + 1e-12) / world                     # host-side, as Trainer.run_epoch computes it
        sl = slice(rank * B, (rank + 1) * B)

        def align_override(t_hat, v_hat):
            ta, va = hdist.gather_features(t_hat.detach(), v_hat.detach())
            ta = ta.clone(); va = va.clone()
            ta[sl] = t_hat; va[sl] = v_hat
            return R.align_loss_from_pooled(ta, va) * world
        out = R.forward(pr, cfg, wv, torch.tensor(b['video']), torch.tensor(b['video_seq_len']), torch.tensor(b['word_ids']),
                        torch.tensor(b['char_ids']),
                        labels=(torch.tensor(b['y1']), torch.tensor(b['y2']), torch.tensor(b['match_labels']),
                                torch.tensor(b['inner_labels'], dtype=torch.float32)),
                        align_override=align_override, match_denom=denom, shard_of_longer_batch=True)
        names = list(pr.keys())
        gl = torch.autograd.grad(out['loss'], [pr[k] for k in names], allow_unused=True)
        flat = torch.cat([(g if g is not None else torch.zeros_like(pr[k])).reshape(-1) for k, g in zip(names, gl)])
        hdist.allreduce_sum_(flat)
        res.append((flat / world).numpy())
    if rank == 0:
        q.put(res)
    dist.barrier()
    dist.destroy_process_group()


def _epoch_case():
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import al_synth
    import parity_util as pu
    from hual_amd import al
    recs, vis, data_gt, _ = al_synth.make_trainset(12, 6, 32, 16, seed=9, num_words=60, num_chars=30, max_words=7)
    vlen = np.array([r['v_len'] for r in recs])
    s, e = al.labels_from_times(data_gt, vlen)
    for r, a, b in zip(recs, s, e):
        r['s_ind'], r['e_ind'] = int(a), int(b)
    cfg, p, wv, _, _ = pu.make_case(B=2, T=8, L=4, C=4, seed=1, max_vlen=16, num_words=60, vdim=32)
    order = np.random.default_rng(2).permutation(12).astype(np.int32)
    return cfg, p, wv, recs, vis, order


def test_two_rank_epoch_steps_equal_the_single_process_global_batches():
    """two gloo ranks walk shard_plan's steps of a ragged synthetic set on the oracle engine - every shard padded to the GLOBAL
    batch's (T, L, C) with data.pad_batch_to, host-side matching denominator, gathered alignment features - and reproduce, step
    for step, the gradient of the single-process run on the global batches (hual_amd/data.py process_train_batch = the reference
    loader, pinned by tests/golden/labels.npz)"""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import parity_util as pu
    from hual_amd import data
    from hual_amd import dist as hdist
    from oracle import seqpan_ref as R
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29300 + (os.getpid() % 200)
    procs = [ctx.Process(target=_epoch_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr_ in procs:
        pr_.start()
    got = q.get(timeout=300)
    for pr_ in procs:
        pr_.join(timeout=120)
        assert pr_.exitcode == 0
    cfg, p, wv, recs, vis, order = _epoch_case()
    vlen = np.array([r['v_len'] for r in recs]); nw = np.array([len(r['w_ids']) for r in recs])
    mc = np.array([max(len(c) for c in r['c_ids']) for r in recs])
    steps, dropped = hdist.shard_plan(order, 3, 2, vlen, nw, mc, min_chars=4)
    assert len(steps) == 2 and dropped == 0 and len(got) == 2
    assert len({s['shape'] for s in steps}) == 2                   # the two global batches have different padded shapes
    for st, flat in zip(steps, got):
        b = data.process_train_batch([recs[i] for i in st['ids']], vis)
        T, L, C = st['shape']
        b = data.pad_batch_to(b, T, L, C)                          # (only C can grow: min_chars)
        bb = dict(video=torch.tensor(b['video']), lens=torch.tensor(b['video_seq_len']), word_ids=torch.tensor(b['word_ids']),
                  char_ids=torch.tensor(b['char_ids']))
        labels = (torch.tensor(b['y1']), torch.tensor(b['y2']), torch.tensor(b['match_labels']),
                  torch.tensor(b['inner_labels'], dtype=torch.float32))
        _, ref = pu.oracle_run(cfg, p, wv, bb, labels)
        off = 0
        for k in ref:
            sz = ref[k].numel()
            g = flat[off:off + sz].reshape(tuple(ref[k].shape))
            off += sz
            r = ref[k].numpy()
            assert np.abs(g - r).max() <= 2e-4 * max(1.0, np.abs(r).max()), k

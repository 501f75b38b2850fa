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

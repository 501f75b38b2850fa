"""GPU: the data-parallel step with TWO ranks (both on the one GPU of the test box, collectives over gloo staged through the
host - RCCL refuses two ranks on one device): after one step the summed gradient bucket / 2 must equal the single-process
gradient of the 4-clip batch, and both ranks must hold identical parameters."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _case(lens=(18, 11, 18, 7)):
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import parity_util as pu
    from hual_amd import data
    cfg, p, wv, b, labels = pu.make_case(B=4, T=18, L=6, C=5, seed=33, max_vlen=24)
    lens = np.array(lens, dtype=np.int32)
    b['lens'] = torch.tensor(lens)
    for k in range(4):
        b['video'][k, lens[k]:] = 0.0
    s = np.array([1, 2, 3, 1]); e = np.minimum(np.array([15, 8, 12, 5]), lens - 2)
    y1, y2, mm, ii = data.make_labels(s, e, lens, max_len=18)
    labels = (torch.tensor(y1), torch.tensor(y2), torch.tensor(mm), torch.tensor(ii, dtype=torch.float32))
    return cfg, p, wv, b, labels


def _worker(rank, world, port, q, lens, seg=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import parity_util as pu
    from hual_amd.train import Trainer
    cfg, p, wv, b, labels = _case(lens)
    B = 4 // world
    sl = slice(rank * B, (rank + 1) * B)
    m = pu.hip_model(cfg, p, wv)
    if seg:
        os.environ['HUAL_DP_GRAPH'] = 'seg'
    tr = Trainer(m, world=world, use_graph=seg)
    tr.set_batch(b['video'][sl].numpy(), b['lens'][sl].numpy(), b['word_ids'][sl].numpy(), b['char_ids'][sl].numpy(),
                 *[x[sl].numpy() for x in labels])
    tr.step(lr=1e-3, drop_rate=0.0)
    torch.cuda.synchronize()
    if seg:
        assert tr.dp_launch.startswith('three hipGraphs'), tr.dp_launch
    q.put((rank, m.grads.detach().cpu().numpy() / world, m.params.detach().cpu().numpy(), float(tr.last_loss())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('lens,seg', [((18, 11, 18, 7), False), ((18, 11, 9, 7), False), ((18, 11, 9, 7), True)])
def test_two_ranks_on_one_gpu_match_single_process(lens, seg):
    """lens (18, 11, 9, 7): the second shard holds no full-length clip; it is padded to the GLOBAL T = 18 (Trainer.set_batch,
    data-parallel mode), which is what keeps the decomposition exact - with a shard-local T the reference's unmasked
    conv_block (modules.py:59-70) would see different rows behind each clip's end.
    seg: the step as three hipGraphs of the launches with the two collectives eager between them (Trainer._step_dp, the default launch
    mode with more than one rank)"""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import parity_util as pu
    from hual_amd.train import Trainer
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29800 + (os.getpid() % 150) + (7 if lens[2] == 9 else 0) + (13 if seg else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, lens, seg)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = dict()
    for _ in range(2):
        rank, g, par, loss = q.get(timeout=300)
        got[rank] = (g, par, loss)
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    cfg, p, wv, b, labels = _case(lens)
    m = pu.hip_model(cfg, p, wv)
    tr = Trainer(m, world=1, use_graph=False)
    tr.set_batch(b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy(), *[x.numpy() for x in labels])
    tr.step(lr=1e-3, drop_rate=0.0)
    torch.cuda.synchronize()
    g1 = m.grads.detach().cpu().numpy()
    p1 = m.params.detach().cpu().numpy()
    scale = max(1.0, float(np.abs(g1).max()))
    assert np.abs(got[0][0] - g1).max() <= 2e-4 * scale
    assert np.array_equal(got[0][1], got[1][1])                # replicas stay identical
    assert np.abs(got[0][1] - p1).max() < 2.5e-3               # one Adam step at lr 1e-3 (sign-like first update, ~3.2e-3 max)

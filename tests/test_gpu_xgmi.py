"""GPU: the one-shot peer-mapped all-reduce (csrc/xgmi.hip, hual_amd/xgmi.py) with TWO processes on the one GPU of the
test box - IPC mappings of the same device work where RCCL refuses two ranks - against the host-staged sum over gloo: bit-equal, call
after call, and as the gradient exchange of the data-parallel train step."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    os.environ['HUAL_ALLREDUCE'] = 'custom'
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from hual_amd import dist as hdist
    from hual_amd.xgmi import OneShotAllReduce
    dev = torch.device('cuda:0')
    n = 1186508 + 4                                              # the Charades bucket, padded to a multiple of 4
    n -= n % 4
    g = torch.Generator(device='cpu').manual_seed(100 + rank)
    flat = torch.randn(n, generator=g).to(dev)
    ar = OneShotAllReduce(flat)
    ok = True
    for it in range(4):
        want = flat.cpu()
        dist.all_reduce(want)                                    # the host-staged sum (what hual_amd.dist does over gloo)
        ar()
        torch.cuda.synchronize()
        ar.check()
        ok = ok and torch.equal(flat.cpu(), want)
        flat.mul_(0.5).add_(float(rank + it))                    # another bucket for the next call
    # captured into a hipGraph and replayed: the sequence number lives on the device and advances with every replay.  After the first
    # call both ranks hold the same sum, so every further call doubles it exactly
    flat.copy_(torch.randn(n, generator=g).to(dev))
    want = flat.cpu()
    dist.all_reduce(want)
    ar()                                                         # (eager warm call: both ranks at the same sum)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        gr.capture_begin()
        ar(stream=st)
        gr.capture_end()
    torch.cuda.current_stream().wait_stream(st)
    dist.barrier()
    for _ in range(8):
        gr.replay()
    torch.cuda.synchronize()
    ar.check()
    ok = ok and torch.equal(flat.cpu(), want * float(world ** 8))
    # the same through the data-parallel train step: HUAL_ALLREDUCE=custom routes hual_amd.dist.allreduce_sum_(model.grads) here
    import test_gpu_dp2 as t2
    import parity_util as pu
    from hual_amd.train import Trainer
    cfg, p, wv, b, labels = t2._case()
    B = 4 // world
    sl = slice(rank * B, (rank + 1) * B)
    m = pu.hip_model(cfg, p, wv)
    tr = Trainer(m, world=world, use_graph=True)             # the default launch with more than one rank: segment graphs, collectives eager
    used = hdist._custom.get(m.grads.data_ptr()) is not None
    tr.set_batch(b['video'][sl].numpy(), b['lens'][sl].numpy(), b['word_ids'][sl].numpy(), b['char_ids'][sl].numpy(),
                 *[x[sl].numpy() for x in labels])
    tr.step(lr=1e-3, drop_rate=0.0)
    torch.cuda.synchronize()
    assert tr.dp_launch.startswith('three hipGraphs'), tr.dp_launch
    hdist._custom[m.grads.data_ptr()].check()
    q.put((rank, ok, used, m.grads.detach().cpu().numpy() / world, m.params.detach().cpu().numpy()))
    ar.close()
    hdist.disable_custom_allreduce()
    dist.barrier()
    dist.destroy_process_group()


def test_one_shot_allreduce_two_processes_one_gpu():
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import parity_util as pu
    import test_gpu_dp2 as t2
    from hual_amd.train import Trainer
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29400 + (os.getpid() % 90)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = {}
    for _ in range(2):
        r = q.get(timeout=300)
        got[r[0]] = r[1:]
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    assert got[0][0] and got[1][0]                               # four calls, bit-equal to the host-staged sum on both ranks
    assert got[0][1] and got[1][1]                               # the trainer's bucket was routed through the custom collective
    assert np.array_equal(got[0][2], got[1][2]) and np.array_equal(got[0][3], got[1][3])      # identical bits on both ranks
    cfg, p, wv, b, labels = t2._case()
    m = pu.hip_model(cfg, p, wv)
    tr = Trainer(m, world=1, use_graph=False)
    tr.set_batch(b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy(), *[x.numpy() for x in labels])
    tr.step(lr=1e-3, drop_rate=0.0)
    torch.cuda.synchronize()
    g1 = m.grads.detach().cpu().numpy()
    assert np.abs(got[0][2] - g1).max() <= 2e-4 * max(1.0, float(np.abs(g1).max()))

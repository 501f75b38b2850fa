"""GPU: the data-parallel step (all_gather of alignment features, external alignment loss, flat all-reduce, prescaled
AdamWD) on a 1-rank RCCL process group must reproduce the single-process step exactly in structure and to rounding in
value.  (Multi-rank exactness of the decomposition itself is proven on CPU with gloo in tests/test_dp_gloo.py.)"""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist

import parity_util as pu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def rccl_group():
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(29600 + os.getpid() % 300))
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
    yield
    dist.destroy_process_group()


def _dp_run(monkeypatch, graph, steps=3, sync_check=False):
    """the data-parallel step on the 1-rank RCCL group with the collectives really issued (HUAL_DP_FORCE_COLLECTIVES)"""
    from hual_amd.train import Trainer
    monkeypatch.setenv('HUAL_DP_FORCE_COLLECTIVES', '1')
    monkeypatch.setenv('HUAL_DP_GRAPH', '1' if graph else '0')
    cfg, p, wv, b, labels = pu.make_case(B=4, T=24, L=7, C=5, seed=21)
    feeds = (b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy(), *[x.numpy() for x in labels])
    m = pu.hip_model(cfg, p, wv)
    m.set_rng(7, 3)
    tr = Trainer(m, world=1, use_graph=graph, force_dp=True)      # graph: the default with the nccl backend
    tr.set_batch(*feeds)
    tr.step(lr=1e-4, drop_rate=0.2)                  # first step: allocations, RCCL channel setup, (graph capture)
    torch.cuda.synchronize()
    if sync_check:
        torch.cuda.set_sync_debug_mode('error')      # any synchronising torch call (.item(), host copies) raises from here on
    try:
        for _ in range(steps - 1):
            tr.step(lr=1e-4, drop_rate=0.2)
    finally:
        if sync_check:
            torch.cuda.set_sync_debug_mode('default')
    torch.cuda.synchronize()
    return float(tr.last_loss()), m.params.detach().cpu().numpy().copy(), tr


def test_dp_step_enqueues_without_host_synchronisation(rccl_group, monkeypatch):
    """steady-state data-parallel steps (RCCL all-gather + all-reduce included) must not synchronise the host: the
    matching-loss denominator is a device scalar, the alignment gradient rows go straight into the workspace"""
    loss, params, tr = _dp_run(monkeypatch, graph=False, steps=4, sync_check=True)
    assert np.isfinite(loss) and np.isfinite(params).all()
    assert tr.graph is None


def test_dp_step_captured_as_graph_matches_eager(rccl_group, monkeypatch):
    """the default with the nccl backend: the whole data-parallel step with its RCCL collectives replayed as one hipGraph"""
    l0, p0, _ = _dp_run(monkeypatch, graph=False)
    l1, p1, tr = _dp_run(monkeypatch, graph=True, sync_check=True)
    assert tr.graph is not None
    np.testing.assert_allclose(l0, l1, rtol=2e-4, atol=2e-4)
    assert np.abs(p0 - p1).max() < 5e-4      # three Adam steps without bias correction move every weight by ~1e-3


def test_dp_path_on_one_rank_matches_single_path(rccl_group):
    """single path vs data-parallel path (1-rank RCCL group), two steps.  Step 2 restarts BOTH paths from the single path's
    parameters, Adam slots and dropout counter: the forwards are then identical (no atomics in the forward: same ReLU sets, same
    masks) and the gradients are compared PER TENSOR, element by element - a wrong small tensor (a bias, a layer-norm or
    label-embedding gradient, a missing column part of d v_hat) cannot hide in a global norm."""
    from hual_amd.train import Trainer
    cfg, p, wv, b, labels = pu.make_case(B=4, T=24, L=7, C=5, seed=21)
    feeds = (b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy(),
             *[x.numpy() for x in labels])
    ms, trs = [], []
    for force_dp in (False, True):
        m = pu.hip_model(cfg, p, wv)
        m.set_rng(7, 3)
        tr = Trainer(m, world=1, use_graph=False, force_dp=force_dp)
        tr.set_batch(*feeds)
        ms.append(m)
        trs.append(tr)

    def per_tensor(step):
        g0, g1 = ms[0].grads_dict(), ms[1].grads_dict()
        gmax = max(float(np.abs(v).max()) for v in g0.values())
        for k in g0:
            # (floor: a tensor whose gradient is mathematically zero - the key biases, softmax is invariant to them - holds rounding
            #  noise of the products it is the difference of; 1e-3 of the largest gradient of the step is far below any real entry)
            scale = max(float(np.abs(g0[k]).max()), 1e-3 * gmax)
            d = float(np.abs(g0[k] - g1[k]).max())
            # the two paths differ by the summation order of the alignment loss and of the float atomics of the weight gradients
            assert d <= 2e-4 * scale, (step, k, d, scale)

    for step in range(2):
        if step == 1:                                # same state on both sides before step 2
            for a, c in zip((ms[0].params, ms[0].adam_m, ms[0].adam_v, ms[0].rng_state),
                            (ms[1].params, ms[1].adam_m, ms[1].adam_v, ms[1].rng_state)):
                c.copy_(a)
        losses = []
        for tr in trs:
            tr.step(lr=1e-4, drop_rate=0.2)
            losses.append(float(tr.last_loss()))
        torch.cuda.synchronize()
        np.testing.assert_allclose(losses[0], losses[1], rtol=1e-5, atol=1e-5)
        per_tensor(step)
    p0, p1 = ms[0].params.detach().cpu().numpy(), ms[1].params.detach().cpu().numpy()
    assert np.abs(p0 - p1).max() < 5e-4      # one Adam step without bias correction moves every weight by ~3e-4 at lr 1e-4


def test_bench_multi_rank_path_on_one_rank(tmp_path):
    """bench.py's N > 1 code path - RCCL process group, data-parallel step (captured with its collectives), the timed all-reduce
    of the gradient bucket, MAX-reduce of the elapsed time - on a 1-rank nccl group (HUAL_DP_FORCE_COLLECTIVES=1)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HUAL_DP_FORCE_COLLECTIVES='1', WORLD_SIZE='1', RANK='0', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(29900 + os.getpid() % 90))
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--batch', '8', '--T', '32', '--L', '8', '--C', '5',
                        '--vdim', '256', '--steps', '4', '--warmup', '1', '--prewarm', '2', '--no-cpu-baseline', '--no-roofline',
                        '--epoch-samples', '256', '--anet-samples', '512'],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    out = json.loads(line)
    assert out['n_gpus'] == 1 and out['value'] > 0 and out['steps'] == 4
    assert out['config']['parallelism'] == 'dp1'
    rc = out['rccl']
    assert rc['rccl_ranks'] == 1 and rc['backend'] == 'nccl' and rc['allreduce_us'] > 0
    assert rc['step_launch'].startswith('hipGraph')          # the captured data-parallel step is the default with nccl
    assert np.isfinite(out['config']['final_loss'])
    # the epoch-loop legs ran on the data-parallel code path too (Trainer.run_epoch: shard plan, host-side denominators, eager steps
    # with the collectives on the stream), incl. the reference's ActivityNet length distribution
    el = out['epoch_loop']
    assert 'error' not in el and el['step_launch_modes']['eager'] == el['steps'] and el['value'] > 0
    for leg in out['epoch_loop_anet']:
        assert 'error' not in leg and leg['lengths_from'] == 'anet' and leg['step_launch_modes']['eager'] == leg['steps']


def test_bench_line_survives_a_leg_that_does_not_finish():
    """the legs after the timed region (roofline, CPU baseline, epoch loops, the all-reduce figures) run under a deadline: when it
    passes, rank 0 prints the line with the legs that did finish and every rank leaves with exit code 0"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HUAL_BENCH_LEG_DEADLINE_S='2')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--batch', '8', '--T', '32', '--L', '8', '--C', '5', '--vdim', '256',
                        '--steps', '4', '--warmup', '1', '--prewarm', '2'], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out['value'] > 0 and out['steps'] == 4 and 'legs_cut_short' in out      # (the CPU baseline alone takes ~20 s)
    assert 'leg deadline reached' in r.stderr


def test_bench_two_ranks_rehearsal_on_one_gpu():
    """bench.py --gpus 2 as TWO processes on the one GPU of the test box (HUAL_BENCH_ONE_DEVICE=1, collectives over gloo): the N > 1
    code path that a one-rank group cannot reach - the data-parallel step between two real ranks, the guarded one-shot all-reduce leg
    (hual_amd/xgmi.py: peer mappings, first call compared with the backend's sum, then timed), the data-parallel epoch-loop legs with
    their shard plans, rank 0 alone in the roofline leg while rank 1 waits in the next collective"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = str(29700 + os.getpid() % 90)
    procs = []
    for r in range(2):
        env = dict(os.environ, HUAL_BENCH_ONE_DEVICE='1', HUAL_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0', WORLD_SIZE='2',
                   RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1', MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--batch', '8', '--T', '32', '--L', '8',
                                       '--C', '5', '--vdim', '256', '--steps', '4', '--warmup', '1', '--prewarm', '2', '--no-cpu-baseline',
                                       '--epoch-samples', '256', '--anet-samples', '512'],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    line = [l for l in outs[0][0].splitlines() if l.startswith('{')][-1]
    assert not [l for l in outs[1][0].splitlines() if l.startswith('{')]          # rank 0 prints the one line
    out = json.loads(line)
    assert out['n_gpus'] == 2 and out['config']['parallelism'] == 'dp2' and out['config']['global_batch'] == 16
    ca = out['rccl']['custom_allreduce']
    assert 'error' not in ca and ca['us'] > 0 and ca['status_word'] == 0 and ca['max_rel_diff_vs_rccl'] < 1e-5, ca
    assert out['roofline'] is not None and out['roofline']['kernel']
    el = out['epoch_loop']
    modes = lambda x: sum(x['step_launch_modes'][k] for k in ('eager', 'captured', 'replayed'))
    # (with more than one rank the steps of a known shape run as three segment graphs with the collectives eager between them)
    assert 'error' not in el and el['n_gpus'] == 2 and modes(el) == el['steps'] and el['step_launch_modes']['replayed'] > 0
    assert out['rccl']['step_launch'].startswith('three hipGraphs')
    for leg in out['epoch_loop_anet']:
        assert 'error' not in leg and leg['n_gpus'] == 2 and leg['lengths_from'] == 'anet' and modes(leg) == leg['steps']

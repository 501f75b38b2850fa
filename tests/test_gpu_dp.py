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


def test_dp_path_on_one_rank_matches_single_path():
    from hual_amd.train import Trainer
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(29600 + os.getpid() % 300))
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
    try:
        cfg, p, wv, b, labels = pu.make_case(B=4, T=24, L=7, C=5, seed=21)
        feeds = (b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy(),
                 *[x.numpy() for x in labels])
        out = []
        for force_dp in (False, True):
            m = pu.hip_model(cfg, p, wv)
            m.set_rng(7, 3)
            tr = Trainer(m, world=1, use_graph=False, force_dp=force_dp)
            tr.set_batch(*feeds)
            losses = []
            for _ in range(2):
                tr.step(lr=1e-4, drop_rate=0.2)
                losses.append(float(tr.last_loss()))
            out.append((losses, m.params.detach().cpu().numpy().copy(), m.grads.detach().cpu().numpy().copy()))
        (l0, p0, g0), (l1, p1, g1) = out
        np.testing.assert_allclose(l0[:1], l1[:1], rtol=1e-5, atol=1e-5)
        # step 2 starts from weights that Adam (no bias correction: the first update is ~lr * 3.16 * sign-like) moved by
        # amounts that depend on rounding where a gradient is ~0 (the two paths sum the alignment loss and the float
        # atomics of dW in different orders), so its loss agrees to ~1e-5 relative, not to the last bits
        np.testing.assert_allclose(l0[1:], l1[1:], rtol=2e-4, atol=2e-4)
        assert np.abs(g0 - g1).max() <= 1e-4 * max(1.0, np.abs(g0).max())
        assert np.abs(p0 - p1).max() < 5e-4      # Adam's first steps move every weight by ~3e-4 at lr 1e-4
    finally:
        dist.destroy_process_group()

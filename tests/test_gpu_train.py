"""GPU: whole training steps (forward + backward + clip_by_global_norm + AdamWeightDecay) against the CPU oracle's
train_step (ops.py:119-174), eager and as a replayed hipGraph."""
import collections

import numpy as np
import pytest
import torch

import parity_util as pu
from oracle import seqpan_ref as R

pytestmark = pytest.mark.gpu


def _oracle_steps(cfg, p, wv, b, labels, nsteps, lr, drop, seed, offset0):
    p = collections.OrderedDict((k, v.clone()) for k, v in p.items())
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v = {k: torch.zeros_like(t) for k, t in p.items()}
    batch = (b['video'], b['lens'], b['word_ids'], b['char_ids'])
    losses = []
    for s in range(nsteps):
        p, m, v, info = R.train_step(p, m, v, cfg, wv, batch, labels, lr, drop, seed=seed, offset=offset0 + s)
        losses.append(float(info['loss']))
    return p, losses


@pytest.mark.parametrize('use_graph', [False, True])
def test_three_train_steps_match_oracle(use_graph):
    from hual_amd.train import Trainer
    lr, drop, seed, off = 1e-3, 0.2, 99, 5
    cfg, p, wv, b, labels = pu.well_conditioned_case(drop_rate=drop, rng_seed=seed, rng_offset=off, B=4, T=24, L=7, C=5, seed=21)
    ref_p1, _ = _oracle_steps(cfg, p, wv, b, labels, 1, lr, drop, seed, off)
    ref_p, ref_losses = _oracle_steps(cfg, p, wv, b, labels, 3, lr, drop, seed, off)
    m = pu.hip_model(cfg, p, wv)
    m.set_rng(seed, off)
    tr = Trainer(m, world=1, use_graph=use_graph)
    tr.set_batch(b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy(),
                 *[x.numpy() for x in labels])
    losses = []
    for s in range(3):
        tr.step(lr=lr, drop_rate=drop)
        losses.append(float(tr.last_loss()))
        if s == 0:
            # step 1 on a well conditioned batch (no ReLU within rounding of 0): the update rule itself is checked
            # tightly.  Adam without bias correction moves every weight by ~lr*3.16 in step 1.
            got = m.state_dict()
            for k, v in ref_p1.items():
                moved = float(np.abs(v.numpy() - p[k].numpy()).max())
                d = float(np.abs(got[k] - v.numpy()).max())
                assert d < 2e-5 + 0.02 * moved, (k, d, moved)
    # later steps: ReLU decisions near 0 may legitimately differ between two float32 implementations
    np.testing.assert_allclose(losses[:1], ref_losses[:1], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(losses, ref_losses, rtol=5e-2, atol=5e-2)


def test_weight_decay_mask_and_clip_on_device():
    from hual_amd import lib
    m = pu.hip_model(*pu.make_case()[:3])
    n = m.params.numel()
    g = torch.Generator().manual_seed(0)
    m.grads.copy_(torch.randn(n, generator=g))
    p0 = m.params.clone()
    m.apply_gradients(0.5)
    torch.cuda.synchronize()
    gn = float(m.grads.norm())
    gc = m.grads * (1.0 / max(gn, 1.0))
    nm, nv = 0.1 * gc, 0.001 * gc * gc
    upd = nm / (nv.sqrt() + 1e-6) + m.decay * p0
    exp = p0 - 0.5 * upd
    assert float((m.params - exp).abs().max()) < 5e-5
    assert float((m.adam_m - nm).abs().max()) < 1e-7

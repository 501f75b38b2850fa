"""GPU: whole training steps (forward + backward + clip_by_global_norm + AdamWeightDecay) against the CPU oracle's
train_step (ops.py:119-174), eager and as a replayed hipGraph."""
import collections

import numpy as np
import pytest
import torch

import parity_util as pu
from oracle import seqpan_ref as R

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('use_graph', [False, True])
def test_three_train_steps_match_oracle(use_graph):
    """three consecutive steps on one batch; after every HIP step the oracle takes the same step with the ReLU active
    sets of that HIP forward (parity_util.relu_pins), so loss and parameters are comparable tightly at every step"""
    from hual_amd.train import Trainer
    lr, drop, seed, off = 1e-3, 0.2, 99, 5
    cfg, p, wv, b, labels = pu.make_case(B=4, T=24, L=7, C=5, seed=21)
    B, T, L = 4, 24, 7
    m = pu.hip_model(cfg, p, wv)
    m.set_rng(seed, off)
    tr = Trainer(m, world=1, use_graph=use_graph)
    tr.set_batch(b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy(),
                 *[x.numpy() for x in labels])
    rp = collections.OrderedDict((k, v.clone()) for k, v in p.items())
    rm = {k: torch.zeros_like(v) for k, v in p.items()}
    rv = {k: torch.zeros_like(v) for k, v in p.items()}
    batch = (b['video'], b['lens'], b['word_ids'], b['char_ids'])
    for s in range(3):
        prev = {k: v.clone() for k, v in rp.items()}
        tr.step(lr=lr, drop_rate=drop)
        torch.cuda.synchronize()
        pins = pu.relu_pins(m, B, T, L)
        rp, rm, rv, info = R.train_step(rp, rm, rv, cfg, wv, batch, labels, lr, drop, seed=seed, offset=off + s, relu_pin=pins)
        np.testing.assert_allclose(float(tr.last_loss()), float(info['loss']), rtol=1e-3, atol=1e-3)
        assert torch.equal(tr.start_index.cpu(), info['start_index']) and torch.equal(tr.end_index.cpu(), info['end_index'])
        got = m.state_dict()
        # AdamWeightDecay without bias correction moves a weight by ~lr * m / sqrt(v): about 3.16 * lr in step 1 whatever the
        # size of its gradient, so an entry whose gradient is pure rounding noise may land anywhere within that step
        for k, v in rp.items():
            moved = float(np.abs(v.numpy() - prev[k].numpy()).max())
            d = float(np.abs(got[k] - v.numpy()).max())
            assert d < 2e-5 + 0.02 * moved, (s, k, d, moved)
        # continue the oracle from the HIP parameters so that step s+1 compares ONE step, not an accumulated drift
        rp = collections.OrderedDict((k, torch.from_numpy(got[k])) for k in rp)
        rm = {k: torch.from_numpy(a) for k, a in m.table.unpack(m.adam_m.cpu().numpy()).items()}
        rv = {k: torch.from_numpy(a) for k, a in m.table.unpack(m.adam_v.cpu().numpy()).items()}


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

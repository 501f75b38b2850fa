"""GPU: the fused multi-layer kernels against the launch sequences they replace (same arithmetic, operation for
operation, so the saved intermediates and outputs must agree BIT FOR BIT), at shapes that exercise tile boundaries:
clips shorter / longer than a workgroup's rows, row counts that are not multiples of the tile, single clips."""
import numpy as np
import pytest
import torch

import parity_util as pu

pytestmark = pytest.mark.gpu


def _run(case, drop, env, monkeypatch, names, backward=False):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    cfg, p, wv, b, labels = case
    m = pu.hip_model(cfg, p, wv)
    m.set_rng(5, 7)
    out = m.forward(b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy(), drop_rate=drop,
                    labels=tuple(x.numpy() for x in labels))
    if backward:
        m.backward()
    torch.cuda.synchronize()
    taps = {n: m.tap(n).clone() for n in names}
    res = {k: out[k].clone() for k in ('start_logits', 'end_logits', 'match_scores')}
    grads = m.grads.clone() if backward else None
    for k in env:
        monkeypatch.delenv(k)
    return taps, res, grads


CB_FWD_TAPS = ['cb.x0'] + ['%s.%s%d' % (t, n, i) for t in ('cb', 'fe0', 'fe1') for n in ('c', 'y', 'mean', 'rstd') for i in range(4)] + \
              ['%s.x%d' % (t, i) for t in ('cb', 'fe0', 'fe1') for i in range(1, 5)] + ['fe0.x0', 'fe1.x0']


@pytest.mark.parametrize('shape', [dict(B=3, T=37, L=9, C=4, seed=11, max_vlen=40), dict(B=1, T=5, L=3, C=4, seed=2, max_vlen=8),
                                   dict(B=8, T=64, L=20, C=8, seed=9, max_vlen=64), dict(B=5, T=100, L=30, C=6, seed=4, max_vlen=100),
                                   dict(B=2, T=256, L=24, C=6, seed=51, max_vlen=256)])
@pytest.mark.parametrize('drop', [0.0, 0.2])
def test_fused_conv_block_forward_bit_exact(shape, drop, monkeypatch):
    case = pu.make_case(**shape)
    t1, r1, _ = _run(case, drop, {'HUAL_FUSE_CB': '1'}, monkeypatch, CB_FWD_TAPS)
    t0, r0, _ = _run(case, drop, {'HUAL_FUSE_CB': '0'}, monkeypatch, CB_FWD_TAPS)
    for n in CB_FWD_TAPS:
        assert torch.equal(t1[n], t0[n]), (n, float((t1[n] - t0[n]).abs().max()))
    for k in r1:
        assert torch.equal(r1[k], r0[k]), k


CB_BWD_TAPS = ['d.%s.x0' % t for t in ('cb', 'fe0', 'fe1')] + ['d.%s.z%d' % (t, i) for t in ('cb', 'fe0', 'fe1') for i in range(4)]


@pytest.mark.parametrize('shape', [dict(B=3, T=37, L=9, C=4, seed=11, max_vlen=40), dict(B=1, T=5, L=3, C=4, seed=2, max_vlen=8),
                                   dict(B=8, T=64, L=20, C=8, seed=9, max_vlen=64), dict(B=5, T=100, L=30, C=6, seed=4, max_vlen=100),
                                   dict(B=2, T=256, L=24, C=6, seed=51, max_vlen=256)])
@pytest.mark.parametrize('drop', [0.0, 0.2])
def test_fused_conv_block_backward(shape, drop, monkeypatch):
    """gradient wrt the block input and every dZ operand: to 5e-6 of the tensor's scale (the two paths agree to the last bit
    or two per layer - the compiler contracts a*b+c chains of the row phase differently - and the difference is carried
    through the layers below); parameter gradients (sums over all rows, taken in a different association and - in both
    paths - with float atomics): to 2e-5 of the tensor's scale"""
    case = pu.make_case(**shape)
    t1, _, g1 = _run(case, drop, {'HUAL_FUSE_CB': '1'}, monkeypatch, CB_BWD_TAPS, backward=True)
    t0, _, g0 = _run(case, drop, {'HUAL_FUSE_CB': '0'}, monkeypatch, CB_BWD_TAPS, backward=True)
    for n in CB_BWD_TAPS:
        d, sc = float((t1[n] - t0[n]).abs().max()), float(t0[n].abs().max())
        assert d <= 5e-6 * max(sc, 1e-30), (n, d, sc)
    cfg, p, wv, b, labels = case
    m = pu.hip_model(cfg, p, wv)
    d1, d0 = m.table.unpack(g1.cpu().numpy()), m.table.unpack(g0.cpu().numpy())
    for k in d1:
        scale = float(np.abs(d0[k]).max())
        # (+ 1e-5 absolute: gradients that are zero in exact arithmetic - the key biases under the softmax - are rounding noise
        # of the split-bf16 attention products (~5e-6 on sums over all rows), and the float atomics of tri_bwd_kernel make the
        # trilinear gradients vary by ~1e-6 from run to run)
        assert float(np.abs(d1[k] - d0[k]).max()) <= 2e-5 * scale + 1e-5, (k, float(np.abs(d1[k] - d0[k]).max()), scale)


DA_FWD_TAPS = ['da%d.%s' % (li, n) for li in range(2) for n in ('ln1', 'lnt', 'mean', 'rstd', 'qkv', 'ktvt', 's_att', 'x_att', 's', 'x', 'sg',
                                                                'xg', 'o', 'g', 'gate', 'val', 'mha', 'res', 'l2', 'mean2', 'rstd2', 'out')] + \
              ['fe%d.%s' % (ps, n) for ps in range(2) for n in ('a', 'ln1.mean', 'ln1.rstd', 'qkv', 'att', 'res', 'l2', 'ln2.mean', 'ln2.rstd', 'out')] + \
              ['head.sfn', 'head.efn', 'head.mean', 'head.rstd', 'head.hs', 'head.he']


@pytest.mark.parametrize('shape', [dict(B=3, T=37, L=9, C=4, seed=11, max_vlen=40), dict(B=1, T=5, L=3, C=4, seed=2, max_vlen=8),
                                   dict(B=8, T=64, L=20, C=8, seed=9, max_vlen=64), dict(B=64, T=128, L=20, C=8, seed=12345, max_vlen=128)])
@pytest.mark.parametrize('drop', [0.0, 0.2])
def test_fused_dual_attention_forward_bit_exact(shape, drop, monkeypatch):
    """ln_proj_kernel + da_post_kernel against ln_fwd / dense / chained launches (dual attention layers, the predictor's
    feature encoders and hidden layers): every saved tensor and the model outputs, bit for bit (B=64 T=128: the bench shape)"""
    case = pu.make_case(**shape)
    t1, r1, _ = _run(case, drop, {'HUAL_FUSE_DA': '1'}, monkeypatch, DA_FWD_TAPS)
    t0, r0, _ = _run(case, drop, {'HUAL_FUSE_DA': '0'}, monkeypatch, DA_FWD_TAPS)
    for n in DA_FWD_TAPS:
        assert torch.equal(t1[n], t0[n]), (n, float((t1[n] - t0[n]).abs().max()))
    for k in r1:
        assert torch.equal(r1[k], r0[k]), k


DA_BWD_TAPS = ['d.da%d.%s' % (li, n) for li in (1, 0) for n in ('res', 'z1', 'sc', 'val', 'ln1a', 'g', 'zsg', 'zxg', 's', 'x', 's_att', 'x_att', 'in')] + \
              ['d.fe%d.%s' % (ps, n) for ps in (1, 0) for n in ('res', 'att', 'x4')] + ['d.cb.x0']


@pytest.mark.parametrize('shape', [dict(B=3, T=37, L=9, C=4, seed=11, max_vlen=40), dict(B=1, T=5, L=3, C=4, seed=2, max_vlen=8),
                                   dict(B=8, T=64, L=20, C=8, seed=9, max_vlen=64), dict(B=64, T=128, L=20, C=8, seed=12345, max_vlen=128)])
@pytest.mark.parametrize('drop', [0.0, 0.2])
def test_fused_dual_attention_backward(shape, drop, monkeypatch):
    """ln_proj_bwd_kernel + da_mid_bwd_kernel against the dense / elementwise / ln_bwd launches they replace: every gradient
    tensor a later kernel or a weight-gradient job reads, to 5e-6 of its scale (see test_fused_conv_block_backward), and
    all parameter gradients to 2e-5"""
    case = pu.make_case(**shape)
    t1, _, g1 = _run(case, drop, {'HUAL_FUSE_DA': '1'}, monkeypatch, DA_BWD_TAPS, backward=True)
    t0, _, g0 = _run(case, drop, {'HUAL_FUSE_DA': '0'}, monkeypatch, DA_BWD_TAPS, backward=True)
    for n in DA_BWD_TAPS:
        d, sc = float((t1[n] - t0[n]).abs().max()), float(t0[n].abs().max())
        assert d <= 5e-6 * max(sc, 1e-30), (n, d, sc)
    cfg, p, wv, b, labels = case
    m = pu.hip_model(cfg, p, wv)
    d1, d0 = m.table.unpack(g1.cpu().numpy()), m.table.unpack(g0.cpu().numpy())
    for k in d1:
        scale = float(np.abs(d0[k]).max())
        assert float(np.abs(d1[k] - d0[k]).max()) <= 2e-5 * scale + 1e-5, (k, float(np.abs(d1[k] - d0[k]).max()), scale)

"""Pins for the CPU oracle (oracle/seqpan_ref.py).  The reference has no tests or golden vectors for the
model path and TensorFlow is not installable here, so the oracle is pinned by:
known-answer vectors of the RNG, fp32-vs-fp64 agreement, the structural invariants the reference graph
implies, the parameter-count formula, and autograd gradcheck in fp64."""
import numpy as np
import pytest
import torch

from oracle import philox as px
from oracle import seqpan_ref as R


@pytest.fixture(scope='module')
def small():
    cfg = R.default_cfg(max_vlen=32, num_words=60)
    p = R.init_params(cfg, seed=1)
    wv = R.init_word_vectors(cfg)
    b = R.synthetic_batch(cfg, 3, 20, 6, 5, seed=3)
    return cfg, p, wv, b


def _labels(b, T):
    from hual_amd import data
    y1, y2, m, i = data.make_labels(b['s_ind'], b['e_ind'], b['lens'].numpy(), max_len=T)
    return torch.tensor(y1), torch.tensor(y2), torch.tensor(m), torch.tensor(i, dtype=torch.float32)


def test_philox_known_answers():
    # Random123 kat_vectors: philox4x32-10
    assert [int(x) for x in px.philox4x32_10(0, 0, 0, 0, 0, 0)] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert [int(x) for x in px.philox4x32_10(*([0xffffffff] * 6))] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert [int(x) for x in px.philox4x32_10(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0)] \
        == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_philox7_known_answers():
    """the generator the build uses since round 4: Philox4x32-7, Random123 kat_vectors ("philox4x32 7 ...")"""
    assert px.PHILOX_ROUNDS == 7
    assert [int(x) for x in px.philox4x32(0, 0, 0, 0, 0, 0)] == [0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48]
    assert [int(x) for x in px.philox4x32(*([0xffffffff] * 6))] == [0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662]
    assert [int(x) for x in px.philox4x32(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0)] \
        == [0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a]


def test_dropout_mask_statistics():
    r = px.DropoutRNG(seed=7, offset=3, rate=0.2)
    m = r.mask(px.SITE_VIDEO, np.arange(512), 1024)
    assert set(np.unique(m)) == {np.float32(0.0), np.float32(1.25)}
    assert abs((m > 0).mean() - 0.8) < 5e-3
    m2 = px.DropoutRNG(seed=7, offset=4, rate=0.2).mask(px.SITE_VIDEO, np.arange(512), 1024)
    assert (m != m2).mean() > 0.2


def test_gumbel_noise_of_the_matching_head():
    """ops.py:6-9 / layers.py:163-166: uniform draws on the 2^-24 grid from one Philox call per row (site SITE_GUMBEL); the noise
    -log(-log(u + 1e-20) + 1e-20) is finite at both ends of the grid and has the Gumbel(0, 1) mean (Euler's constant)"""
    u = px.gumbel_uniform(seed=5, offset=7, rows=np.arange(20000))
    assert u.shape == (20000, 4) and u.dtype == np.float32
    assert u.min() >= 0.0 and u.max() < 1.0 and np.all(u * 2.0 ** 24 == np.floor(u * 2.0 ** 24))
    w = px.philox4x32(0, 3, px.SITE_GUMBEL, 7, 5, 0)
    assert [float(x) for x in u[3]] == [float(np.float32(int(x) >> 8) * np.float32(2.0 ** -24)) for x in w]
    ends = np.array([0.0, 1.0 - 2.0 ** -24], dtype=np.float32)
    n_ends = -np.log(-np.log(ends + np.float32(1e-20)) + np.float32(1e-20))
    assert np.isfinite(n_ends).all() and n_ends[0] < -3.8 and n_ends[1] > 16.0
    noise = -np.log(-np.log(u.astype(np.float64) + 1e-20) + 1e-20)
    assert abs(noise.mean() - 0.5772) < 0.02
    assert (px.gumbel_uniform(5, 8, np.arange(64)) != u[:64]).mean() > 0.9


def test_param_count_formula():
    assert R.param_count(R.default_cfg(num_chars=40)) == 1186508       # SURVEY.md App. A, Charades YAML
    sh = R.param_shapes(R.default_cfg())
    assert sh['conv_block/depthwise_conv_layers_0/depthwise_filter'][0] == [7, 1, 128, 1]
    assert sh['q2v_attn/dense/kernel'][0] == [1, 512, 128]
    assert not R.uses_weight_decay('conv_block/layer_norm_0/layer_norm_scale')
    assert not R.uses_weight_decay('d_attn_0/dual_multihead_attention/bilinear_1/bias')
    assert R.uses_weight_decay('pos_emb/position_embeddings') and R.uses_weight_decay('label_emb')


def test_fp32_vs_fp64(small):
    cfg, p, wv, b = small
    o32 = R.forward(p, cfg, wv, b['video'], b['lens'], b['word_ids'], b['char_ids'])
    p64 = {k: v.double() for k, v in p.items()}
    o64 = R.forward(p64, cfg, wv.double(), b['video'].double(), b['lens'], b['word_ids'], b['char_ids'])
    for k in ('start_logits', 'end_logits', 'match_scores'):
        assert (o32[k].double() - o64[k]).abs().max().item() < 1e-4
    assert torch.equal(o32['start_index'], o64['start_index']) and torch.equal(o32['end_index'], o64['end_index'])


def test_invariants(small):
    cfg, p, wv, b = small
    T = b['video'].shape[1]
    lab = _labels(b, T)
    o = R.forward(p, cfg, wv, b['video'], b['lens'], b['word_ids'], b['char_ids'], labels=lab, want_tap=True)
    tap = o['tap']
    vm = (torch.arange(T)[None] < b['lens'][:, None])
    # dual_multihead_attention output == 0 at padded from-rows (layers.py:110)
    assert tap['da0.mha.v'][~vm].abs().max().item() == 0.0
    qm = b['word_ids'] != 0
    assert tap['da1.mha.q'][~qm].abs().max().item() == 0.0
    # outputs == 0 at padded T (model.py:97); match_scores rows sum to 1
    assert tap['outputs'][~vm].abs().max().item() == 0.0
    assert (o['match_scores'].sum(-1) - 1).abs().max().item() < 1e-6
    assert bool((o['start_index'] <= o['end_index']).all())
    assert bool((o['end_index'] < b['lens']).all())
    for k in ('loss', 'loc_loss', 'match_loss', 'align_loss'):
        assert torch.isfinite(o[k])
    # v-side and q-side share one weight set: count of d_attn variables is per layer, not per side
    n = [k for k in p if k.startswith('d_attn_0/')]
    assert len(n) == 2 * 3 + 2 * 10 + 2 * 3 + 4


def test_dropout_is_reproducible_and_offsets_differ(small):
    cfg, p, wv, b = small
    args = (p, cfg, wv, b['video'], b['lens'], b['word_ids'], b['char_ids'])
    a = R.forward(*args, drop_rate=0.2, seed=11, offset=1)['start_logits']
    a2 = R.forward(*args, drop_rate=0.2, seed=11, offset=1)['start_logits']
    c = R.forward(*args, drop_rate=0.2, seed=11, offset=2)['start_logits']
    assert torch.equal(a, a2) and not torch.equal(a, c)


def test_gradcheck_blocks_fp64():
    torch.manual_seed(0)
    cfg = R.default_cfg(dim=16, num_heads=2, vdim=12, max_vlen=8, num_words=20, num_chars=10, char_dim=6, word_dim=10)
    p = {k: v.double() for k, v in R.init_params(cfg, seed=2).items()}
    B, T, L = 2, 6, 4
    v = torch.randn(B, T, 16, dtype=torch.double, requires_grad=True)
    q = torch.randn(B, L, 16, dtype=torch.double, requires_grad=True)
    vm = torch.tensor([[1, 1, 1, 1, 1, 1], [1, 1, 1, 1, 0, 0]], dtype=torch.int32)
    qm = torch.tensor([[1, 1, 1, 1], [1, 1, 0, 0]], dtype=torch.int32)
    f1 = lambda a, c: R.dual_attn_block(a, c, p, 'd_attn_0', 2, vm, qm, None, 0, None)
    assert torch.autograd.gradcheck(f1, (v, q), atol=1e-6)
    f2 = lambda a, c: R.cq_attention(a, c, vm, qm, p, 'q2v_attn', None, 0, None, 0, None)
    assert torch.autograd.gradcheck(f2, (v, q), atol=1e-6)
    inner = torch.tensor([[0, 1, 1, 1, 0, 0], [0, 1, 1, 0, 0, 0]], dtype=torch.double)
    f3 = lambda a, c: R.lossfun_aligment(c, a, qm, vm, inner)
    assert torch.autograd.gradcheck(f3, (v, q), atol=1e-6)


def test_overfit_loss_decreases():
    cfg = R.default_cfg(max_vlen=16, num_words=40)
    p = R.init_params(cfg, seed=4)
    wv = R.init_word_vectors(cfg)
    b = R.synthetic_batch(cfg, 4, 12, 5, 5, seed=9)
    lab = _labels(b, 12)
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v = {k: torch.zeros_like(t) for k, t in p.items()}
    batch = (b['video'], b['lens'], b['word_ids'], b['char_ids'])
    losses = []
    for it in range(10):
        p, m, v, info = R.train_step(p, m, v, cfg, wv, batch, lab, lr=1e-4, drop_rate=0.0)
        losses.append(float(info['loss']))
    assert losses[-1] < losses[0]


def test_adam_weight_decay_semantics():
    p = {'a/kernel': torch.ones(3), 'a/bias': torch.ones(3)}
    g = {'a/kernel': torch.full((3,), 0.5), 'a/bias': torch.full((3,), 0.5)}
    m = {k: torch.zeros(3) for k in p}
    v = {k: torch.zeros(3) for k in p}
    p2, m2, v2 = R.adam_weight_decay_step(dict(p), g, m, v, lr=0.1)
    upd = (0.1 * 0.5) / (np.sqrt(0.001 * 0.25) + 1e-6)           # no bias correction (ops.py:166-168)
    assert abs(float(p2['a/bias'][0]) - (1 - 0.1 * upd)) < 1e-6
    assert abs(float(p2['a/kernel'][0]) - (1 - 0.1 * (upd + 0.01))) < 1e-6


def test_train_step_leaves_the_callers_state_alone():
    """a discarded train_step (warm-up, timing scan) must not advance the caller's parameters or Adam slots: bench.py's
    oracle trajectory in round 3 started from slots three discarded steps had written into"""
    cfg = R.default_cfg(max_vlen=16, num_words=40)
    p = R.init_params(cfg, seed=4)
    wv = R.init_word_vectors(cfg)
    b = R.synthetic_batch(cfg, 2, 10, 5, 5, seed=9)
    lab = _labels(b, 10)
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v = {k: torch.zeros_like(t) for k, t in p.items()}
    p0 = {k: t.clone() for k, t in p.items()}
    batch = (b['video'], b['lens'], b['word_ids'], b['char_ids'])
    a = R.train_step(p, m, v, cfg, wv, batch, lab, lr=1e-3, drop_rate=0.2, seed=1, offset=0)
    assert all(float(m[k].abs().max()) == 0.0 and float(v[k].abs().max()) == 0.0 for k in m)
    assert all(torch.equal(p[k], p0[k]) for k in p)
    c = R.train_step(p, m, v, cfg, wv, batch, lab, lr=1e-3, drop_rate=0.2, seed=1, offset=0)
    assert all(torch.equal(a[0][k], c[0][k]) for k in p) and float(a[3]['loss']) == float(c[3]['loss'])
    assert any(float(a[1][k].abs().max()) > 0 for k in m)


def test_softmax_cr_is_a_float32_softmax():
    """the reproducible softmax used for the span selection agrees with torch's float32 softmax to a few ulps and with
    the float64 softmax to float32 rounding; rows sum to 1"""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(32, 200, generator=g) * 4
    x[:, 150:] = -1e30                                            # masked tail (ops.py:89)
    p = R.softmax_cr(x)
    ref64 = torch.softmax(x.double(), dim=1)
    assert float((p.double() - ref64).abs().max()) < 2e-7
    assert float((p - torch.softmax(x, dim=1)).abs().max()) < 3e-7
    assert float((p.sum(1) - 1).abs().max()) < 1e-5
    assert float(p[:, 150:].abs().max()) == 0.0


def test_relu_pin_only_moves_rounding_level_units():
    """relu_pin (test aid): pinning the active sets to the oracle's own signs changes nothing; flipping the unit closest
    to zero changes the forward by at most |z| of that unit"""
    cfg = R.default_cfg(max_vlen=32, num_words=60)
    p = R.init_params(cfg, seed=1)
    wv = R.init_word_vectors(cfg)
    b = R.synthetic_batch(cfg, 2, 12, 5, 5, seed=3)
    out = R.forward(p, cfg, wv, b['video'], b['lens'], b['word_ids'], b['char_ids'], want_tap=True)
    tap = out['tap']
    pins = {'cb.v': [tap['cb.y%d.v' % i] > 0 for i in range(4)], 'cb.q': [tap['cb.y%d.q' % i] > 0 for i in range(4)],
            'fe0': [tap['fe0.y%d' % i] > 0 for i in range(4)], 'fe1': [tap['fe1.y%d' % i] > 0 for i in range(4)]}
    out2 = R.forward(p, cfg, wv, b['video'], b['lens'], b['word_ids'], b['char_ids'], relu_pin=pins)
    assert torch.equal(out['start_logits'], out2['start_logits'])

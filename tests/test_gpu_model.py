"""End-to-end GPU parity: HIP SeqPAN (through the C ABI) vs the CPU oracle on identical seeded inputs.
Tolerance 1e-3 (north_star), span argmax indices bit exact."""
import numpy as np
import pytest
import torch

import parity_util as pu

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _assert_rows(rows, kinds):
    pu.assert_rows(rows, kinds, TOL)       # gradients: relative to the tensor's own scale (parity_util.row_ok)


@pytest.mark.parametrize('drop', [0.0, 0.2])
def test_forward_backward_parity_small(drop):
    case = pu.make_case()
    rows, idx_equal, o, h, m = pu.compare(*case, drop_rate=drop)
    _assert_rows(rows, ('tap', 'out', 'loss'))
    assert idx_equal
    _assert_rows(rows, ('grad', 'gl2'))


def test_second_shape_with_dropout():
    case = pu.make_case(B=4, T=24, L=7, C=5, seed=21)
    rows, idx_equal, o, h, m = pu.compare(*case, drop_rate=0.2)
    _assert_rows(rows, ('tap', 'out', 'loss', 'grad', 'gl2'))
    assert idx_equal


def test_ragged_shapes_parity():
    # T, L not multiples of 16; C = 4 (minimum); one clip of length 1-ish neighbours
    case = pu.make_case(B=3, T=37, L=9, C=4, seed=11, max_vlen=40)
    rows, idx_equal, o, h, m = pu.compare(*case, drop_rate=0.1)
    _assert_rows(rows, ('tap', 'out', 'loss', 'grad', 'gl2'))
    assert idx_equal


@pytest.mark.parametrize('shape', [dict(), dict(B=4, T=64, L=12, C=6, seed=8, max_vlen=64, vdim=512)])
def test_bfloat16_video_feed_parity(shape):
    """hual_batch.video_dtype = HUAL_DTYPE_BF16 (BASELINE configs[1]: bf16 clip features): the feature-load and the
    video_conv1d weight-gradient kernels read bfloat16 features; same graph on the same values in the oracle"""
    case = pu.make_case(**shape)
    rows, idx_equal, o, h, m = pu.compare(*case, drop_rate=0.2, video_bf16=True)
    _assert_rows(rows, ('tap', 'out', 'loss', 'grad', 'gl2'))
    assert idx_equal


def test_bfloat16_video_feed_any_width():
    """vdim 320 (not a multiple of 128: the feature-load launch ends on a partial weight block) with bfloat16 features"""
    case = pu.make_case(vdim=320)
    rows, idx_equal, o, h, m = pu.compare(*case, drop_rate=0.2, video_bf16=True)
    _assert_rows(rows, ('tap', 'out', 'loss', 'grad', 'gl2'))
    assert idx_equal


def test_gumbel_branch_of_the_matching_loss():
    """loss.no_gumbel false (layers.py:163-166, ops.py:6-9): logits = (logits + gumbel noise) / tau in the matching head, noise from
    the shared Philox stream (site HUAL_SITE_GUMBEL) - every tensor, the losses and all gradients, with and without dropout; the
    noise is also there in label-free calls (the reference's graph samples it whenever match_scores is evaluated)"""
    for rate in (0.0, 0.2):
        cfg, p, wv, b, labels = pu.make_case(B=3, T=20, L=6, C=5, seed=11)
        cfg['no_gumbel'] = False
        cfg['tau'] = 0.3
        rows, idx_equal, o, h, m = pu.compare(cfg, p, wv, b, labels, drop_rate=rate)
        _assert_rows(rows, ('tap', 'out', 'loss', 'grad', 'gl2'))
        assert idx_equal
    # the noise changes the scores (the test would pass vacuously if the flag were ignored on both sides)
    cfg2, p2, wv2, b2, labels2 = pu.make_case(B=3, T=20, L=6, C=5, seed=11)
    feeds = (b2['video'], b2['lens'], b2['word_ids'], b2['char_ids'])
    plain = pu.hip_model(cfg2, p2, wv2).forward(*feeds, drop_rate=0.0)['match_scores'].cpu().numpy()
    noisy = m.forward(*feeds, drop_rate=0.0)['match_scores'].cpu().numpy()
    assert np.abs(plain - noisy).max() > 0.05


def test_weight_outside_the_fp16_image_range_poisons_the_loss():
    """the dense weights travel as fp16 hi + lo images scaled by 2^10 (csrc/bf16x3.h): |w| >= 63 does not fit.  The pack launch
    flags such a weight and the loss launch turns the flag into NaN losses instead of silently wrong products"""
    import collections
    cfg, p, wv, b, labels = pu.make_case(B=2, T=16, L=6, C=5, seed=3)
    feeds = (b['video'], b['lens'], b['word_ids'], b['char_ids'])
    m = pu.hip_model(cfg, p, wv)
    ok = m.forward(*feeds, drop_rate=0.0, labels=labels)
    assert np.isfinite(float(ok['loss']))
    p2 = collections.OrderedDict((k, v.clone()) for k, v in p.items())
    p2['d_attn_0/dual_multihead_attention/f_key/kernel'].view(-1)[5] = 100.0
    m2 = pu.hip_model(cfg, p2, wv)
    bad = m2.forward(*feeds, drop_rate=0.0, labels=labels)
    assert np.isnan(float(bad['loss'])) and np.isnan(float(bad['loc_loss']))
    # label-free calls have no loss launch to carry the flag: the logits are NaN and the span indices -1
    free_ok = m.forward(*feeds, drop_rate=0.0)
    assert np.isfinite(free_ok['start_logits'].cpu().numpy()).all() and int(free_ok['start_index'].min()) >= 0
    free_bad = m2.forward(*feeds, drop_rate=0.0)
    assert np.isnan(free_bad['start_logits'].cpu().numpy()).all() and np.isnan(free_bad['end_logits'].cpu().numpy()).all()
    assert (free_bad['start_index'].cpu().numpy() == -1).all() and (free_bad['end_index'].cpu().numpy() == -1).all()


def test_activation_outside_the_fp16_operand_range_is_loud():
    """activations travel to the matrix cores as fp16 pairs of 2^4 x value (attention Q / K / V, the fused chains' operands, the
    weight-gradient launch's A side): |x| >= 4094 does not fit.  Unlike an oversized WEIGHT (flagged by the pack launch) nothing checks
    activations up front - the products turn Inf - Inf = NaN and the NaN has to arrive where the caller looks: NaN losses with labels,
    NaN logits and span indices -1 without (never a plausible-looking span)."""
    import collections
    cfg, p, wv, b, labels = pu.make_case(B=2, T=16, L=6, C=5, seed=3)
    feeds = (b['video'], b['lens'], b['word_ids'], b['char_ids'])
    p2 = collections.OrderedDict((k, v.clone()) for k, v in p.items())
    p2['d_attn_0/dual_multihead_attention/query/bias'].view(-1)[3] = 5000.0      # a bias is not range checked: Q[:, 3] = 5000 + ...
    m2 = pu.hip_model(cfg, p2, wv)
    bad = m2.forward(*feeds, drop_rate=0.0, labels=labels)
    assert np.isnan(float(bad['loss']))
    free = m2.forward(*feeds, drop_rate=0.0)
    assert np.isnan(free['start_logits'].cpu().numpy()).any()
    assert (free['start_index'].cpu().numpy() == -1).all() and (free['end_index'].cpu().numpy() == -1).all()


def test_feed_errors_are_python_exceptions_like_the_references():
    """the façade raises where sess.run would (runner_utils.py:53-65 feeds): the placeholder's T is the batch's longest clip
    (model.py:31), the feature width is the configured one (model.py:17), a word has at least the char CNN's four characters
    (modules.py:19-38), a clip fits the position table (modules.py:44)"""
    from hual_amd import lib
    cfg, p, wv, b, labels = pu.make_case(B=2, T=12, L=5, C=5, seed=4, max_vlen=16)
    m = pu.hip_model(cfg, p, wv)
    v, ln, w, c = b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy()
    m.forward(v, ln, w, c)                                            # fine as it is
    with pytest.raises(ValueError, match='max\\(video_seq_len\\)'):
        m.forward(np.concatenate([v, np.zeros_like(v[:, :1])], axis=1), ln, w, c)
    with pytest.raises(ValueError, match='vdim'):
        m.forward(v[:, :, :512], ln, w, c)
    with pytest.raises(ValueError, match='4 chars'):
        m.forward(v, ln, w, c[:, :, :3])
    long_v = np.zeros((2, 20, v.shape[2]), dtype=np.float32)
    with pytest.raises(lib.HualError, match='max_vlen'):
        m.forward(long_v, np.array([20, 20], dtype=np.int32), w, c)   # 20 frames against a position table of 16
    torch.cuda.synchronize()
    o = m.forward(v, ln, w, c)                                        # and the model is still usable
    assert torch.isfinite(o['start_logits']).all()

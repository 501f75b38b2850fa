"""End-to-end GPU parity: HIP SeqPAN (through the C ABI) vs the CPU oracle on identical seeded inputs.
Tolerance 1e-3 (north_star), span argmax indices bit exact."""
import pytest
import torch

import parity_util as pu

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _assert_rows(rows, kinds):
    bad = [(k, n, d, r) for (k, n, d, r) in rows if k in kinds and not (d <= TOL or d <= TOL * r)]
    assert not bad, 'parity failures:\n' + pu.format_report(bad)


@pytest.mark.parametrize('drop', [0.0, 0.2])
def test_forward_backward_parity_small(drop):
    case = pu.make_case()
    rows, idx_equal, o, h, m = pu.compare(*case, drop_rate=drop)
    _assert_rows(rows, ('tap', 'out', 'loss'))
    assert idx_equal
    _assert_rows(rows, ('grad',))


def test_second_shape_with_dropout():
    case = pu.make_case(B=4, T=24, L=7, C=5, seed=21)
    rows, idx_equal, o, h, m = pu.compare(*case, drop_rate=0.2)
    _assert_rows(rows, ('tap', 'out', 'loss', 'grad'))
    assert idx_equal


def test_ragged_shapes_parity():
    # T, L not multiples of 16; C = 4 (minimum); one clip of length 1-ish neighbours
    case = pu.make_case(B=3, T=37, L=9, C=4, seed=11, max_vlen=40)
    rows, idx_equal, o, h, m = pu.compare(*case, drop_rate=0.1)
    _assert_rows(rows, ('tap', 'out', 'loss', 'grad'))
    assert idx_equal


@pytest.mark.parametrize('shape', [dict(), dict(B=4, T=64, L=12, C=6, seed=8, max_vlen=64, vdim=512)])
def test_bfloat16_video_feed_parity(shape):
    """hual_batch.video_dtype = HUAL_DTYPE_BF16 (BASELINE configs[1]: bf16 clip features): the feature-load and the
    video_conv1d weight-gradient kernels read bfloat16 features; same graph on the same values in the oracle"""
    case = pu.make_case(**shape)
    rows, idx_equal, o, h, m = pu.compare(*case, drop_rate=0.2, video_bf16=True)
    _assert_rows(rows, ('tap', 'out', 'loss', 'grad'))
    assert idx_equal


def test_bfloat16_video_feed_any_width():
    """vdim 320 (not a multiple of 128: the feature-load launch ends on a partial weight block) with bfloat16 features"""
    case = pu.make_case(vdim=320)
    rows, idx_equal, o, h, m = pu.compare(*case, drop_rate=0.2, video_bf16=True)
    _assert_rows(rows, ('tap', 'out', 'loss', 'grad'))
    assert idx_equal

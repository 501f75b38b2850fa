"""GPU parity at the other shapes BASELINE.json names and at the edges of the supported range:
the bench shape itself (configs[1]: B=64, T=128, vdim=1024), the per-GPU shape of configs[3] (B=32, T=256), configs[0]
(B=16, T=64, vdim=512), ActivityNet dims (char_dim 100, max_vlen 100), single-clip batches, clips of length 1-2 frames
next to full-length clips, one-word queries.  Tolerance 1e-3 for EVERY tensor (north_star); gradients are held to 1e-3 of
the tensor's OWN largest magnitude (parity_util.row_ok - no absolute arm), span indices equal."""
import numpy as np
import pytest
import torch

import parity_util as pu

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _check(rows, idx_equal, kinds=('tap', 'out', 'loss', 'grad', 'gl2')):
    """parity_util.row_ok: forward tensors abs-or-rel 1e-3; every gradient tensor RELATIVE: d <= 1e-3 max(max|ref|, 1e-3 gmax)"""
    pu.assert_rows(rows, kinds, TOL)
    assert idx_equal


def _check_all(case, drop_rate):
    """every tap, output, loss term and ALL gradient tensors within 1e-3, span indices equal; the oracle evaluates its
    ReLUs on the active sets of the HIP forward (parity_util.relu_pins), so no gradient tolerance depends on how close
    some pre-activation happens to be to zero"""
    rows, idx_equal, o, h, m = pu.compare(*case, drop_rate=drop_rate)
    _check(rows, idx_equal)
    return rows


def test_bench_shape_c2():
    """BASELINE.json configs[1] = the shape bench.py times: B=64, T=128, vdim=1024, L=20, C=8, dropout 0.2.  This is the only
    shape at which the 48-row dense variant, the automatic dW row split and the XCD-aware attention order (>= 8 clips) run."""
    case = pu.make_case(B=64, T=128, L=20, C=8, seed=12345, max_vlen=128, vdim=1024)
    rows, idx_equal, o, h, m = pu.compare(*case, drop_rate=0.2)
    _check(rows, idx_equal)
    pin = [r for r in rows if r[0] == 'pin'][0]
    print('c2 ReLU pins: %d of %d units differ from the sign of the oracle z (all rounding-level)' % (pin[2], pin[3]))
    # how much does the oracle's reproducible softmax (seqpan_ref.softmax_cr, the arithmetic loc_kernel follows) matter?
    # spans from a plain float32 softmax (torch's kernel; tf.nn.softmax is a third implementation) on the same logits:
    from oracle import seqpan_ref as R
    v_mask = (torch.arange(128).unsqueeze(0) < case[3]['lens'].long().unsqueeze(1)).to(torch.int32)
    si, ei = R.ans_predictor(o['start_logits'].detach(), o['end_logits'].detach(), v_mask, softmax=lambda x: torch.softmax(x, 1))
    n_diff = int(((si != o['start_index']) | (ei != o['end_index'])).sum())
    print('c2 spans that differ between softmax_cr and torch.softmax: %d of 64' % n_diff)
    assert n_diff == 0


def test_bench_shape_c2_unpinned_fp64_oracle():
    """the bench shape against the PLAIN oracle - no ReLU pins, nothing taken from the implementation under test - evaluated
    in float64: every forward tap, output and loss term within 1e-3, span indices equal"""
    case = pu.make_case(B=64, T=128, L=20, C=8, seed=12345, max_vlen=128, vdim=1024)
    rows, idx_equal, o, h, m = pu.compare(*case, drop_rate=0.2, with_grads=False, pin_relu=False, oracle_dtype=torch.float64)
    assert not [r for r in rows if r[0] == 'pin']
    _check(rows, idx_equal, kinds=('tap', 'out', 'loss'))


@pytest.mark.parametrize('shape', [dict(B=3, T=20, L=6, C=5, seed=3, max_vlen=32), dict(B=16, T=64, L=20, C=8, seed=777, max_vlen=64, vdim=512)])
def test_unpinned_oracle_forward(shape):
    """small shape and c1 against the plain float32 oracle with its own ReLU signs (forward tensors; gradients need the
    shared active sets, see parity_util.compare)"""
    rows, idx_equal, o, h, m = pu.compare(*pu.make_case(**shape), drop_rate=0.2, with_grads=False, pin_relu=False)
    _check(rows, idx_equal, kinds=('tap', 'out', 'loss'))


def test_c4_per_gpu_shape():
    """BASELINE.json configs[3] per GPU: B=32, T=256 (ActivityNet, 8 GPUs x 32 clips), dropout 0.2"""
    case = pu.make_case(B=32, T=256, L=20, C=8, seed=4321, max_vlen=256, vdim=1024)
    _check_all(case, 0.2)


def test_two_row_tile_shape():
    """R = B (T + L) = 5920 rows: 24 rows per workgroup = TWO 16-row tiles in the T-form kernels of the dual-attention block
    (da_post_kernel<2>, da_mid_bwd_kernel<2>: one, two and three tiles are separate instantiations - c1 runs <1>, c2 / c4 run <3>),
    the last tile half used"""
    case = pu.make_case(B=40, T=128, L=20, C=8, seed=2468, max_vlen=128, vdim=512)
    _check_all(case, 0.2)


def test_four_row_tile_shape():
    """R = 64 (256 + 20) = 17664 rows: more than 256 workgroups of the largest tile everywhere - ln_proj_kernel<4> (64 rows per
    workgroup, four row tiles), the 48-row kernels (da_post / da_mid_bwd / ln_proj_bwd <3>) in two rounds of workgroups"""
    case = pu.make_case(B=64, T=256, L=20, C=8, seed=1357, max_vlen=256, vdim=512)
    _check_all(case, 0.2)


def test_c1_shape():
    """BASELINE.json configs[0]: B=16, T=64, 'D=512' read as vdim=512, dropout 0.2"""
    case = pu.make_case(B=16, T=64, L=20, C=8, seed=777, max_vlen=64, vdim=512)
    _check_all(case, 0.2)


def test_c1_shape_yaml_vdim():
    """BASELINE.json configs[0] with the YAML's vdim 1024: B=16, T=64, L=20, C=8 - the shape both legs of
    bench.py's cpu_baseline are timed on"""
    case = pu.make_case(B=16, T=64, L=20, C=8, seed=778, max_vlen=64, vdim=1024)
    _check_all(case, 0.2)


def test_bench_shape_c2_bf16_feed():
    """BASELINE.json configs[1] AS WRITTEN: bfloat16 clip features at B=64, T=128, vdim=1024 (hual_batch.video_dtype = BF16) -
    every tap, output, loss term and all 170 gradients within 1e-3 of the oracle run on the same (bf16-representable) values"""
    case = pu.make_case(B=64, T=128, L=20, C=8, seed=12345, max_vlen=128, vdim=1024)
    rows, idx_equal, o, h, m = pu.compare(*case, drop_rate=0.2, video_bf16=True)
    _check(rows, idx_equal)


def test_activitynet_dims():
    # configs/anet/SeqPAN.yaml: char_dim 100, max_vlen 100; longest sentences ~30 words
    case = pu.make_case(B=3, T=100, L=30, C=9, seed=41, max_vlen=100, char_dim=100)
    _check_all(case, 0.2)


def test_t256_configs3_shape():
    case = pu.make_case(B=2, T=256, L=24, C=6, seed=51, max_vlen=256)
    _check_all(case, 0.1)


def test_long_clips_without_dropout():
    """T > 128 with dropout OFF (evaluation-style step): the no-dropout instantiations of the long-clip kernels - the eight-wave attention
    backward job (csrc/attn.hip attn_bwd_big_kernel) and the 16-wave context-query kernels (csrc/cqwide.hip) - on a ragged length"""
    case = pu.make_case(B=2, T=200, L=17, C=5, seed=53, max_vlen=224)
    _check_all(case, 0.0)


def test_long_clips_with_long_queries():
    """T = 256 with a 40-word query: no LDS form of the context-query kernels holds the 256 x 48 score matrices - the global-operand
    kernels keep them in global memory (csrc/cq.hip cq_fwd_global / cq_bwd_global: slow, but every T, L <= 256 runs)"""
    case = pu.make_case(B=2, T=256, L=40, C=5, seed=71, max_vlen=256)
    _check_all(case, 0.2)


def test_long_queries():
    """L = 40 words (> 32): the staged context-query kernels take their general softmax branches (rows / columns longer than
    a 32-lane half) - T + L still within their 160 padded rows"""
    case = pu.make_case(B=3, T=64, L=40, C=5, seed=47, max_vlen=64)
    _check_all(case, 0.2)


@pytest.mark.parametrize('shape', [dict(B=4, T=100, L=79, C=22, seed=91, max_vlen=100),       # the longest query / word of the fixture
                                   dict(B=6, T=100, L=45, C=13, seed=92, max_vlen=100),       # a typical long-query batch at max_vlen 100
                                   dict(B=3, T=256, L=64, C=16, seed=93, max_vlen=256)])      # the same at configs[3]'s T
def test_activitynet_annotation_lengths(shape):
    """the padded shapes the reference's OWN ActivityNet annotations produce (tests/golden/lengths_anet.npz <- data/anet_gt/train.json
    through data_loader.py:23-28: queries of up to 79 words, words of up to 22 characters; a quarter of the batches at batch 16 holds a
    query of more than 32 words): the global-operand context-query kernels, the pool_align_bwd instantiations for 33-128 words, the
    char CNN at 22 characters per word"""
    _check_all(pu.make_case(char_dim=100, **shape), 0.2)


@pytest.mark.parametrize('shape', [dict(B=1, T=256, L=256, C=4, seed=5, max_vlen=256),       # both sides at the kernels' maximum
                                   dict(B=2, T=256, L=128, C=32, seed=6, max_vlen=256),      # 32-character words
                                   dict(B=2, T=3, L=200, C=5, seed=7, max_vlen=200)])        # a query far longer than its clip
def test_maximum_sizes(shape):
    """the edges of the supported range (T, L <= 256; hual_seqpan_forward rejects more): every context-query form, the attention
    kernels at 256 keys on both sides, the char CNN at 32 characters per word"""
    _check_all(pu.make_case(**shape), 0.2)


def test_single_clip_batch():
    case = pu.make_case(B=1, T=33, L=5, C=4, seed=61, max_vlen=40)
    _check_all(case, 0.0)


def test_tiny_clips_and_one_word_queries():
    """video_seq_len = 1 and 2 next to a full clip, a query of one word: padded rows dominate, every mask path is hit"""
    cfg, p, wv, b, labels = pu.make_case(B=4, T=19, L=6, C=5, seed=71, max_vlen=24)
    lens = np.array([19, 1, 2, 7], dtype=np.int32)
    b['lens'] = torch.tensor(lens)
    for k in range(4):
        b['video'][k, lens[k]:] = 0.0
    b['word_ids'][1, 1:] = 0
    b['char_ids'][1, 1:] = 0
    from hual_amd import data
    s = np.array([3, 0, 0, 2]); e = np.array([15, 0, 1, 5])
    y1, y2, mm, ii = data.make_labels(s, e, lens, max_len=19)
    labels = (torch.tensor(y1), torch.tensor(y2), torch.tensor(mm), torch.tensor(ii, dtype=torch.float32))
    _check_all((cfg, p, wv, b, labels), 0.0)


@pytest.mark.parametrize('vdim', [512, 320, 2048])
def test_other_feature_widths(vdim):
    """vdim 512 (BASELINE configs[0], K-split feature kernel with 128-row quarters), 320 (generic dense launch: not a
    multiple of 256) and 2048 (generic deep-K launch: above the K-split kernel's LDS budget)"""
    case = pu.make_case(B=3, T=21, L=6, C=5, seed=81, max_vlen=24, vdim=vdim)
    _check_all(case, 0.2)


@pytest.mark.parametrize('shape', [dict(B=1, T=5, L=3, C=4, seed=2, max_vlen=8), dict(B=8, T=64, L=20, C=8, seed=9, max_vlen=64),
                                   dict(B=5, T=100, L=30, C=6, seed=4, max_vlen=100), dict(B=3, T=37, L=9, C=4, seed=11, max_vlen=40),
                                   dict(B=2, T=9, L=5, C=16, seed=13, max_vlen=16)])
@pytest.mark.parametrize('drop', [0.0, 0.2])
def test_tile_boundary_shapes(shape, drop):
    """shapes that exercise the tile boundaries of the fused multi-layer kernels: clips shorter / longer than a workgroup's
    rows, row counts that are not multiples of the tile, a single 5-frame clip; chars per word 4 / 8 / 16 (the char CNN's pool epilogue)
    and 6 (its stand-alone pooling kernel)"""
    _check_all(pu.make_case(**shape), drop)

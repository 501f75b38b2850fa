"""GPU: the per-block C-ABI entry points (include/hual_seqpan.h) against the corresponding functions of the
CPU oracle on random activations, forward and backward, dropout on: hual_video_proj_ln_fwd, hual_conv_block_fwd/bwd
(modules.py:59-70), hual_dual_attn_fwd/bwd (modules.py:73-89, layers.py:59-111), hual_cq_attn_fwd/bwd (layers.py:114-130),
hual_predictor_fwd/bwd (modules.py:143-160).  Tolerance 1e-3 of each tensor's scale; span indices equal."""
import ctypes

import numpy as np
import pytest
import torch

import parity_util as pu
from oracle import philox as px
from oracle import seqpan_ref as R

pytestmark = pytest.mark.gpu
DROP, SEED, OFFSET = 0.2, 5, 7


class Block:
    """a model + one batch + everything a block call needs"""

    def __init__(self, **shape):
        from hual_amd import lib
        self.lib = lib
        self.l = lib.load()
        self.cfg, self.p, self.wv, self.b, self.labels = pu.make_case(**shape)
        b = self.b
        self.B, self.T = b['video'].shape[:2]
        self.L = b['word_ids'].shape[1]
        self.C = b['char_ids'].shape[2]
        self.Nv, self.Nq = self.B * self.T, self.B * self.L
        self.R = self.Nv + self.Nq
        self.m = pu.hip_model(self.cfg, self.p, self.wv)
        self.m.set_rng(SEED, OFFSET)
        self.bt, self.keep, _ = self.m._prep(b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy())
        self.ws = self.m._workspace(self.B, self.T, self.L, self.C)
        self.opts = self.m._opts(DROP)
        self.dev = self.m.device
        self.rng = px.DropoutRNG(SEED, OFFSET, DROP)
        self.rows_v = np.arange(self.Nv)
        self.rows_q = self.Nv + np.arange(self.Nq)
        self.v_mask = (torch.arange(self.T).unsqueeze(0) < b['lens'].long().unsqueeze(1)).to(torch.int32)
        self.q_mask = (b['word_ids'] != 0).to(torch.int32)
        self.grads = torch.full_like(self.m.params, 3.0)       # overwritten by every *_bwd call

    def args(self):
        lib = self.lib
        return [ctypes.byref(self.m.cfg), lib.ptr(self.m.params), ctypes.byref(self.bt), ctypes.byref(self.opts)]

    def tail(self):
        return [self.lib.ptr(self.ws), self.ws.numel(), self.lib.stream_ptr()]

    def rand(self, rows, seed, scale=1.0):
        return (torch.randn(rows, 128, generator=torch.Generator().manual_seed(seed)) * scale)

    def split(self, x):
        return x[:self.Nv].reshape(self.B, self.T, 128), x[self.Nv:].reshape(self.B, self.L, 128)

    def params_grad(self):
        return self.m.table.unpack(self.grads.cpu().numpy())

    def oracle_params(self):
        return {k: t.detach().clone().requires_grad_(True) for k, t in self.p.items()}


def _close(name, got, want, tol=1e-3):
    got, want = torch.as_tensor(got).double().reshape(-1), torch.as_tensor(want).double().reshape(-1)
    d, sc = float((got - want).abs().max()), float(want.abs().max())
    assert d <= tol * max(sc, 1e-3), (name, d, sc)


def _check_param_grads(blk, pr, prefix_ok):
    """every parameter gradient of the block within 1e-3 of its scale; gradients that are zero in exact arithmetic (the key
    biases under the softmax) are rounding noise and are held to 1e-6 of the block's largest gradient instead; nothing
    outside the block may receive a gradient"""
    hg = blk.params_grad()
    gmax = max(float(t.grad.abs().max()) for t in pr.values() if t.grad is not None)
    for k, t in pr.items():
        g = t.grad
        if g is None:
            assert float(np.abs(hg[k]).max()) == 0.0, ('gradient outside the block', k)
            continue
        assert prefix_ok(k), k
        d, sc = float(np.abs(hg[k].reshape(-1) - g.numpy().reshape(-1)).max()), float(g.abs().max())
        assert d <= 1e-3 * max(sc, 1e-3 * gmax), ('grad ' + k, d, sc, gmax)


SHAPES = [dict(B=3, T=37, L=9, C=4, seed=11, max_vlen=40), dict(B=4, T=64, L=20, C=6, seed=9, max_vlen=64)]


@pytest.mark.parametrize('shape', SHAPES)
def test_video_proj_ln_fwd(shape):
    blk = Block(**shape)
    lib = blk.lib
    x0 = torch.empty(blk.R, 128, device=blk.dev)
    lib.check(blk.l.hual_video_proj_ln_fwd(ctypes.byref(blk.m.cfg), lib.ptr(blk.m.params), lib.ptr(blk.m.word_table), ctypes.byref(blk.bt),
                                           ctypes.byref(blk.opts), lib.ptr(x0), *blk.tail()))
    b = blk.b
    out = R.forward(blk.p, blk.cfg, blk.wv, b['video'], b['lens'], b['word_ids'], b['char_ids'], drop_rate=DROP, seed=SEED, offset=OFFSET,
                    want_tap=True)
    ref = torch.cat([out['tap']['cb.x0.v'].reshape(blk.Nv, 128), out['tap']['cb.x0.q'].reshape(blk.Nq, 128)])
    _close('x0', x0.cpu(), ref)


@pytest.mark.parametrize('shape', SHAPES)
def test_conv_block_fwd_bwd(shape):
    blk = Block(**shape)
    lib = blk.lib
    x, dy = blk.rand(blk.R, 1), blk.rand(blk.R, 2)
    xd, dyd = x.to(blk.dev), dy.to(blk.dev)
    y, dx = torch.empty_like(xd), torch.empty_like(xd)
    lib.check(blk.l.hual_conv_block_fwd(*blk.args(), lib.ptr(xd), lib.ptr(y), *blk.tail()))
    lib.check(blk.l.hual_conv_block_bwd(*blk.args(), lib.ptr(dyd), lib.ptr(dx), lib.ptr(blk.grads), *blk.tail()))
    torch.cuda.synchronize()
    pins_v = [blk.m.tap_bits('cb.rb%d' % i)[:blk.Nv].reshape(blk.B, blk.T, -1) for i in range(4)]
    pins_q = [blk.m.tap_bits('cb.rb%d' % i)[blk.Nv:].reshape(blk.B, blk.L, -1) for i in range(4)]
    pr = blk.oracle_params()
    xr = x.clone().requires_grad_(True)
    xv, xq = blk.split(xr)
    tv, tq = {}, {}
    yv = R.conv_block(xv, pr, 'conv_block', blk.rng, px.SITE_CONV, blk.rows_v, tv, 'cb', pins_v)
    yq = R.conv_block(xq, pr, 'conv_block', blk.rng, px.SITE_CONV, blk.rows_q, tq, 'cb', pins_q)
    # the pins are the kernel's own active sets: audit them against the oracle's pre-activations
    pu.assert_pins_ok([('v%d' % i, tv['cb.z%d' % i], pins_v[i]) for i in range(4)] +
                      [('q%d' % i, tq['cb.z%d' % i], pins_q[i]) for i in range(4)])
    ref = torch.cat([yv.reshape(blk.Nv, 128), yq.reshape(blk.Nq, 128)])
    _close('y', y.cpu(), ref.detach())
    ref.backward(dy)
    _close('dx', dx.cpu(), xr.grad)
    _check_param_grads(blk, pr, lambda k: k.startswith('conv_block/'))


# clips of more than 128 frames: the backward's large job runs on eight waves (csrc/attn.hip attn_bwd_big_kernel): all eight key blocks,
# six of them (two idle waves, three waves without a partner on their dQ slot), an odd number of query pairs
DA_LONG_SHAPES = [dict(B=2, T=256, L=20, C=4, seed=41, max_vlen=256), dict(B=3, T=170, L=9, C=4, seed=42, max_vlen=192),
                  dict(B=2, T=131, L=12, C=4, seed=43, max_vlen=160)]


@pytest.mark.parametrize('shape', SHAPES + DA_LONG_SHAPES)
@pytest.mark.parametrize('layer', [0, 1])
def test_dual_attn_fwd_bwd(shape, layer):
    blk = Block(**shape)
    lib = blk.lib
    x, dy = blk.rand(blk.R, 3), blk.rand(blk.R, 4)
    xd, dyd = x.to(blk.dev), dy.to(blk.dev)
    y, dx = torch.empty_like(xd), torch.empty_like(xd)
    lib.check(blk.l.hual_dual_attn_fwd(*blk.args(), layer, lib.ptr(xd), lib.ptr(y), *blk.tail()))
    lib.check(blk.l.hual_dual_attn_bwd(*blk.args(), layer, lib.ptr(dyd), lib.ptr(dx), lib.ptr(blk.grads), *blk.tail()))
    pr = blk.oracle_params()
    xr = x.clone().requires_grad_(True)
    v, q = blk.split(xr)
    n, site = 'd_attn_%d' % layer, px.SITE_DA + 8 * layer
    H = blk.cfg.num_heads
    v_ = R.dual_attn_block(v, q, pr, n, H, blk.v_mask, blk.q_mask, blk.rng, site, blk.rows_v)
    q_ = R.dual_attn_block(q, v, pr, n, H, blk.q_mask, blk.v_mask, blk.rng, site, blk.rows_q)
    ref = torch.cat([v_.reshape(blk.Nv, 128), q_.reshape(blk.Nq, 128)])
    _close('y', y.cpu(), ref.detach())
    ref.backward(dy)
    _close('dx', dx.cpu(), xr.grad)
    _check_param_grads(blk, pr, lambda k: k.startswith(n + '/'))


# csrc/cqwide.hip (queries of at most 32 words): clips of more than 128 frames (16 waves, two 128-column blocks: both full, a ragged
# second block, three words / 32 words); the shapes beyond it - queries of more than 32 words - run the staged kernels of csrc/cq.hip
CQ_WIDE_SHAPES = [dict(B=3, T=256, L=20, C=4, seed=21, max_vlen=256), dict(B=2, T=170, L=32, C=4, seed=22, max_vlen=192),
                  dict(B=9, T=129, L=3, C=4, seed=23, max_vlen=160), dict(B=5, T=100, L=30, C=4, seed=33, max_vlen=100),
                  dict(B=3, T=64, L=40, C=4, seed=24, max_vlen=64),
                  # score matrices beyond every LDS form: the global-operand kernels with the matrices in global memory too
                  dict(B=2, T=256, L=40, C=4, seed=61, max_vlen=256), dict(B=2, T=128, L=100, C=4, seed=64, max_vlen=128)]


@pytest.mark.parametrize('shape', SHAPES + CQ_WIDE_SHAPES)
def test_cq_attn_fwd_bwd(shape):
    blk = Block(**shape)
    lib = blk.lib
    x, dy = blk.rand(blk.R, 5), blk.rand(blk.R, 6)
    xd, dyd = x.to(blk.dev), dy.to(blk.dev)
    feats, dx = torch.empty_like(xd), torch.empty_like(xd)
    lib.check(blk.l.hual_cq_attn_fwd(*blk.args(), lib.ptr(xd), lib.ptr(feats), *blk.tail()))
    lib.check(blk.l.hual_cq_attn_bwd(*blk.args(), lib.ptr(dyd), lib.ptr(dx), lib.ptr(blk.grads), *blk.tail()))
    pr = blk.oracle_params()
    xr = x.clone().requires_grad_(True)
    v, q = blk.split(xr)
    q2v = R.cq_attention(v, q, blk.v_mask, blk.q_mask, pr, 'q2v_attn', blk.rng, px.SITE_TRI + 0, blk.rows_v, px.SITE_TRI + 1, blk.rows_q)
    v2q = R.cq_attention(q, v, blk.q_mask, blk.v_mask, pr, 'v2q_attn', blk.rng, px.SITE_TRI + 2, blk.rows_q, px.SITE_TRI + 3, blk.rows_v)
    ref = torch.cat([q2v.reshape(blk.Nv, 128), v2q.reshape(blk.Nq, 128)])
    _close('feats', feats.cpu(), ref.detach())
    ref.backward(dy)
    _close('dx', dx.cpu(), xr.grad)
    _check_param_grads(blk, pr, lambda k: k.startswith('q2v_attn/') or k.startswith('v2q_attn/'))


@pytest.mark.parametrize('shape', SHAPES)
def test_predictor_fwd_bwd(shape):
    blk = Block(**shape)
    lib = blk.lib
    B, T, Nv = blk.B, blk.T, blk.Nv
    x = blk.rand(Nv, 7) * blk.v_mask.reshape(Nv, 1).float()             # `outputs` is masked (model.py:97)
    g = torch.Generator().manual_seed(8)
    ds, de = torch.randn(B, T, generator=g), torch.randn(B, T, generator=g)
    xd = x.to(blk.dev)
    s_log, e_log = torch.empty(B, T, device=blk.dev), torch.empty(B, T, device=blk.dev)
    si, ei = torch.empty(B, dtype=torch.int64, device=blk.dev), torch.empty(B, dtype=torch.int64, device=blk.dev)
    dx = torch.empty_like(xd)
    lib.check(blk.l.hual_predictor_fwd(*blk.args(), lib.ptr(xd), lib.ptr(s_log), lib.ptr(e_log), lib.ptr(si), lib.ptr(ei), *blk.tail()))
    dsd, ded = ds.to(blk.dev), de.to(blk.dev)
    lib.check(blk.l.hual_predictor_bwd(*blk.args(), lib.ptr(dsd), lib.ptr(ded), lib.ptr(dx), lib.ptr(blk.grads), *blk.tail()))
    torch.cuda.synchronize()
    pin = {}
    for ps in range(2):
        pin['fe%d' % ps] = [blk.m.tap_bits('fe%d.rb%d' % (ps, i)).reshape(B, T, -1) for i in range(4)]
    pin['head.hs'] = (blk.m.tap('head.hs').cpu() > 0).reshape(B, T, -1)
    pin['head.he'] = (blk.m.tap('head.he').cpu() > 0).reshape(B, T, -1)
    pr = blk.oracle_params()
    xr = x.clone().requires_grad_(True)
    tap = {}
    s_ref, e_ref = R.conditioned_predictor(xr.reshape(B, T, 128), pr, blk.cfg.num_heads, blk.v_mask, blk.rng, blk.rows_v, tap, pin)
    pu.assert_pins_ok([('fe%d.z%d' % (ps, i), tap['fe%d.z%d' % (ps, i)], pin['fe%d' % ps][i]) for ps in range(2) for i in range(4)] +
                      [('head.zs', tap['head.zs'], pin['head.hs']), ('head.ze', tap['head.ze'], pin['head.he'])])
    _close('start_logits', s_log.cpu(), s_ref.detach())
    _close('end_logits', e_log.cpu(), e_ref.detach())
    # the span argmax on the kernel's own logits (rounding-level differences in the logits may move a near tie)
    si_ref, ei_ref = R.ans_predictor(s_log.cpu(), e_log.cpu(), blk.v_mask)
    assert torch.equal(si.cpu(), si_ref) and torch.equal(ei.cpu(), ei_ref)
    ((s_ref * ds).sum() + (e_ref * de).sum()).backward()
    _close('d_outputs', dx.cpu(), xr.grad)
    _check_param_grads(blk, pr, lambda k: k.startswith('predictor/'))

"""GPU: per-kernel entry points against plain PyTorch fp32/fp64 references of the same op
(layer_norm layers.py:7-17, attention core layers.py:80-96, ans_predictor layers.py:194-203)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    return torch.device('cuda:0')


@pytest.mark.parametrize('R', [1, 37, 9472])
def test_layer_norm_fwd(dev, R):
    from hual_amd import lib
    g = torch.Generator().manual_seed(R)
    x = (torch.randn(R, 128, generator=g) * 3 + 0.5).to(dev)
    gamma, beta = torch.randn(128, generator=g).to(dev), torch.randn(128, generator=g).to(dev)
    y, mean, rstd = torch.empty_like(x), torch.empty(R, device=dev), torch.empty(R, device=dev)
    lib.check(lib.load().hual_layer_norm_fwd(lib.ptr(x), lib.ptr(gamma), lib.ptr(beta), lib.ptr(y), lib.ptr(mean), lib.ptr(rstd),
                                             R, lib.stream_ptr()))
    xd = x.double()
    mu = xd.mean(-1, keepdim=True)
    var = ((xd - mu) ** 2).mean(-1, keepdim=True)               # biased variance, eps inside the rsqrt
    ref = (xd - mu) * torch.rsqrt(var + 1e-6) * gamma.double() + beta.double()
    assert (y.double() - ref).abs().max().item() < 2e-5
    assert (mean.double() - mu[:, 0]).abs().max().item() < 1e-5


@pytest.mark.parametrize('B,Tq,Tk', [(2, 16, 16), (3, 37, 9), (2, 128, 128), (1, 20, 256)])
def test_attention_fwd(dev, B, Tq, Tk):
    from hual_amd import lib
    g = torch.Generator().manual_seed(B * 1000 + Tq + Tk)
    Q = torch.randn(B * Tq, 128, generator=g).to(dev)
    K = torch.randn(B * Tk, 128, generator=g).to(dev)
    V = torch.randn(B * Tk, 128, generator=g).to(dev)
    qlen = torch.randint(1, Tq + 1, (B,), generator=g)
    klen = torch.randint(1, Tk + 1, (B,), generator=g)
    qm = (torch.arange(Tq)[None, :] < qlen[:, None]).float().reshape(-1).to(dev)
    km = (torch.arange(Tk)[None, :] < klen[:, None]).float().reshape(-1).to(dev)
    O = torch.empty(B * Tq, 128, device=dev)
    lib.check(lib.load().hual_attention_fwd(lib.ptr(Q), 128, lib.ptr(K), lib.ptr(V), 128, lib.ptr(O), 128, B, Tq, Tk, lib.ptr(qm),
                                            lib.ptr(km), lib.stream_ptr()))
    q = Q.double().view(B, Tq, 8, 16).transpose(1, 2)
    k = K.double().view(B, Tk, 8, 16).transpose(1, 2)
    v = V.double().view(B, Tk, 8, 16).transpose(1, 2)
    s = q @ k.transpose(-1, -2) / 4.0
    mask = qm.double().view(B, 1, Tq, 1) * km.double().view(B, 1, 1, Tk)
    s = s + (1.0 - mask) * (-1e30)                               # additive mask: fully masked rows become uniform
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * Tq, 128)
    assert (O.double() - ref).abs().max().item() < 2e-5


def _span_ref(s, e, m):
    """ans_predictor of the oracle (oracle/seqpan_ref.py, layers.py:194-203) on CPU float32"""
    from oracle import seqpan_ref as R
    return R.ans_predictor(s.cpu(), e.cpu(), m.cpu())


def _span_hip(dev, s, e, m):
    from hual_amd import lib
    B, T = s.shape
    si = torch.empty(B, dtype=torch.int64, device=dev)
    ei = torch.empty(B, dtype=torch.int64, device=dev)
    lib.check(lib.load().hual_span_argmax(lib.ptr(s), lib.ptr(e), lib.ptr(m), lib.ptr(si), lib.ptr(ei), B, T, lib.stream_ptr()))
    return si.cpu(), ei.cpu()


@pytest.mark.parametrize('B,T', [(4, 7), (16, 64), (3, 256), (64, 128)])
def test_span_argmax_bit_exact(dev, B, T):
    g = torch.Generator().manual_seed(B + T)
    s = (torch.randn(B, T, generator=g) * 3).to(dev)
    e = (torch.randn(B, T, generator=g) * 3).to(dev)
    s[0, :3] = s[0, 0]                                           # ties: first index wins
    lens = torch.randint(1, T + 1, (B,), generator=g)
    lens[0] = T
    m = (torch.arange(T)[None, :] < lens[:, None]).float().to(dev)
    si, ei = _span_hip(dev, s, e, m)
    rs, re_ = _span_ref(s, e, m)
    assert (si <= ei).all()
    assert torch.equal(si, rs) and torch.equal(ei, re_)


def test_span_argmax_near_ties_bit_exact(dev):
    """candidates whose products differ by a few float32 ulps or not at all: the winner is decided by the rounding of
    exp, of the softmax denominator and of the product, which the kernel and the oracle perform identically
    (correctly rounded exp, order-independent denominator, IEEE division and product)"""
    B, T = 512, 96
    g = torch.Generator().manual_seed(2024)
    base_s = torch.randn(B, 1, generator=g) * 2
    base_e = torch.randn(B, 1, generator=g) * 2
    # logits = a common level + a few ulps of jitter: every (i <= j) pair is a near tie
    s = base_s + torch.randint(-3, 4, (B, T), generator=g).float() * 2.0 ** -21
    e = base_e + torch.randint(-3, 4, (B, T), generator=g).float() * 2.0 ** -21
    # half of the batch: two plateaus of exactly equal logits far apart (exact ties -> first index)
    s[B // 2:, 10:20] = s[B // 2:, 10:11]
    s[B // 2:, 60:70] = s[B // 2:, 10:11]
    e[B // 2:, 30:40] = e[B // 2:, 30:31]
    e[B // 2:, 80:90] = e[B // 2:, 30:31]
    lens = torch.randint(T // 2, T + 1, (B,), generator=g)
    m = (torch.arange(T)[None, :] < lens[:, None]).float()
    s, e, m = s.to(dev), e.to(dev), m.to(dev)
    si, ei = _span_hip(dev, s, e, m)
    rs, re_ = _span_ref(s, e, m)
    assert torch.equal(si, rs) and torch.equal(ei, re_)

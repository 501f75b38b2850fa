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


@pytest.mark.parametrize('B,T', [(4, 7), (16, 64), (3, 256)])
def test_span_argmax_bit_exact(dev, B, T):
    from hual_amd import lib
    g = torch.Generator().manual_seed(B + T)
    s = (torch.randn(B, T, generator=g) * 3).to(dev)
    e = (torch.randn(B, T, generator=g) * 3).to(dev)
    s[0, :3] = s[0, 0]                                           # ties: first index wins
    lens = torch.randint(1, T + 1, (B,), generator=g)
    lens[0] = T
    m = (torch.arange(T)[None, :] < lens[:, None]).float().to(dev)
    si = torch.empty(B, dtype=torch.int64, device=dev)
    ei = torch.empty(B, dtype=torch.int64, device=dev)
    lib.check(lib.load().hual_span_argmax(lib.ptr(s), lib.ptr(e), lib.ptr(m), lib.ptr(si), lib.ptr(ei), B, T, lib.stream_ptr()))
    # float32 reference with the reference's op order (layers.py:194-203)
    sm = s.cpu() * m.cpu() + (-1e30) * (1.0 - m.cpu())
    em = e.cpu() * m.cpu() + (-1e30) * (1.0 - m.cpu())
    ps, pe = torch.softmax(sm, 1), torch.softmax(em, 1)
    outer = torch.triu(ps.unsqueeze(2) * pe.unsqueeze(1))
    rs = torch.max(torch.max(outer, dim=2)[0], dim=1)[1]
    re_ = torch.max(torch.max(outer, dim=1)[0], dim=1)[1]
    assert (si.cpu() <= ei.cpu()).all()
    # the kernel's softmax uses its own exp: indices must agree wherever the top two candidates differ by more than rounding
    for b in range(B):
        if int(si[b]) != int(rs[b]) or int(ei[b]) != int(re_[b]):
            rowmax = torch.max(outer[b], dim=1)[0]
            colmax = torch.max(outer[b], dim=0)[0]
            assert abs(float(rowmax[int(si[b])] - rowmax[int(rs[b])])) <= 1e-6 * float(rowmax.max())
            assert abs(float(colmax[int(ei[b])] - colmax[int(re_[b])])) <= 1e-6 * float(colmax.max())

"""GPU: per-kernel entry points against plain PyTorch fp32/fp64 references of the same op
(layer_norm layers.py:7-17, attention core layers.py:80-96, ans_predictor layers.py:194-203)."""
import ctypes

import numpy as np
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


def _attn_case(dev, B, Tq, Tk, seed):
    g = torch.Generator().manual_seed(seed)
    Q = torch.randn(B * Tq, 128, generator=g).to(dev)
    K = torch.randn(B * Tk, 128, generator=g).to(dev)
    V = torch.randn(B * Tk, 128, generator=g).to(dev)
    qlen = torch.randint(1, Tq + 1, (B,), generator=g)
    klen = torch.randint(1, Tk + 1, (B,), generator=g)
    qlen[0], klen[0] = Tq, Tk
    qm = (torch.arange(Tq)[None, :] < qlen[:, None]).float().reshape(-1).to(dev)
    km = (torch.arange(Tk)[None, :] < klen[:, None]).float().reshape(-1).to(dev)
    return Q, K, V, qm, km


def _attn_ref(Q, K, V, qm, km, B, Tq, Tk, drop=None):
    """float64 restatement of layers.py:80-96 for merged heads; drop = (seed, offset, rate, site) uses the oracle's 16-bit
    Philox decisions (oracle/philox.py mask_attn) with RNG row = query row * 8 + head"""
    q = Q.view(B, Tq, 8, 16).transpose(1, 2)
    k = K.view(B, Tk, 8, 16).transpose(1, 2)
    v = V.view(B, Tk, 8, 16).transpose(1, 2)
    s = q @ k.transpose(-1, -2) / 4.0
    mask = qm.view(B, 1, Tq, 1) * km.view(B, 1, 1, Tk)
    s = s + (1.0 - mask) * (-1e30)                               # additive mask: fully masked rows become uniform
    pr = torch.softmax(s, -1)
    if drop is not None:
        from oracle import philox as px
        seed, offset, rate, site = drop
        rng = px.DropoutRNG(seed, offset, rate)
        rid = (np.arange(B * Tq).reshape(B, 1, Tq) * 8 + np.arange(8).reshape(1, 8, 1)).reshape(-1)
        m = torch.from_numpy(rng.mask_attn(site, rid, Tk)).to(pr.dtype).reshape(B, 8, Tq, Tk).to(pr.device)
        pr = pr * m
    return (pr @ v).transpose(1, 2).reshape(B * Tq, 128)


@pytest.mark.parametrize('B,Tq,Tk', [(2, 16, 16), (3, 37, 9), (2, 128, 128), (1, 20, 256), (2, 100, 30), (1, 256, 256), (3, 1, 1)])
def test_attention_fwd(dev, B, Tq, Tk):
    """fp16-pair products (three passes, 22-bit operands) + fp32 softmax: at the level of a float32 PyTorch evaluation of the same
    function (3e-7 .. 1.4e-6 measured on these cases against 3e-7 .. 7e-7, scripts/exp/attn_err.py); gate 6e-6 on O(1) outputs (bar: 1e-3)"""
    from hual_amd import lib
    Q, K, V, qm, km = _attn_case(dev, B, Tq, Tk, B * 1000 + Tq + Tk)
    O = torch.empty(B * Tq, 128, device=dev)
    lib.check(lib.load().hual_attention_fwd(lib.ptr(Q), 128, lib.ptr(K), lib.ptr(V), 128, lib.ptr(O), 128, B, Tq, Tk, lib.ptr(qm),
                                            lib.ptr(km), lib.stream_ptr()))
    ref = _attn_ref(Q.double(), K.double(), V.double(), qm.double(), km.double(), B, Tq, Tk)
    assert (O.double() - ref).abs().max().item() < 6e-6


def test_attention_fwd_rejects_more_than_256_queries(dev):
    """a launch's unit codes hold 16 query tiles per job: Tq > 256 is refused with an error code (never launched), in the
    forward as in the backward (include/hual_seqpan.h)"""
    from hual_amd import lib
    B, Tq, Tk = 1, 272, 32
    Q, K, V, qm, km = _attn_case(dev, B, Tq, Tk, 5)
    O = torch.empty(B * Tq, 128, device=dev)
    rc = lib.load().hual_attention_fwd(lib.ptr(Q), 128, lib.ptr(K), lib.ptr(V), 128, lib.ptr(O), 128, B, Tq, Tk, lib.ptr(qm),
                                       lib.ptr(km), lib.stream_ptr())
    assert rc < 0 and b'Tq <= 256' in lib.load().hual_last_error()
    with pytest.raises(lib.HualError):
        lib.check(rc)


@pytest.mark.parametrize('B,Tq,Tk', [(2, 16, 16), (3, 37, 9), (2, 128, 128), (1, 20, 256), (2, 128, 20), (1, 256, 20), (2, 200, 32), (2, 100, 100), (1, 256, 256), (3, 1, 1)])
@pytest.mark.parametrize('rate', [0.0, 0.2])
def test_attention_fwd_bwd_with_dropout(dev, B, Tq, Tk, rate):
    """hual_attention_fwd_save + hual_attention_bwd against float64 autograd of the same function with the oracle's dropout
    mask: the kernels drop exactly the same probabilities; gradients within 3e-6 of each tensor's scale (measured 1-7e-7, the level of
    float32 PyTorch autograd on the same function: round 5 moved the kernels from bf16 pairs - 2e-5 .. 1e-4 here - to fp16 pairs)"""
    from hual_amd import lib
    l = lib.load()
    Q, K, V, qm, km = _attn_case(dev, B, Tq, Tk, 77 + B * 1000 + Tq + Tk)
    g = torch.Generator().manual_seed(5)
    dO = torch.randn(B * Tq, 128, generator=g).to(dev)
    seed, offset, site = 0x1234567800000042, 3, 9
    rng_state = torch.tensor(np.array([seed & 0xffffffff, seed >> 32, offset], dtype=np.uint32).view(np.int32)).to(dev)
    ldm = l.hual_attention_keep_row_bytes(Tk)
    keep = torch.zeros(B * Tq * 8, ldm, dtype=torch.uint8, device=dev)
    stats = torch.zeros(2, B * Tq * 8, device=dev)
    O = torch.empty(B * Tq, 128, device=dev)
    lib.check(l.hual_attention_fwd_save(lib.ptr(Q), 128, lib.ptr(K), lib.ptr(V), 128, lib.ptr(O), 128, B, Tq, Tk, lib.ptr(qm), lib.ptr(km),
                                        lib.ptr(stats), lib.ptr(keep), ldm, lib.ptr(rng_state), rate, site, lib.stream_ptr()))
    dQ, dK, dV = torch.full_like(Q, 7.0), torch.full_like(K, 7.0), torch.full_like(V, 7.0)      # written, not accumulated
    lib.check(l.hual_attention_bwd(lib.ptr(Q), 128, lib.ptr(K), lib.ptr(V), 128, lib.ptr(O), 128, lib.ptr(stats), lib.ptr(keep), ldm,
                                   lib.ptr(dO), 128, lib.ptr(dQ), 128, lib.ptr(dK), lib.ptr(dV), 128, B, Tq, Tk, lib.ptr(qm), lib.ptr(km),
                                   lib.ptr(rng_state), rate, site, lib.stream_ptr()))
    Qd, Kd, Vd = (t.double().cpu().requires_grad_(True) for t in (Q, K, V))
    ref = _attn_ref(Qd, Kd, Vd, qm.double().cpu(), km.double().cpu(), B, Tq, Tk, (seed, offset, rate, site) if rate > 0 else None)
    # (6e-6, not 3e-6: an operand element whose residual lies below fp16's normal range - 1.5 % of them - loses it in the matrix pipe:
    #  up to 2^-14 / 16 = 3.8e-6 absolute on that element, visible one to one where a single key carries the whole probability)
    assert (O.double().cpu() - ref.detach()).abs().max().item() < 6e-6
    ref.backward(dO.double().cpu())
    for name, got, want in (('dQ', dQ, Qd.grad), ('dK', dK, Kd.grad), ('dV', dV, Vd.grad)):
        err = (got.double().cpu() - want).abs().max().item()
        # (+ 3e-6 absolute: with a single key the softmax is constant and dQ, dK are rounding noise around an exact zero - dP - delta
        #  with delta = dO . O from the forward's own rounding of O)
        assert err <= 3e-6 * want.abs().max().item() + 3e-6, (name, err, want.abs().max().item())


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

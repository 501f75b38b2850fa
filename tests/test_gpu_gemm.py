"""GPU parity of the MFMA GEMM kernels against a plain torch fp32/fp64 matmul (conv1d k=1, layers.py:20-29)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need the MI355X box'
    return torch.device('cuda:0')


def _ref(A, W, b, act):
    y = A.double() @ W.double()
    if b is not None:
        y = y + b.double()
    if act == 1:
        y = torch.relu(y)
    elif act == 2:
        y = torch.sigmoid(y)
    return y


# G: workgroups of the persistent weight-gradient launch of the training step (0 = one per CU)
@pytest.mark.parametrize('M,K,N,G', [(64, 64, 128, 1), (37, 128, 128, 2), (1000, 400, 128, 7), (9472, 128, 128, 512),
                                     (4096, 1024, 128, 100), (513, 16, 128, 512), (33, 272, 128, 3), (9472, 128, 128, 0),
                                     (1000, 400, 128, 0), (64, 64, 128, 0)])
def test_linear_dw(dev, M, K, N, G):
    from hual_amd import lib
    g = torch.Generator(device='cpu').manual_seed(M + K + N)
    A = torch.randint(-2, 3, (M, K), generator=g).float().to(dev)
    dY = torch.randint(-2, 3, (M, N), generator=g).float().to(dev)
    dW = torch.zeros(K, N, device=dev)
    db = torch.zeros(N, device=dev)
    lib.linear_dw(A, dY, dW, db, workgroups=G)
    assert torch.equal(dW, A.t() @ dY)          # small integers: exact in fp32 regardless of summation order
    assert torch.equal(db, dY.sum(0))
    A = torch.randn(M, K, generator=g).to(dev)
    dY = torch.randn(M, N, generator=g).to(dev)
    dW.zero_(); db.zero_()
    lib.linear_dw(A, dY, dW, db, workgroups=G)
    ref = A.double().t() @ dY.double()
    # fp16-pair operands (round 5): 2-5e-7 of the largest entry measured (float32 torch.matmul: 1-3e-6), scripts/exp/dw_err.py
    assert (dW.double() - ref).abs().max().item() < 2e-6 * ref.abs().max().item()


@pytest.mark.parametrize('M,K', [(9472, 128), (4096, 1024), (1000, 400)])
@pytest.mark.parametrize('mode', ['grad_up', 'grad_down', 'grad_spike', 'tiny'])
def test_linear_dw_gradient_magnitudes(dev, M, K, mode):
    """the weight-gradient launch keeps a RUNNING power-of-two scale for dY (fp16 pairs: the contraction runs over all rows of a
    workgroup's run, csrc/gemm.hip dw_f16_segment): gradients that grow along the rows (the scale is lowered again and again and the
    accumulators follow), that shrink, that spike by 1e4 in one row, and that sit at 1e-12 - all within 2e-6 of the largest entry of
    the float64 product (measured 2-6e-7; float32 torch.matmul 0.3-7e-6)"""
    from hual_amd import lib
    g = torch.Generator(device='cpu').manual_seed(M + K)
    A = torch.randn(M, K, generator=g)
    dY = torch.randn(M, 128, generator=g)
    r = torch.arange(M).float() / M
    if mode == 'grad_up':
        dY = dY * (1e-7 * 10 ** (5 * r))[:, None]
    elif mode == 'grad_down':
        dY = dY * (1e-2 * 10 ** (-5 * r))[:, None]
    elif mode == 'grad_spike':
        dY = dY * 1e-5
        dY[M // 2] *= 1e4
    else:
        dY = dY * 1e-12
    A, dY = A.to(dev), dY.to(dev)
    dW = torch.zeros(K, 128, device=dev)
    db = torch.zeros(128, device=dev)
    lib.linear_dw(A, dY, dW, db, workgroups=0)
    ref = A.double().t() @ dY.double()
    assert torch.isfinite(dW).all()
    assert (dW.double() - ref).abs().max().item() < 2e-6 * ref.abs().max().item()
    assert (db.double() - dY.double().sum(0)).abs().max().item() <= 1e-5 * dY.double().abs().sum(0).max().item()


@pytest.mark.parametrize('mode', ['large', 'rising', 'tiny'])
def test_linear_dw_activation_magnitudes(dev, mode):
    """the A side of a weight gradient is not bounded by construction (products of two activations, unnormalised block outputs): operands
    far beyond the 4094 that a fixed 2^4 fp16 scale could hold, a magnitude that rises by 2^20 along the rows of a segment (the running
    scale is lowered on the way and the accumulators with it), and operands of 1e-10"""
    from hual_amd import lib
    M, K = 1500, 256
    g = torch.Generator(device='cpu').manual_seed(77)
    A = torch.randn(M, K, generator=g)
    dY = torch.randn(M, 128, generator=g) * 1e-3
    if mode == 'large':
        A = A * 3e5
    elif mode == 'rising':
        A = A * torch.logspace(-2, 4, M).unsqueeze(1)
    else:
        A = A * 1e-10
    A, dY = A.to(dev), dY.to(dev)
    dW = torch.zeros(K, 128, device=dev)
    db = torch.zeros(128, device=dev)
    lib.linear_dw(A, dY, dW, db, workgroups=0)
    ref = A.double().t() @ dY.double()
    assert torch.isfinite(dW).all()
    assert (dW.double() - ref).abs().max().item() < 2e-6 * ref.abs().max().item()


@pytest.mark.parametrize('M,K', [(16, 64), (37, 128), (1000, 400), (2048, 1024), (9472, 128), (100, 256), (48, 8)])
@pytest.mark.parametrize('act', [0, 1])
def test_linear_bf16x3_forward(dev, M, K, act):
    """split dense (three MFMA passes on fp16 hi/lo operands with per-row power-of-two scaling, bf16x3.h "f16x3"): exact
    on small integers, fp32-level accuracy on random data"""
    from hual_amd import lib
    g = torch.Generator(device='cpu').manual_seed(M * 3 + K + act)
    Ai = torch.randint(-3, 4, (M, K), generator=g).float().to(dev)
    Wi = torch.randint(-3, 4, (K, 128), generator=g).float().to(dev)
    assert torch.equal(lib.linear_bf16x3(Ai, Wi), Ai @ Wi)
    A = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(K, 128, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(128, generator=g).to(dev)
    Y = lib.linear_bf16x3(A, W, b, act=act)
    ref = _ref(A, W, b, act)
    err = (Y.double() - ref).abs().max().item()
    assert err < 2e-6 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize('K', [128, 256, 400])
def test_linear_split_dynamic_range(dev, K):
    """rows whose magnitudes span 1e-12 .. 1e6 (gradient-like and feature-like operands) and entries spanning 2^20 inside a
    row: every output row is accurate relative to its OWN scale - the fp16 operands are scaled per row and per 128-deep chunk"""
    from hual_amd import lib
    g = torch.Generator(device='cpu').manual_seed(K)
    M = 600
    A = torch.randn(M, K, generator=g)
    A = A * torch.exp2(torch.randint(-10, 11, (M, K), generator=g).float())          # wide range inside a row
    rowscale = torch.exp2(torch.randint(-40, 21, (M, 1), generator=g).float())        # and across rows
    A = (A * rowscale).to(dev)
    W = (torch.randn(K, 128, generator=g) / K ** 0.5).to(dev)
    Y = lib.linear_bf16x3(A, W)
    ref = A.double() @ W.double()
    scale = (A.double().abs() @ W.double().abs()).max(dim=1, keepdim=True).values      # sum of |terms|: the natural error scale
    rel = ((Y.double() - ref).abs() / scale).max().item()
    assert torch.isfinite(Y).all()
    assert rel < 1e-6, rel
    # an all-zero row and a row of tiny denormal-range values stay finite and exact / accurate
    A2 = torch.zeros(32, K, device=dev)
    A2[1] = 1e-38
    Y2 = lib.linear_bf16x3(A2, W)
    assert torch.isfinite(Y2).all() and float(Y2[0].abs().max()) == 0.0


@pytest.mark.parametrize('M,N', [(37, 128), (500, 256), (9472, 128), (100, 512), (300, 400), (64, 1024)])
def test_linear_bf16x3_transposed(dev, M, N):
    """dX[M,N] = dY[M,128] . W^T with W stored [N,128], through the image of the transposed weight"""
    from hual_amd import lib
    g = torch.Generator(device='cpu').manual_seed(M + N)
    dY = torch.randint(-3, 4, (M, 128), generator=g).float().to(dev)
    W = torch.randint(-3, 4, (N, 128), generator=g).float().to(dev)
    assert torch.equal(lib.linear_bf16x3(dY, W, trans_w=True), dY @ W.t())
    dY = torch.randn(M, 128, generator=g).to(dev)
    W = torch.randn(N, 128, generator=g).to(dev) / 11.0
    out = lib.linear_bf16x3(dY, W, trans_w=True)
    ref = dY.double() @ W.double().t()
    assert (out.double() - ref).abs().max().item() < 2e-6 * max(1.0, ref.abs().max().item())

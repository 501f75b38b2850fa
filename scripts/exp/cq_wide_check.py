"""Context-query block (hual_cq_attn_fwd / _bwd) against the CPU oracle at long-clip shapes: errors relative to each tensor's
largest element.  HUAL_CQ_NO_WIDE=1 runs the staged / global-operand kernels of csrc/cq.hip instead of csrc/cqwide.hip (for comparison)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_blocks as tb      # noqa: E402
from oracle import philox as px   # noqa: E402
from oracle import seqpan_ref as R  # noqa: E402

shapes = tb.CQ_WIDE_SHAPES + [dict(B=32, T=256, L=20, C=4, seed=31, max_vlen=256)]
shapes += tb.SHAPES + [dict(B=64, T=128, L=20, C=4, seed=32, max_vlen=128), dict(B=5, T=100, L=30, C=4, seed=33, max_vlen=100)]
for shape in shapes:
    blk = tb.Block(**shape)
    lib = blk.lib
    x, dy = blk.rand(blk.R, 5), blk.rand(blk.R, 6)
    xd, dyd = x.to(blk.dev), dy.to(blk.dev)
    feats, dx = torch.empty_like(xd), torch.empty_like(xd)
    lib.check(blk.l.hual_cq_attn_fwd(*blk.args(), lib.ptr(xd), lib.ptr(feats), *blk.tail()))
    lib.check(blk.l.hual_cq_attn_bwd(*blk.args(), lib.ptr(dyd), lib.ptr(dx), lib.ptr(blk.grads), *blk.tail()))
    torch.cuda.synchronize()
    pr = {k: t.detach().double().clone().requires_grad_(True) for k, t in blk.p.items()}
    xr = x.double().clone().requires_grad_(True)
    v, q = blk.split(xr)
    q2v = R.cq_attention(v, q, blk.v_mask, blk.q_mask, pr, 'q2v_attn', blk.rng, px.SITE_TRI + 0, blk.rows_v, px.SITE_TRI + 1, blk.rows_q)
    v2q = R.cq_attention(q, v, blk.q_mask, blk.v_mask, pr, 'v2q_attn', blk.rng, px.SITE_TRI + 2, blk.rows_q, px.SITE_TRI + 3, blk.rows_v)
    ref = torch.cat([q2v.reshape(blk.Nv, 128), v2q.reshape(blk.Nq, 128)])
    ref.backward(dy.double())

    def err(a, b):
        a, b = torch.as_tensor(a).double().reshape(-1), torch.as_tensor(b).double().reshape(-1)
        return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)
    out = {'feats.v': err(feats.cpu()[:blk.Nv], ref.detach()[:blk.Nv]), 'feats.q': err(feats.cpu()[blk.Nv:], ref.detach()[blk.Nv:]),
           'dx.v': err(dx.cpu()[:blk.Nv], xr.grad[:blk.Nv]), 'dx.q': err(dx.cpu()[blk.Nv:], xr.grad[blk.Nv:])}
    hg = blk.params_grad()
    for k, t in pr.items():
        if t.grad is not None:
            out[k] = err(hg[k], t.grad)
    print(shape, ' '.join('%s=%.1e' % (k.replace('efficient_trilinear/', ''), v) for k, v in out.items()), flush=True)

import os, sys
import numpy as np, torch
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_blocks as tb
import parity_util as pu
from oracle import philox as px
from oracle import seqpan_ref as R
for shape in [dict(B=2, T=256, L=40, C=4, seed=61, max_vlen=256), dict(B=3, T=200, L=36, C=4, seed=62, max_vlen=224),
              dict(B=2, T=160, L=64, C=4, seed=63, max_vlen=160), dict(B=2, T=128, L=100, C=4, seed=64, max_vlen=128),
              dict(B=2, T=40, L=90, C=4, seed=65, max_vlen=96)]:
    blk = tb.Block(**shape)
    lib = blk.lib
    def err(a, b):
        a, b = torch.as_tensor(a).double().reshape(-1), torch.as_tensor(b).double().reshape(-1)
        return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)
    x, dy = blk.rand(blk.R, 5), blk.rand(blk.R, 6)
    xd, dyd = x.to(blk.dev), dy.to(blk.dev)
    feats, dx = torch.empty_like(xd), torch.empty_like(xd)
    lib.check(blk.l.hual_cq_attn_fwd(*blk.args(), lib.ptr(xd), lib.ptr(feats), *blk.tail()))
    lib.check(blk.l.hual_cq_attn_bwd(*blk.args(), lib.ptr(dyd), lib.ptr(dx), lib.ptr(blk.grads), *blk.tail()))
    pr = {k: t.detach().double().clone().requires_grad_(True) for k, t in blk.p.items()}
    xr = x.double().clone().requires_grad_(True)
    v, q = blk.split(xr)
    q2v = R.cq_attention(v, q, blk.v_mask, blk.q_mask, pr, 'q2v_attn', blk.rng, px.SITE_TRI + 0, blk.rows_v, px.SITE_TRI + 1, blk.rows_q)
    v2q = R.cq_attention(q, v, blk.q_mask, blk.v_mask, pr, 'v2q_attn', blk.rng, px.SITE_TRI + 2, blk.rows_q, px.SITE_TRI + 3, blk.rows_v)
    ref = torch.cat([q2v.reshape(blk.Nv, 128), v2q.reshape(blk.Nq, 128)])
    ref.backward(dy.double())
    hg = blk.params_grad()
    print(shape, 'feats %.1e dx %.1e dparam %.1e' % (err(feats.cpu(), ref.detach()), err(dx.cpu(), xr.grad), max(err(hg[k], t.grad) for k, t in pr.items() if t.grad is not None)), flush=True)
# whole model at a shape that needs the global form
case = pu.make_case(B=2, T=256, L=40, C=5, seed=71, max_vlen=256)
rows, idx_equal, o, h, m = pu.compare(*case, drop_rate=0.2)
pu.assert_rows(rows)
print('whole model T256 L40: %d rows ok, spans equal %s' % (len(rows), idx_equal))

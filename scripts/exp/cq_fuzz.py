"""Random shapes through the context-query block (hual_cq_attn_fwd / _bwd) and the dual-attention block (layer 0) against the CPU oracle:
B in 1..5, T in 4..256, L in 3..40 (3..32 beyond 128 frames; both context-query paths, both attention-backward paths); prints the worst relative error per shape
and fails above 2e-5 of a tensor's scale."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_blocks as tb
from oracle import philox as px
from oracle import seqpan_ref as R

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
worst = 0.0
for it in range(n):
    T = int(rng.integers(4, 257)); L = int(rng.integers(3, 41 if T <= 128 else 33)); B = int(rng.integers(1, 6))      # (L > 32 needs T <= 128: INTEGRATION.md)
    shape = dict(B=B, T=T, L=L, C=4, seed=int(rng.integers(1, 10000)), max_vlen=max(T, L, 8))
    blk = tb.Block(**shape)
    lib = blk.lib
    def err(a, b):
        a, b = torch.as_tensor(a).double().reshape(-1), torch.as_tensor(b).double().reshape(-1)
        return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)
    out = {}
    # context-query block
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
    out['cq.feats'] = err(feats.cpu(), ref.detach()); out['cq.dx'] = err(dx.cpu(), xr.grad)
    hg = blk.params_grad()
    out['cq.dparam'] = max(err(hg[k], t.grad) for k, t in pr.items() if t.grad is not None)
    # dual attention, layer 0
    x, dy = blk.rand(blk.R, 3), blk.rand(blk.R, 4)
    xd, dyd = x.to(blk.dev), dy.to(blk.dev)
    y, dx = torch.empty_like(xd), torch.empty_like(xd)
    lib.check(blk.l.hual_dual_attn_fwd(*blk.args(), 0, lib.ptr(xd), lib.ptr(y), *blk.tail()))
    lib.check(blk.l.hual_dual_attn_bwd(*blk.args(), 0, lib.ptr(dyd), lib.ptr(dx), lib.ptr(blk.grads), *blk.tail()))
    pr = {k: t.detach().double().clone().requires_grad_(True) for k, t in blk.p.items()}
    xr = x.double().clone().requires_grad_(True)
    v, q = blk.split(xr)
    H = blk.cfg.num_heads
    v_ = R.dual_attn_block(v, q, pr, 'd_attn_0', H, blk.v_mask, blk.q_mask, blk.rng, px.SITE_DA, blk.rows_v)
    q_ = R.dual_attn_block(q, v, pr, 'd_attn_0', H, blk.q_mask, blk.v_mask, blk.rng, px.SITE_DA, blk.rows_q)
    ref = torch.cat([v_.reshape(blk.Nv, 128), q_.reshape(blk.Nq, 128)])
    ref.backward(dy.double())
    out['da.y'] = err(y.cpu(), ref.detach()); out['da.dx'] = err(dx.cpu(), xr.grad)
    w = max(out.values())
    worst = max(worst, w)
    print('B%d T%d L%d  ' % (B, T, L) + ' '.join('%s=%.1e' % kv for kv in out.items()) + ('   <<<<' if w > 2e-5 else ''), flush=True)
print('worst', worst)
sys.exit(1 if worst > 2e-5 else 0)

"""the one failure of the round-6 long-query fuzz (B48 T16 L48 C15 vdim512, seed 350101: char_embs/filter_1 gradient 4e-3 of its max off): where does
the difference sit - a few isolated (word, channel) units whose relu / max-over-characters selection flipped on rounding, or everywhere?
Also evaluates the oracle in float64 (its own selection at higher precision)."""
import sys, os
import numpy as np, torch
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, 'tests'))
import parity_util as pu
shape = dict(B=48, T=16, L=48, C=15, seed=350101, max_vlen=48, vdim=512)
case = pu.make_case(**shape)
for dt in (torch.float32, torch.float64):
    rows, idx_equal, o, h, m = pu.compare(*case, drop_rate=0.2, oracle_dtype=dt)
    hg = m.grads_dict()
    cfg, p, wv, b, labels = case
    o_out, o_grads = pu.oracle_run(cfg, p, wv, b, labels, 0.2, 5, 7, dtype=dt, with_grads=True, relu_pin=pu.relu_pins(m, 48, 16, 48))
    print('oracle dtype', dt)
    for k in ('char_embs/filter_1', 'char_embs/char_table', 'char_embs/filter_0', 'char_embs/filter_2', 'char_embs/filter_3', 'char_embs/bias_1'):
        a = torch.from_numpy(hg[k]).double().reshape(-1); r = o_grads[k].detach().double().reshape(-1)
        d = (a - r).abs(); sc = float(r.abs().max())
        top = torch.topk(d, 5)
        print('  %-24s max|d|/max|ref| %.2e   l2 rel %.2e   elements > 1e-4 max: %d of %d   top5 d/max: %s' % (
            k, float(d.max()) / sc, float(d.norm() / r.norm()), int((d > 1e-4 * sc).sum()), d.numel(), ['%.1e' % (float(x) / sc) for x in top.values]))

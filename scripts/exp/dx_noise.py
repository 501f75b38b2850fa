"""Error of the ACTIVATION gradients along the backward chain (HIP d.* workspace buffers against float64 autograd of the oracle, ReLU
pins shared, dropout 0.2): which stage of the backward pass injects the gradient noise that scripts/exp/grad_noise.py sees in the
parameter gradients.      python scripts/exp/dx_noise.py [c1|c2]"""
import os, sys
import numpy as np, torch
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, 'tests'))
import parity_util as pu
from oracle import seqpan_ref as R
import collections

SHAPES = dict(c1=dict(B=16, T=64, L=20, C=8, seed=12345, max_vlen=64, vdim=1024, num_words=1000),
              c2=dict(B=64, T=128, L=20, C=8, seed=12345, max_vlen=128, vdim=1024))
which = sys.argv[1] if len(sys.argv) > 1 else 'c1'
cfg, p, wv, b, labels = pu.make_case(**SHAPES[which])
B, T = b['video'].shape[:2]; L = b['word_ids'].shape[1]
m = pu.hip_model(cfg, p, wv)
m.set_rng(1, 1)
h = m.forward(b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy(), drop_rate=0.2,
              labels=tuple(x.numpy() for x in labels))
torch.cuda.synchronize()
pins = pu.relu_pins(m, B, T, L)
m.backward(); torch.cuda.synchronize()

def oracle(dtype):
    pr = collections.OrderedDict((k, t.detach().clone().to(dtype).requires_grad_(True)) for k, t in p.items())
    out = R.forward(pr, cfg, wv.to(dtype), b['video'].to(dtype), b['lens'], b['word_ids'], b['char_ids'], drop_rate=0.2, seed=1, offset=1,
                    labels=labels, want_tap=True, relu_pin=pins)
    tap = out['tap']
    want = {}
    def uni(n):
        return [tap[n + '.v'], tap[n + '.q']]
    names = [('d.fuse', [tap['fuse']]), ('d.align.that', [tap['t_hat']]), ('d.align.vhat', [tap['v_hat']]), ('d.cq.feats', [tap['q2v'], tap['v2q']]),
             ('d.cq.c2q', [tap['q2v_attn.c2q'], tap['v2q_attn.c2q']]), ('d.cq.q2c', [tap['q2v_attn.q2c'], tap['v2q_attn.q2c']]),
             ('d.da1.res', uni('da1.res')), ('d.da1.g', uni('da1.g')), ('d.da1.s', uni('da1.s')), ('d.da1.x', uni('da1.x')),
             ('d.da1.s_att', uni('da1.s_att')), ('d.da1.x_att', uni('da1.x_att')),
             ('d.da1.in', uni('da0.out')),
             ('d.da0.res', uni('da0.res')), ('d.da0.g', uni('da0.g')), ('d.da0.s_att', uni('da0.s_att')), ('d.da0.in', uni('cb.x4')),
             ('d.cb.x0', uni('cb.x0')), ('d.outputs.heads+fe', [tap['outputs']])]
    flat = [t for _, ts in names for t in ts]
    gs = torch.autograd.grad(out['loss'], flat, allow_unused=True)
    k = 0
    for n, ts in names:
        parts = []
        for t in ts:
            g = gs[k]; k += 1
            parts.append((g if g is not None else torch.zeros_like(t)).reshape(-1, t.shape[-1]))
        want[n] = torch.cat(parts, 0).double()
    return want

def fwd_err():
    outs = {}
    for dt in (torch.float64, torch.float32):
        pr = collections.OrderedDict((k, t.detach().clone().to(dt)) for k, t in p.items())
        o = R.forward(pr, cfg, wv.to(dt), b['video'].to(dt), b['lens'], b['word_ids'], b['char_ids'], drop_rate=0.2, seed=1, offset=1,
                      labels=labels, want_tap=True, relu_pin=pins)
        outs[dt] = o
    o64, o32 = outs[torch.float64], outs[torch.float32]
    print('# forward tensors, max|x - f64| (absolute) - hip, f32 oracle, max|f64|')
    for hname, ref in pu.tap_pairs(o64['tap'], B, T, L, cfg.attn_layer):
        if hname in ('cat',) or hname.startswith('cb.c') or hname.startswith('cb.y') or '.c' in hname:
            continue
        got = (m.tap('lin')[:B * T] if hname == 'lin[v]' else m.tap(hname)).double().cpu()
        r32 = dict(pu.tap_pairs(o32['tap'], B, T, L, cfg.attn_layer))[hname].double()
        ref = ref.double()
        print('%-14s %10.2e %10.2e %10.2e' % (hname, float((got - ref).abs().max()), float((r32 - ref).abs().max()), float(ref.abs().max())))
    for k in ('start_logits', 'end_logits', 'match_scores'):
        ref = o64[k].double()
        print('%-14s %10.2e %10.2e %10.2e' % (k, float((h[k].double().cpu() - ref).abs().max()), float((o32[k].double() - ref).abs().max()), float(ref.abs().max())))


if '--fwd' in sys.argv:
    fwd_err()
w64, w32 = oracle(torch.float64), oracle(torch.float32)
print('# %s: activation gradients, max|x - f64| / max|f64|' % which)
print('# %-22s %10s %10s %10s' % ('buffer', 'hip', 'f32 oracle', 'max|ref|'))
for n in w64:
    ref = w64[n]
    if n == 'd.outputs.heads+fe':
        try:
            got = (m.tap('d.outputs.heads') + m.tap('d.fe0.x0')).double().cpu()
        except Exception as e:
            print(n, 'n/a', e); continue
    else:
        got = m.tap(n).double().cpu()
    if got.shape != ref.shape:
        print(n, 'shape mismatch', tuple(got.shape), tuple(ref.shape)); continue
    sc = float(ref.abs().max())
    extra = ''
    if got.shape[0] == B * (T + L):      # video rows / query rows apart
        Nv = B * T
        for nm, sl in (('v', slice(0, Nv)), ('q', slice(Nv, None))):
            s2 = float(ref[sl].abs().max())
            extra += '   %s: %.2e (f32 %.2e) of %.2e' % (nm, float((got[sl] - ref[sl]).abs().max()) / s2, float((w32[n][sl] - ref[sl]).abs().max()) / s2, s2)
    print('%-24s %10.2e %10.2e %10.2e%s' % (n, float((got - ref).abs().max()) / sc, float((w32[n] - ref).abs().max()) / sc, sc, extra))

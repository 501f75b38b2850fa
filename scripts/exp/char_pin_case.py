"""char max-pool pins against the oracle for one small case: every unit whose pinned window is not the oracle's own maximum, with the values"""
import sys, os
import numpy as np, torch
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, 'tests'))
import parity_util as pu
case = pu.make_case(B=1, T=5, L=3, C=4, seed=2, max_vlen=8)
cfg, p, wv, b, labels = case
m = pu.hip_model(cfg, p, wv); m.set_rng(5, 7); m.debug_taps = True
m.forward(b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy(), drop_rate=0.0, labels=tuple(x.numpy() for x in labels))
torch.cuda.synchronize()
pins = pu.relu_pins(m, 1, 5, 3)
o_out, _ = pu.oracle_run(cfg, p, wv, b, labels, 0.0, 5, 7, with_grads=False, relu_pin=pins)
print('char ids', b['char_ids'].tolist())
c0 = 0
for i in range(4):
    z = o_out['tap']['char.z%d' % i].detach().double(); ch = z.shape[1]
    a = pins['char.arg'][:, :, c0:c0 + ch].permute(0, 2, 1).long(); c0 += ch
    own = torch.relu(z).max(dim=3).values
    got = torch.gather(z, 3, a.clamp_min(0).unsqueeze(-1)).squeeze(-1) * (a >= 0).to(z.dtype)
    d = (own - got).abs()
    for idx in (d > 0).nonzero().tolist():
        bb, cc, ll = idx
        print('filter %d channel %2d word %d: windows %s  oracle argmax %d  pinned %d  d %.3e (max|z| %.2f)' % (
            i, cc, ll, ['%.9f' % v for v in z[bb, cc, ll].tolist()], int(z[bb, cc, ll].argmax()), int(a[bb, cc, ll]), float(d[bb, cc, ll]), float(z.abs().max())))

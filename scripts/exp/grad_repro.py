"""run-to-run reproducibility of the gradient bucket on an otherwise idle GPU: which gradient tensors change their bits between two
evaluations of the same batch, and by how much (relative to the tensor's largest element)?   grad_repro.py [runs]"""
import sys, os
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import parity_util as pu
cfg, p, wv, b, labels = pu.make_case(B=16, T=64, L=20, C=8, seed=12345, max_vlen=64, vdim=256)
m = pu.hip_model(cfg, p, wv); m.ws_poison = None
dv = [torch.as_tensor(x).cuda() for x in (b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy())]
lab = [torch.as_tensor(x.numpy()).cuda() for x in labels]
def one():
    m.forward(*dv, drop_rate=0.0, labels=lab); m.backward(); torch.cuda.synchronize()
    return m.grads_dict()
ref = one()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
worst = {}
for _ in range(n):
    g = one()
    for k, v in g.items():
        if not np.array_equal(v, ref[k]):
            sc = max(float(np.abs(ref[k]).max()), 1e-30)
            w = worst.setdefault(k, [0, 0.0, 0])
            w[0] += 1; w[1] = max(w[1], float(np.abs(v - ref[k]).max()) / sc); w[2] = max(w[2], int((v != ref[k]).sum()))
print('%d of %d gradient tensors changed bits in some of %d repeats' % (len(worst), len(ref), n))
for k, (c, r, e) in sorted(worst.items(), key=lambda kv: -kv[1][1])[:60]:
    print('  %-60s runs %3d  max rel %.2e  elements %7d of %d' % (k, c, r, e, ref[k].size))

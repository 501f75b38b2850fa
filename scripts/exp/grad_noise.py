"""per-tensor gradient error of the HIP path against the float64 oracle at the c1 shape (dropout 0.2): which kernels carry the noise"""
import sys, os
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import parity_util as pu
cfg, p, wv, b, labels = pu.make_case(B=16, T=64, L=20, C=8, seed=12345, max_vlen=64, vdim=1024, num_words=1000)
rows, idx_equal, o, h, m = pu.compare(cfg, p, wv, b, labels, drop_rate=0.2, seed=1, offset=1, oracle_dtype=torch.float64)
g = [(d / max(r, 1e-30), n, d, r) for k, n, d, r in rows if k == 'grad']
g.sort(reverse=True)
print('gradient tensors by max|hip - f64| / max|f64|   (%d tensors)' % len(g))
for rel, n, d, r in g[:28]:
    print('%9.2e  %-70s ref %.2e' % (rel, n, r))
print('median %.2e' % np.median([x[0] for x in g]))

"""Per-tensor gradient error of the HIP path against the float64 oracle (dropout 0.2, ReLU pins shared): which kernels carry
the noise, and how far every one of the 170 gradient tensors sits from the relative parity gate (tests/parity_util.row_ok).

    python scripts/exp/grad_noise.py [c1|c2|c4] [out.txt]

c1 = B16 T64 vdim1024 (BASELINE configs[0] with the YAML's vdim), c2 = B64 T128 vdim1024 (configs[1], the bench shape),
c4 = B32 T256 (configs[3] per GPU).  Also runs the float32 oracle against the float64 one on the same pins for scale."""
import os
import sys

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'tests'))
import parity_util as pu  # noqa: E402

SHAPES = dict(c1=dict(B=16, T=64, L=20, C=8, seed=12345, max_vlen=64, vdim=1024, num_words=1000),
              c2=dict(B=64, T=128, L=20, C=8, seed=12345, max_vlen=128, vdim=1024),
              c4=dict(B=32, T=256, L=20, C=8, seed=4321, max_vlen=256, vdim=1024))


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else 'c1'
    out = open(sys.argv[2], 'w') if len(sys.argv) > 2 else sys.stdout
    cfg, p, wv, b, labels = pu.make_case(**SHAPES[which])
    rows, idx_equal, o, h, m = pu.compare(cfg, p, wv, b, labels, drop_rate=0.2, seed=1, offset=1, oracle_dtype=torch.float64)
    B, T = b['video'].shape[:2]
    L = b['word_ids'].shape[1]
    # the float32 oracle on the same pins: what a plain fp32 implementation delivers
    pins = pu.relu_pins(m, B, T, L)
    o32, g32 = pu.oracle_run(cfg, p, wv, b, labels, 0.2, 1, 1, dtype=torch.float32, relu_pin=pins)
    o64, g64 = pu.oracle_run(cfg, p, wv, b, labels, 0.2, 1, 1, dtype=torch.float64, relu_pin=pins)
    gmax = pu.grad_scale(rows)
    g = []
    for k, n, d, r in rows:
        if k != 'grad':
            continue
        d32 = float((g32[n].double() - g64[n]).abs().max())
        g.append((d / max(r, 1e-30), n, d, r, d32 / max(r, 1e-30), d / (1e-3 * max(r, 1e-3 * gmax))))
    g.sort(reverse=True)
    w = lambda s: out.write(s + '\n')
    w('# %s %s  dropout 0.2: gradient tensors, max|hip - f64| / max|f64| (all %d), fp32 PyTorch oracle beside it' % (which, SHAPES[which], len(g)))
    w('# largest gradient of the run %.3e; "gate" = error / (1e-3 max(max|ref|, 1e-3 gmax)): > 1 fails tests/parity_util.row_ok' % gmax)
    w('# %9s %9s %7s  %-72s %s' % ('hip/ref', 'f32/ref', 'gate', 'tensor', 'max|ref|'))
    for rel, n, d, r, rel32, gate in g:
        w('%11.2e %9.2e %7.3f  %-72s %.2e' % (rel, rel32, gate, n, r))
    nz = [x for x in g if x[3] > 1e-3 * gmax * 1e-3]
    w('# median hip %.2e  f32 oracle %.2e   worst gate %.3f   tensors above 5e-6 of their max: %d of %d (ignoring exact-zero gradients: %d of %d)'
      % (np.median([x[0] for x in g]), np.median([x[4] for x in g]), max(x[5] for x in g),
         sum(x[0] > 5e-6 for x in g), len(g), sum(x[0] > 5e-6 for x in nz), len(nz)))
    w('# loss hip %.7f  f64 %.7f  spans equal %s' % (float(h['loss']), float(o['loss']), idx_equal))
    bad = pu.failures(rows)
    w('# parity gate failures: %d' % len(bad))
    if bad:
        w(pu.format_report(bad, gmax))


if __name__ == '__main__':
    main()

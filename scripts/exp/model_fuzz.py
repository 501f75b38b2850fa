"""Whole-model parity at random shapes: forward taps, outputs, loss terms and all gradient tensors of the HIP train step against the
oracle (tests/parity_util.compare, dropout 0.2), B x (T + L) kept small enough for the CPU oracle.  Exercises the row-tile variants of
the fused kernels (tails included) at tile sizes the named test shapes do not hit.
usage: python scripts/exp/model_fuzz.py [cases] [seed] [seconds] [long]
'long': queries of 3..80 words, words of up to 22 characters, clips of at most 128 frames - the lengths of the reference's own ActivityNet
annotations (tests/golden/lengths_anet.npz), which take the context-query kernels' staged / global-operand forms"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import parity_util as pu

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 420.0
long_q = len(sys.argv) > 4 and sys.argv[4] == 'long'
g = np.random.default_rng(seed)
t0 = time.time()
bad = 0
for i in range(n):
    if time.time() - t0 > budget:
        print('time budget reached after %d cases' % i); break
    T = int(g.choice([3, 5, 7, 16, 17, 31, 33, 48, 64, 65, 100, 127, 128, 129, 170, 200, 256]))
    L = int(g.integers(3, 33))                      # (oracle.synthetic_batch needs T, L >= 3; clips of 1-2 frames: tests/test_gpu_shapes.py)
    C = int(g.integers(4, 13))
    if long_q:
        T = int(g.choice([16, 33, 64, 100, 128]))
        L = int(g.integers(24, 81))
        C = int(g.integers(4, 23))                      # (the char CNN's widest filter spans 4 characters)
    rows_max = 6000
    B = int(max(1, min(48, g.integers(1, max(2, rows_max // (T + L) + 1)))))
    vdim = int(g.choice([256, 512, 1024]))
    shape = dict(B=B, T=T, L=L, C=C, seed=int(g.integers(1, 10**6)), max_vlen=max(T, L, 8), vdim=vdim)     # (position table: clips AND queries)
    try:
        rows, idx_equal, o, h, m = pu.compare(*pu.make_case(**shape), drop_rate=0.2)
        f = pu.failures(rows, ('tap', 'out', 'loss', 'grad'), 1e-3)
        ok = not f and idx_equal
        print('%2d %s  R=%d  %s  (%.0f s)' % (i, shape, B * (T + L), 'ok' if ok else 'FAIL idx_equal=%s\n%s' % (idx_equal, pu.format_report(f, pu.grad_scale(rows))), time.time() - t0), flush=True)
        bad += 0 if ok else 1
    except Exception as e:
        print('%2d %s  EXCEPTION %r' % (i, shape, e), flush=True)
        bad += 1
print('failures: %d' % bad)
sys.exit(1 if bad else 0)

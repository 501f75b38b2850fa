"""one whole-model parity case by its shape (as scripts/exp/model_fuzz.py prints it):  fuzz_one.py B T L C seed max_vlen vdim"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import parity_util as pu
B, T, L, C, seed, mv, vd = (int(x) for x in sys.argv[1:8])
rows, idx_equal, o, h, m = pu.compare(*pu.make_case(B=B, T=T, L=L, C=C, seed=seed, max_vlen=mv, vdim=vd), drop_rate=0.2)
f = pu.failures(rows)
print('pins differing from the oracle:', [r for r in rows if r[0] == 'pin'])
print('ok' if not f and idx_equal else 'FAIL idx_equal=%s\n%s' % (idx_equal, pu.format_report(f, pu.grad_scale(rows))))

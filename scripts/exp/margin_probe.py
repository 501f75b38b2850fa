"""How close the parity tests sit to the 1e-3 bar: for a set of cases print the worst ratio min(diff, diff/ref) / 1e-3."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import parity_util as pu
cases = [dict(B=2, T=16, L=5, C=4, max_vlen=16), dict(), dict(B=4, T=24, L=7, C=5, seed=21), dict(B=3, T=37, L=9, C=4, seed=11, max_vlen=40),
         dict(B=6, T=48, L=12, C=6, seed=5, max_vlen=64), dict(B=8, T=64, L=20, C=8, seed=9, max_vlen=64)]
for kw in cases:
    for drop in (0.0, 0.2):
        case = pu.make_case(**kw)
        rows, idx_equal, o, h, m = pu.compare(*case, drop_rate=drop)
        def score(r):
            return min(r[2], r[2] / max(r[3], 1e-30)) / 1e-3
        w = sorted(rows, key=lambda r: -score(r))[:3]
        print(kw, 'drop', drop, 'idx', idx_equal, ' | '.join('%s %s %.2f' % (r[0], r[1].split('/')[-2] + '/' + r[1].split('/')[-1] if '/' in r[1] else r[1], score(r)) for r in w))

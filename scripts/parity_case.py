import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import parity_util as pu
B, T, L, C, seed, drop = [float(x) if '.' in x else int(x) for x in sys.argv[1:7]]
case = pu.make_case(B=B, T=T, L=L, C=C, seed=seed, max_vlen=max(32, T))
rows, idx_equal, o, h, m = pu.compare(*case, drop_rate=drop, seed=99, offset=5)
bad = [r for r in rows if not (r[2] <= 2e-4 or r[2] <= 2e-4 * r[3])]
print("\n".join("%s %.1e" % (r[1], r[2]/max(r[3],1e-30)) for r in bad))
print('n_bad', len(bad), 'of', len(rows), 'idx_equal', idx_equal)

import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import torch
import parity_util as pu
for env in ({}, {'HUAL_GEMM_BF16': '0'}, {'HUAL_DW_IMPL': '0'}, {'HUAL_GEMM_BF16': '0', 'HUAL_DW_IMPL': '0'}, {'HUAL_FEATURE_KSPLIT': '0'}):
    for k in ('HUAL_GEMM_BF16', 'HUAL_DW_IMPL', 'HUAL_FEATURE_KSPLIT'):
        os.environ.pop(k, None)
    os.environ.update(env)
    case = pu.make_case(B=2, T=16, L=5, C=4, max_vlen=16)
    rows, idx_equal, o, h, m = pu.compare(*case, drop_rate=0.2)
    worst = [r for r in rows if not (r[2] <= 1e-3 or r[2] <= 1e-3 * r[3])] + sorted([r for r in rows if 'trilinear' in r[1]], key=lambda r: -r[2])[:3]
    print(env, idx_equal)
    for r in worst:
        print('   %-5s %-60s diff %.3e ref %.3e rel %.2e' % (r[0], r[1], r[2], r[3], r[2] / max(r[3], 1e-30)))

import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import parity_util as pu
case = pu.make_case(B=64, T=128, L=20, C=8, seed=12345, max_vlen=128)
names = ('da0.qkv', 'da0.ktvt', 'da0.s', 'da0.out', 'da1.qkv', 'da1.ktvt', 'da1.out', 'fe0.qkv', 'fe1.out', 'head.hs')
def run(fuse):
    os.environ['HUAL_FUSE_DA'] = fuse
    cfg, p, wv, b, labels = case
    m = pu.hip_model(cfg, p, wv); m.set_rng(5, 7)
    m.forward(b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy(), drop_rate=0.0, labels=tuple(x.numpy() for x in labels))
    torch.cuda.synchronize()
    return {n: m.tap(n).clone() for n in names}
ref = run('0')
ref2 = run('0')
print('unfused vs unfused:', {n: float((ref[n] - ref2[n]).abs().max()) for n in names})
for rep in range(6):
    a = run('1')
    out = []
    for n in names:
        d = (a[n] - ref[n]).abs()
        bad = d > 0
        rows = bad.any(dim=1).nonzero().flatten().tolist()
        if rows:
            cols = bad.any(dim=0).nonzero().flatten().tolist()
            out.append('%s: %d rows %s cols[%d] %s' % (n, len(rows), rows[:6], len(cols), cols[:8]))
    print('rep', rep, out if out else 'all equal')

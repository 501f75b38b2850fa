"""same process, second stream running our forward + backward: does the gradient bucket (before clipping) of the same batch change its
bits under that load, and by how much?   race_gradient.py N [B T L C vdim max_vlen]"""
import sys, os, threading, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import parity_util as pu
sh = [int(x) for x in sys.argv[2:8]] if len(sys.argv) >= 8 else [16, 64, 20, 8, 256, 64]
case = pu.make_case(B=sh[0], T=sh[1], L=sh[2], C=sh[3], seed=12345, max_vlen=sh[5], vdim=sh[4])
cfg, p, wv, b, labels = case
m = pu.hip_model(cfg, p, wv); m.ws_poison = None
m2 = pu.hip_model(cfg, p, wv); m2.ws_poison = None
dv = [torch.as_tensor(x).cuda() for x in (b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy())]
lab = [torch.as_tensor(x.numpy()).cuda() for x in labels]
stop = False
def load():
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        while not stop:
            for _ in range(10):
                m2.forward(*dv, drop_rate=0.0, labels=lab); m2.backward()
            s.synchronize()
def snap():
    o = m.forward(*dv, drop_rate=0.0, labels=lab)
    m.backward()
    torch.cuda.synchronize()
    return m.grads.detach().cpu().numpy().copy(), float(o['loss'])
for _ in range(3): snap()
ref, l0 = snap()
q = [int((snap()[0] != ref).sum()) for _ in range(50)]
th = threading.Thread(target=load); th.start(); time.sleep(0.5)
n = int(sys.argv[1]); nel = []; rel = []
scale = np.abs(ref).max()
for it in range(n):
    g, l = snap()
    d = int((g != ref).sum())
    if d:
        nel.append(d); rel.append(float(np.abs(g - ref).max() / scale))
stop = True; th.join()
big = [(a, '%.1e' % r) for a, r in zip(nel, rel) if r > 1e-5]
print('shape B T L C vdim max_vlen = %s; ' % sh + 'quiet: differing elements per run max %d (runs that differ: %d of 50); under a second stream running forward+backward: %d of %d differ; '
      'of those with max |dg| > 1e-5 of the largest gradient: %d %s (bucket of %d)' % (max(q), sum(x > 0 for x in q), len(nel), n, len(big), big[:12], ref.size))

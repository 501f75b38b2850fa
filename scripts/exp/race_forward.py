"""same process, second stream: which tensor differs first, and does a torch-only second stream disturb the forward too?"""
import sys, os, threading, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import parity_util as pu
case = pu.make_case(B=16, T=64, L=20, C=8, seed=12345, max_vlen=64, vdim=256)
cfg, p, wv, b, labels = case
m = pu.hip_model(cfg, p, wv); m.ws_poison = None; m.debug_taps = True
m2 = pu.hip_model(cfg, p, wv); m2.ws_poison = None
dv = [torch.as_tensor(x).cuda() for x in (b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy())]
kind = sys.argv[1]
stop = False
def load():
    s = torch.cuda.Stream()
    x = torch.randn(4096, 4096, device='cuda')
    with torch.cuda.stream(s):
        while not stop:
            for _ in range(20):
                if kind == 'ours': m2.forward(*dv, drop_rate=0.0)
                else: (x @ x).softmax(1)
            s.synchronize()
def snap():
    o = m.forward(*dv, drop_rate=0.0)
    torch.cuda.synchronize()
    t = {}
    for name, (off, rows, cols) in m._ws_table.items():
        if rows * cols > 0 and not name.startswith(('params.', 'dw.table', 'cq.sr', 'cq.sc')):
            t[name] = (off, m._ws[off:off + rows * cols * 4].cpu().numpy().copy())
    return t
ref = snap()
th = threading.Thread(target=load); th.start(); time.sleep(0.5)
from collections import Counter
first = Counter(); nbad = 0
n = int(sys.argv[2])
for it in range(n):
    cur = snap()
    diff = sorted((off, name) for name, (off, a) in ref.items() if not np.array_equal(a, cur[name][1]))
    if diff:
        nbad += 1; first[diff[0][1]] += 1
stop = True; th.join()
print('second stream runs %s: %d of %d forwards differ; first differing tensor (workspace order): %s' % (kind, nbad, n, first.most_common(8)))

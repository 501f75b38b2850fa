"""same process, second stream running our whole TRAIN STEP (graph replays): one step from the same state, parameters + Adam slots against a quiet
run.  The gradient bucket carries ~1e-7 of float-atomics noise, so an element whose gradient is noise may move by +-3.2 lr either way (no bias
correction): counted are elements off by more than that, and the share of elements off by > 0.1 lr.   race_step.py N"""
import sys, os, threading, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import parity_util as pu
from hual_amd.train import Trainer
lr = 1e-4
cfg, p, wv, b, labels = pu.make_case(B=16, T=64, L=20, C=8, seed=12345, max_vlen=64, vdim=256)
def mk():
    m = pu.hip_model(cfg, p, wv); m.ws_poison = None
    tr = Trainer(m, world=1, use_graph=True)
    tr.set_batch(b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy(), *[x.numpy() for x in labels])
    return m, tr
m, tr = mk(); m2, tr2 = mk()
for _ in range(3): tr.step(lr=lr, drop_rate=0.2); tr2.step(lr=lr, drop_rate=0.2)
torch.cuda.synchronize()
state = [t.clone() for t in (m.params, m.adam_m, m.adam_v, m.rng_state)]
def one():
    for t, s in zip((m.params, m.adam_m, m.adam_v, m.rng_state), state): t.copy_(s)
    tr.step(lr=lr, drop_rate=0.2)
    torch.cuda.synchronize()
    return m.params.detach().cpu().numpy().copy(), m.adam_v.detach().cpu().numpy().copy(), float(tr.last_loss())
ref = one()
def stats(n):
    big = 0; share = 0.0; vmax = 0.0; dl = 0.0
    for _ in range(n):
        pp, vv, l = one()
        d = np.abs(pp - ref[0])
        big += int((d > 2.2 * 3.2 * lr).sum()); share = max(share, float(np.mean(d > 0.1 * lr)))
        vmax = max(vmax, float(np.abs(vv - ref[1]).max() / max(ref[1].max(), 1e-30))); dl = max(dl, abs(l - ref[2]) / max(abs(ref[2]), 1e-9))
    return 'elements off by > 2 x 3.2 lr: %d; largest share off by > 0.1 lr: %.2e; adam_v max rel diff %.1e; loss max rel diff %.1e' % (big, share, vmax, dl)
print('quiet (%d steps):  %s' % (30, stats(30)))
stop = False
def load():
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        while not stop:
            for _ in range(10): tr2.step(lr=lr, drop_rate=0.2)
            s.synchronize()
th = threading.Thread(target=load); th.start(); time.sleep(0.5)
n = int(sys.argv[1])
print('loaded (%d steps): %s' % (n, stats(n)))
stop = True; th.join()

"""shared-GPU soak with the strongest trigger of the round-6 finding: a second stream runs back-to-back matrix instructions (build_exp/co_mfma.so,
tests/aux/co_mfma.hip) while the forward / the gradient / the whole train step / the active-learning scoring run.   race_mfma_soak.py N [B T L C vdim max_vlen]"""
import sys, os, threading, time, ctypes
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import parity_util as pu
from hual_amd.train import Trainer
co = ctypes.CDLL(os.path.join(R, 'build_exp', 'co_mfma.so'))
co.co_mfma_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
sink = torch.zeros(16, device='cuda')
n = int(sys.argv[1])
sh = [int(x) for x in sys.argv[2:8]] if len(sys.argv) >= 8 else [16, 64, 20, 8, 256, 64]      # B T L C vdim max_vlen
print('shape B T L C vdim max_vlen =', sh, flush=True)
cfg, p, wv, b, labels = pu.make_case(B=sh[0], T=sh[1], L=sh[2], C=sh[3], seed=12345, max_vlen=sh[5], vdim=sh[4])
m = pu.hip_model(cfg, p, wv); m.ws_poison = None
dv = [torch.as_tensor(x).cuda() for x in (b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy())]
lab = [torch.as_tensor(x.numpy()).cuda() for x in labels]
class Load:
    def __enter__(self):
        self.stop = False
        def run():
            s = torch.cuda.Stream()
            while not self.stop:
                for _ in range(8): co.co_mfma_launch(ctypes.c_void_p(s.cuda_stream), ctypes.c_void_p(sink.data_ptr()), 1024, 3000)
                s.synchronize()
        self.th = threading.Thread(target=run); self.th.start(); time.sleep(0.3); return self
    def __exit__(self, *a):
        self.stop = True; self.th.join()
def fwd():
    o = m.forward(*dv, drop_rate=0.0); torch.cuda.synchronize()
    return [o[k].cpu().numpy().copy() for k in ('start_logits', 'end_logits', 'match_scores', 'start_index', 'end_index')]
ref = fwd()
with Load(): bad = sum(any(not np.array_equal(a, c) for a, c in zip(ref, fwd())) for _ in range(n))
print('forward: %d of %d differ under the MFMA loop' % (bad, n), flush=True)
def grad():
    m.forward(*dv, drop_rate=0.0, labels=lab); m.backward(); torch.cuda.synchronize()
    return m.grads.detach().cpu().numpy().copy()
g0 = grad(); sc = float(np.abs(g0).max())
with Load(): worst = max(float(np.abs(grad() - g0).max()) / sc for _ in range(n // 4))
print('gradient bucket: worst |dg| / max|g| over %d evaluations under the MFMA loop: %.1e (quiet noise ~3e-7)' % (n // 4, worst), flush=True)
tr = Trainer(m, world=1, use_graph=True)
tr.set_batch(b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy(), *[x.numpy() for x in labels])
for _ in range(3): tr.step(lr=1e-4, drop_rate=0.2)
torch.cuda.synchronize()
state = [t.clone() for t in (m.params, m.adam_m, m.adam_v, m.rng_state)]
def step():
    for t, s in zip((m.params, m.adam_m, m.adam_v, m.rng_state), state): t.copy_(s)
    tr.step(lr=1e-4, drop_rate=0.2); torch.cuda.synchronize()
    return m.params.detach().cpu().numpy().copy(), float(tr.last_loss())
p0, l0 = step()
with Load():
    res = [step() for _ in range(n // 4)]
print('train step (graph, dropout 0.2): largest share of parameters off by > 0.1 lr %.1e, loss equal in %d of %d' % (
    max(float(np.mean(np.abs(pp - p0) > 1e-5)) for pp, _ in res), sum(l == l0 for _, l in res), len(res)), flush=True)

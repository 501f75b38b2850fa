"""One training epoch through the Runner with the training set in HBM (device batch assembly, Trainer.run_epoch) and with the reference's
own data path (process_batch on the host behind a prefetch thread, one pinned upload per step: hual_amd/feeder.py) - what the host
side costs.   python scripts/bench_feed_modes.py [N] [batch]"""
import json, os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
import al_synth
from hual_amd import al, data
from hual_amd.runner import Runner

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
vdim, max_vlen = 1024, 128
recs, vis, data_gt, _ = al_synth.make_trainset(N, 400, vdim, max_vlen, seed=5, num_words=1000, num_chars=40, max_words=20)
for r, g in zip(recs, data_gt):
    s, e = data.time_to_index(g[2][0], g[2][1], r['v_len'], r['duration'])[:2]
    r['s_ind'], r['e_ind'] = int(s), int(e)
cfg = dict(task='synth', train=dict(batch_size=bs, droprate=0.2, lr=1e-4, epochs=1, clip_norm=1.0),
           model=dict(vdim=vdim, dim=128, num_heads=8, word_dim=300, char_dim=50, max_vlen=max_vlen, attn_layer=2),
           loss=dict(match_lambda=1.0, tau=0.3, no_gumbel=True), num_chars=40)
wv = np.random.default_rng(0).normal(0, 0.4, size=(998, 300)).astype(np.float32)
class L:
    def info(self, s): pass
out = dict(samples=N, batch=bs, vdim=vdim, max_vlen=max_vlen)
# the host's share alone: building the padded batches of one epoch
t0 = time.perf_counter()
for lo in range(0, N, bs):
    data.process_train_batch(recs[lo:lo + bs], vis)
out['host_process_batch_ms_per_step'] = round((time.perf_counter() - t0) / ((N + bs - 1) // bs) * 1e3, 3)
for feed in ('device', 'host'):
    r = Runner(cfg, wv, recs, None, vis, ckpt_dir='/tmp/ck_' + feed, logger=L(), feed=feed)
    ms = []
    for ep in range(int(os.environ.get("EPOCHS", "4"))):
        t0 = time.perf_counter()
        r.train_epoch(1e-4)
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t0) / ((N + bs - 1) // bs) * 1e3)
    out[feed] = dict(ms_per_step_by_epoch=[round(x, 3) for x in ms], clips_per_s_last=round(N / (ms[-1] * ((N + bs - 1) // bs) / 1e3), 1))
    del r
    torch.cuda.empty_cache()
print(json.dumps(out))

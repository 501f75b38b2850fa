"""Where does the rare NaN of the runner test's training task come from?  Replays the task step by step (Trainer, graphs on or off),
checks the parameters after every step; on the first NaN restores the state of that step, re-runs forward + backward with debug
taps and lists the intermediates / gradient tensors that hold NaN.   scripts/exp/runner_nan.py [graph 0|1] [max runs]"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import test_gpu_runner as tr
from hual_amd.runner import Runner
from hual_amd.train import Trainer
use_graph = (sys.argv[1] != '0') if len(sys.argv) > 1 else True
nruns = int(sys.argv[2]) if len(sys.argv) > 2 else 200
vdim = 64
vis = tr._videos(24, vdim, 0); train = tr._task(192, vis, 1); test = tr._task(64, vis, 2)
cfg = dict(task='synth', train=dict(batch_size=32, droprate=0.1, lr=2e-3, epochs=8, clip_norm=1.0),
           model=dict(vdim=vdim, dim=128, num_heads=8, word_dim=300, char_dim=50, max_vlen=32, attn_layer=2),
           loss=dict(match_lambda=1.0, tau=0.3, no_gumbel=True), num_chars=10)
wv = np.random.default_rng(0).normal(0, 0.4, size=(40, 300)).astype(np.float32)
class L:
    def info(self, s): pass
for run in range(nruns):
    r = Runner(cfg, wv, train, test, vis, ckpt_dir='/tmp/ckpt_nan', logger=L())
    m, ds = r.model, r.train_set
    t = Trainer(m, world=1, use_graph=use_graph)
    rnd = random.Random(12345 + run)
    step = 0
    for epoch in range(8):
        lr = 2e-3 * (1.0 - epoch / 8)
        order = list(range(len(ds))); rnd.shuffle(order)
        order = np.asarray(order, dtype=np.int32)
        for lo in range(0, len(order), 32):
            sel = order[lo:lo + 32]
            feeds = ds.assemble(sel, min_chars=4)
            t.set_batch_device(feeds)
            snap = [x.clone() for x in (m.params, m.adam_m, m.adam_v, m.rng_state)]
            t.step(lr=lr, drop_rate=0.1)
            torch.cuda.synchronize()
            step += 1
            if bool(torch.isnan(m.params).any()):
                B, T = feeds['video'].shape[:2]; Lq, C = feeds['char_ids'].shape[1:]
                print('run %d epoch %d step %d: NaN parameters after a step at B%d T%d L%d C%d, lens %s' % (run, epoch, step, B, T, Lq, C, feeds['video_seq_len'].cpu().numpy().tolist()))
                for x, s in zip((m.params, m.adam_m, m.adam_v, m.rng_state), snap): x.copy_(s)
                m.debug_taps = True
                o = m.forward(feeds['video'], feeds['video_seq_len'], feeds['word_ids'], feeds['char_ids'], drop_rate=0.1,
                              labels=(feeds['y1'], feeds['y2'], feeds['match_labels'], feeds['inner_labels']))
                torch.cuda.synchronize()
                print('  eager replay of the step: loss', float(o['loss']), float(o['loc_loss']), float(o['match_loss']), float(o['align_loss']))
                bad = []
                for name, (off, rows, cols) in sorted(m._ws_table.items(), key=lambda kv: kv[1][0]):
                    if cols <= 0 or rows <= 0 or name.startswith('d.') or '.rb' in name or '.kb' in name or 'keep' in name: continue
                    try:
                        v = m.tap(name)
                    except Exception:
                        continue
                    n = int(torch.isnan(v).sum())
                    if n: bad.append((name, n, v.numel()))
                print('  forward taps with NaN:', bad[:12])
                m.backward(); torch.cuda.synchronize()
                g = m.table.unpack(m.grads.cpu().numpy())
                print('  gradient tensors with NaN:', [k for k, v in g.items() if np.isnan(v).any()][:12], 'of', len(g))
                sys.exit(0)
print('no NaN in %d runs (graph %s)' % (nruns, use_graph))

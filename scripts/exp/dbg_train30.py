"""debug: per-step pin audit of the 30-step teacher-forced run (tests/test_gpu_train.py)"""
import collections, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import parity_util as pu
from oracle import seqpan_ref as R
from hual_amd.train import Trainer
use_graph = os.environ.get('GRAPH', '1') == '1'
lr, drop, seed, off = 1e-3, 0.2, 99, 5
cfg, p, wv, b, labels = pu.make_case(B=4, T=24, L=7, C=5, seed=21)
B, T, L = 4, 24, 7
m = pu.hip_model(cfg, p, wv); m.set_rng(seed, off)
tr = Trainer(m, world=1, use_graph=use_graph)
tr.set_batch(b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy(), *[x.numpy() for x in labels])
rp = collections.OrderedDict((k, v.clone()) for k, v in p.items())
rm = {k: torch.zeros_like(v) for k, v in p.items()}; rv = {k: torch.zeros_like(v) for k, v in p.items()}
batch = (b['video'], b['lens'], b['word_ids'], b['char_ids'])
for s in range(int(os.environ.get('STEPS', '30'))):
    tr.step(lr=lr, drop_rate=drop); torch.cuda.synchronize()
    pins = pu.relu_pins(m, B, T, L)
    rp, rm, rv, info = R.train_step(rp, rm, rv, cfg, wv, batch, labels, lr, drop, seed=seed, offset=off + s, relu_pin=pins, want_tap=True)
    tap = info['tap']
    try:
        n, total = pu.audit_pins(tap, pins); msg = 'ok n=%d' % n
    except AssertionError as e:
        msg = 'FAIL ' + str(e)[:120]
    print(s, 'loss hip %.5f oracle %.5f' % (float(tr.last_loss()), float(info['loss'])), msg, flush=True)
    got = m.state_dict()
    rp = collections.OrderedDict((k, torch.from_numpy(got[k])) for k in rp)
    rm = {k: torch.from_numpy(a) for k, a in m.table.unpack(m.adam_m.cpu().numpy()).items()}
    rv = {k: torch.from_numpy(a) for k, a in m.table.unpack(m.adam_v.cpu().numpy()).items()}

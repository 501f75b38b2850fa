"""margins of tests/test_gpu_train.py::test_twenty_step_trajectory_against_free_running_oracle over repeated HIP runs (the float64 oracle trajectory
is computed once): worst ratio of the logit deviation to its envelope 1e-3 . 1.5^step, worst loss differences against their bars, span equality"""
import sys, os
import numpy as np, torch
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, 'tests'))
import parity_util as pu
import test_gpu_train as T
from hual_amd.train import Trainer
lr, drop, seed, off, steps = 1e-4, 0.2, 31, 11, 14
cfg, p, wv, b, labels = pu.make_case(B=4, T=24, L=7, C=5, seed=21)
o64 = T._free_run(cfg, p, wv, b, labels, torch.float64, lr, drop, seed, off, steps)
worst_env = np.zeros(steps); worst_loss = np.zeros(steps); span_bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    m = pu.hip_model(cfg, p, wv); m.set_rng(seed, off)
    tr = Trainer(m, world=1, use_graph=True)
    tr.set_batch(b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy(), *[x.numpy() for x in labels])
    for s in range(steps):
        tr.step(lr=lr, drop_rate=drop); torch.cuda.synchronize()
        d = max(float((tr.start_logits.cpu().double() - o64[s][1]).abs().max()), float((tr.end_logits.cpu().double() - o64[s][2]).abs().max()))
        worst_env[s] = max(worst_env[s], d / (1e-3 * 1.5 ** s))
        worst_loss[s] = max(worst_loss[s], abs(float(tr.last_loss()) - o64[s][0]) / max(abs(o64[s][0]), 1.0))
        if s < 4 and not (torch.equal(tr.start_index.cpu(), o64[s][3]) and torch.equal(tr.end_index.cpu(), o64[s][4])): span_bad += 1
print('deviation / envelope per step (must stay <= 1):', np.round(worst_env, 3).tolist())
print('loss rel diff per step (bars 1e-3 for steps 0-3, 1e-2 for 4-7):', ['%.1e' % x for x in worst_loss[:8]])
print('span mismatches in steps 0-3:', span_bad)

"""GPU box: repeat the 3-step training run of tests/test_gpu_train.py several times and print the losses (run-to-run
variation comes from float atomics; large jumps mean the trajectory is chaotic at this learning rate)."""
import sys
import numpy as np
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import parity_util as pu
from hual_amd.train import Trainer

lr = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-3
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
drop, seed, off = 0.2, 99, 5
cfg, p, wv, b, labels = pu.well_conditioned_case(drop_rate=drop, rng_seed=seed, rng_offset=off, B=4, T=24, L=7, C=5, seed=21)
for graph in (False, True):
    for rep in range(reps):
        m = pu.hip_model(cfg, p, wv)
        m.set_rng(seed, off)
        tr = Trainer(m, world=1, use_graph=graph)
        tr.set_batch(b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy(), *[x.numpy() for x in labels])
        losses = []
        for s in range(4):
            tr.step(lr=lr, drop_rate=drop)
            losses.append(float(tr.last_loss()))
            if s == 0:
                got = m.state_dict()      # as the test does: a host pause + D2H copy between replays
        flag = '' if abs(losses[2] - 23.286) < 0.01 or lr != 1e-3 else '   <<<<<< OUTLIER'
        print('graph=%d rep %d' % (graph, rep), ' '.join('%.6f' % x for x in losses), flag, flush=True)

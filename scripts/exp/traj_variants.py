"""the c1 free-running trajectory (tests/test_gpu_train.py::test_c1_trajectory_matches_the_clean_fp64_oracle) for several builds of
the library: traj_variants.py oracle -> gpurun_out/traj_o64.json ; traj_variants.py hip <label> -> one line of relative deviations.
Each build runs in its own process (the library is loaded once per process): cp <lib> hual_amd/libhual_seqpan.so first."""
import sys, os, json
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'tests'))
import parity_util as pu
lr, drop, seed, off, steps = 1e-4, 0.2, 1, 1, 10
cfg, p, wv, b, labels = pu.make_case(B=16, T=64, L=20, C=8, seed=12345, max_vlen=64, vdim=1024, num_words=1000)
out = os.path.join(R, 'gpurun_out', 'traj_o64.json')
if sys.argv[1] == 'oracle':
    import test_gpu_train as tg
    o64 = tg._free_run(cfg, p, wv, b, labels, torch.float64, lr, drop, seed, off, steps)
    json.dump([float(x[0]) for x in o64], open(out, 'w'))
    print('oracle', [round(float(x[0]), 5) for x in o64])
    o32 = tg._free_run(cfg, p, wv, b, labels, torch.float32, lr, drop, seed, off, steps)
    print('%-10s' % 'f32oracle', ' '.join('%.1e' % (abs(float(a[0]) - float(r[0])) / max(abs(float(r[0])), 1.0)) for a, r in zip(o32, o64)))
else:
    from hual_amd.train import Trainer
    o = json.load(open(out))
    m = pu.hip_model(cfg, p, wv)
    m.set_rng(seed, off)
    tr = Trainer(m, world=1, use_graph=True)
    tr.set_batch(b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy(), *[x.numpy() for x in labels])
    rel = []
    for s in range(steps):
        tr.step(lr=lr, drop_rate=drop)
        torch.cuda.synchronize()
        rel.append(abs(float(tr.last_loss()) - o[s]) / max(abs(o[s]), 1.0))
    print('%-10s' % sys.argv[2], ' '.join('%.1e' % r for r in rel))

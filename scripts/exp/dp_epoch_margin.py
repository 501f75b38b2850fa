"""how close to its bars does tests/test_gpu_dp_epoch.py run?  the single-process run against ITSELF, repeated: share of parameters within
0.1 lr after the two epochs, span agreement of the first epoch"""
import sys, os
import numpy as np, torch
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, 'tests'))
import test_gpu_dp_epoch as T
ref = T._run(1)
fr, sm = [], []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    cur = T._run(1)
    d = np.abs(cur[2] - ref[2])
    fr.append(float(np.mean(d <= 0.1 * T.LR)))
    sm.append(float(np.mean((cur[3][0][0] == ref[3][0][0]) & (cur[3][0][1] == ref[3][0][1]))))
print('share of parameters within 0.1 lr: min %.4f median %.4f; first-epoch spans equal: min %.3f median %.3f (bars: 0.9 / 0.9)' % (min(fr), np.median(fr), min(sm), np.median(sm)))

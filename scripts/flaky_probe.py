"""GPU box: call the train-step test functions in a loop inside one process (flakiness probe)."""
import sys
import traceback
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import test_gpu_train as t

order = sys.argv[1] if len(sys.argv) > 1 else 'FT'
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
    for ch in order:
        g = ch == 'T'
        try:
            t.test_three_train_steps_match_oracle(g)
            print('rep %d graph=%d ok' % (rep, g), flush=True)
        except AssertionError as e:
            msg = [l for l in str(e).splitlines() if 'ACTUAL' in l or "('" in l]
            print('rep %d graph=%d FAIL %s' % (rep, g, msg[:1]), flush=True)

"""print ms/step and the per-kernel table of a bench.py JSON line"""
import json
import sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["ms_per_step"], 'ms/step;  roofline:', {k: v for k, v in d["roofline"].items() if k != 'families'})
for f in d["roofline"]["families"]:
    print('%-34s x%-3d %8.1f us  %s' % (f['kernel'], f['launches_per_step'], f['us_per_step'], f['tflops'] or ''))

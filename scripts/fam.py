import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
print(d['ms_per_step'], d['value'], d['roofline']['step']['launches_per_step'])
for f in d['roofline']['families']:
    if len(sys.argv)<3 or any(k in f['kernel'] for k in sys.argv[2:]): print(f['kernel'], f['launches_per_step'], f['us_per_step'])

import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]);print(d["ms_per_step"])
for f in d["roofline"]["families"]: print('%-24s x%-3d %8.1f' % (f['kernel'], f['launches_per_step'], f['us_per_step']))

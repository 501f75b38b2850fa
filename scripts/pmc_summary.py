"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel name: mean of each counter per dispatch.
usage: python scripts/pmc_summary.py 'gpurun_out/pmc/*counter_collection.csv' [name-filter]"""
import csv
import glob
import re
import sys
from collections import defaultdict

rows = defaultdict(lambda: defaultdict(list))
for p in glob.glob(sys.argv[1]):
    with open(p) as f:
        for r in csv.DictReader(f):
            name = re.sub(r'\(.*$', '', r['Kernel_Name']).replace('void ', '').replace('hual::', '')
            rows[name][r['Counter_Name']].append(float(r['Counter_Value']))
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for name in sorted(rows):
    if flt not in name:
        continue
    c = rows[name]
    n = max(len(v) for v in c.values())
    print('%-46s dispatches %d' % (name, n))
    for k in sorted(c):
        v = c[k]
        print('    %-32s mean %14.1f   sum %16.0f' % (k, sum(v) / len(v), sum(v)))

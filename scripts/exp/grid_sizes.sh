#!/bin/bash
# workgroup count and size of every launch of a step (kernel trace of a short bench run): scripts/exp/grid_sizes.sh [bench args]
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $out/prof_grid
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/prof_grid -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --prewarm 5 --no-cpu-baseline --no-roofline --no-epoch-loop --no-graph "$@" > /dev/null 2> $out/grid_rocprof.err || { tail -5 $out/grid_rocprof.err; exit 2; }
tr=$(ls $out/prof_grid/*/*kernel_trace.csv $out/prof_grid/*kernel_trace.csv 2>/dev/null | head -1)
python3 - "$tr" <<'PY'
import csv, sys, re, collections
c = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r'\(.*$', '', r['Kernel_Name']).replace('void ', '').replace('hual::', '').strip()
    gx, gy, gz = int(r['Grid_Size_X']), int(r['Grid_Size_Y']), int(r['Grid_Size_Z'])
    wx, wy, wz = int(r['Workgroup_Size_X']), int(r['Workgroup_Size_Y']), int(r['Workgroup_Size_Z'])
    key = (name, gx * gy * gz // (wx * wy * wz), wx * wy * wz, r.get('LDS_Block_Size', r.get('LDS_Block_Size_v', '')))
    c[key] = c.get(key, 0) + 1
for (name, wgs, wsz, lds), n in c.items():
    print('%-46s workgroups %5d x %4d threads  lds %7s  launches %d' % (name[:46], wgs, wsz, lds, n))
PY
rm -rf $out/prof_grid

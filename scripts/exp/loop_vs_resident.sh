#!/bin/bash
# per-kernel durations of the SAME padded shape (B16 T100 L30 C11) as a resident-batch replay and inside the epoch loop (one-shape set)
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $out/prof_lvr_a $out/prof_lvr_b
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/prof_lvr_a -o p --output-format csv -- python3 $R/bench.py --batch 16 --T 100 --L 30 --C 11 --steps 300 --no-cpu-baseline --no-roofline --no-epoch-loop > /dev/null 2> $out/lvr_a.err || { tail -3 $out/lvr_a.err; exit 2; }
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/prof_lvr_b -o p --output-format csv -- python3 $R/scripts/exp/epoch_oneshape.py --only-one > $out/lvr_b.log 2> $out/lvr_b.err || { tail -3 $out/lvr_b.err; exit 2; }
python3 - $(ls $out/prof_lvr_a/*/*kernel_trace.csv $out/prof_lvr_a/*kernel_trace.csv 2>/dev/null | head -1) $(ls $out/prof_lvr_b/*/*kernel_trace.csv $out/prof_lvr_b/*kernel_trace.csv 2>/dev/null | head -1) <<'PY'
import csv, sys, re
from collections import defaultdict
def load(f):
    rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), re.sub(r'\(.*$', '', r['Kernel_Name']).replace('void ', '').replace('hual::', '')[:56]) for r in csv.DictReader(open(f))]
    rows.sort()
    rows = rows[len(rows) // 2:]                      # steady state: the second half of the run
    d = defaultdict(lambda: [0, 0])
    for s, e, n in rows:
        d[n][0] += 1; d[n][1] += e - s
    steps = d['pack_weights_kernel'][0]
    gaps = sum(max(0, rows[i + 1][0] - rows[i][1]) for i in range(len(rows) - 1))
    return d, steps, gaps
a, sa, ga = load(sys.argv[1]); b, sb, gb = load(sys.argv[2])
print('steps: resident %d, loop %d; idle per step: resident %.1f us, loop %.1f us' % (sa, sb, ga / sa / 1e3, gb / sb / 1e3))
ta = sum(v[1] for v in a.values()) / sa / 1e3; tb = sum(v[1] for v in b.values()) / sb / 1e3
print('kernel time per step: resident %.1f us, loop %.1f us' % (ta, tb))
for k in sorted(set(a) | set(b), key=lambda k: -(b.get(k, [0, 0])[1] / sb - a.get(k, [0, 0])[1] / sa)):
    xa = a.get(k, [0, 0])[1] / sa / 1e3; xb = b.get(k, [0, 0])[1] / sb / 1e3
    if abs(xb - xa) >= 0.4: print('%-58s resident %7.1f  loop %7.1f  %+6.1f' % (k, xa, xb, xb - xa))
PY
rm -rf $out/prof_lvr_a $out/prof_lvr_b

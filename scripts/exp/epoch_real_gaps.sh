#!/bin/bash
# device-side view of the epoch loop on the reference's shape distribution (scripts/exp/epoch_real.py): per step (between two assembly
# launches) span / kernel time / idle, and the per-kernel totals.  usage: epoch_real_gaps.sh <mode> [epoch_real args]
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out
mode=${1:-limit100000}; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $out/prof_erg
timeout -k 10 400 rocprofv3 --kernel-trace -d $out/prof_erg -o p --output-format csv -- python3 $R/scripts/exp/epoch_real.py --modes $mode "$@" > $out/erg_$mode.log 2> $out/erg.err || { tail -5 $out/erg.err; exit 2; }
tr=$(ls $out/prof_erg/*/*kernel_trace.csv $out/prof_erg/*kernel_trace.csv 2>/dev/null | head -1)
python3 - "$tr" <<'PY'
import csv, sys, re
from collections import defaultdict
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
idx = [i for i, r in enumerate(rows) if 'assemble' in r[2]]
print('assembly launches', len(idx))
idx = idx[len(idx) * 2 // 3:]          # the last third of the run: steady state
spans, per, gaps = [], defaultdict(lambda: [0, 0]), defaultdict(int)
for a, b in zip(idx[:-1], idx[1:]):
    seg = rows[a:b]
    span = rows[b][0] - seg[0][0]
    if span > 5e6: continue            # epoch boundary (span fetch, host shuffles)
    busy = sum(e - s for s, e, _ in seg)
    spans.append((span, busy, len(seg)))
    gaps['assemble -> first kernel of the step'] += seg[1][0] - seg[0][1]
    gaps['last kernel of the step -> next assemble'] += rows[b][0] - seg[-1][1]
    gaps['between kernels of the step'] += sum(seg[k + 1][0] - seg[k][1] for k in range(1, len(seg) - 1))
    for s, e, nm in seg:
        nm = re.sub(r'\(.*$', '', nm).replace('void ', '').replace('hual::', '')[:60]
        per[nm][0] += 1; per[nm][1] += e - s
n = len(spans)
print('steps analysed %d: mean span %.1f us, kernel time %.1f us, idle %.1f us, kernels/step %.1f' % (n, sum(s[0] for s in spans) / n / 1e3, sum(s[1] for s in spans) / n / 1e3, sum(s[0] - s[1] for s in spans) / n / 1e3, sum(s[2] for s in spans) / n))
for k, v in gaps.items():
    print('  idle %-45s %7.1f us/step' % (k, v / n / 1e3))
for nm, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:48]:
    print('%-62s x%6.2f/step %8.1f us/step (avg %6.2f)' % (nm, c / n, t / n / 1e3, t / c / 1e3))
PY
rm -rf $out/prof_erg
tail -3 $out/erg_$mode.log

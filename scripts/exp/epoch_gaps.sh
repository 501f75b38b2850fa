#!/bin/bash
# where the epoch loop's time goes on the device: between consecutive assembly launches, the kernel time, the idle time and the largest gaps
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $out/prof_eg
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace -d $out/prof_eg -o p --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --prewarm 10 --no-cpu-baseline --no-roofline > $out/eg_bench.json 2> $out/eg.err || { tail -5 $out/eg.err; exit 2; }
tr=$(ls $out/prof_eg/*/*kernel_trace.csv $out/prof_eg/*kernel_trace.csv 2>/dev/null | head -1)
python3 - "$tr" <<'PY'
import csv, sys
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
idx = [i for i, r in enumerate(rows) if 'assemble_kernel' in r[2]]
print('assembly launches', len(idx))
spans = []
for a, b in zip(idx[:-1], idx[1:]):
    seg = rows[a:b]
    if len(seg) < 50 or len(seg) > 80: continue
    span = rows[b][0] - seg[0][0]
    busy = sum(e - s for s, e, _ in seg)
    gaps = sorted(((seg[k + 1][0] if k + 1 < len(seg) else rows[b][0]) - seg[k][1], seg[k][2][:40], (seg[k + 1][2] if k + 1 < len(seg) else rows[b][2])[:40]) for k in range(len(seg)))[::-1]
    spans.append((span, busy, len(seg), gaps[:3]))
spans = spans[len(spans) // 2:]
n = len(spans)
print('steps analysed', n, ' mean span %.1f us  mean kernel time %.1f us  mean idle %.1f us  kernels/step %.1f' % (sum(s[0] for s in spans) / n / 1e3, sum(s[1] for s in spans) / n / 1e3, sum(s[0] - s[1] for s in spans) / n / 1e3, sum(s[2] for s in spans) / n))
for s in spans[:4]:
    print('  span %.1f busy %.1f  largest gaps: %s' % (s[0] / 1e3, s[1] / 1e3, '; '.join('%.1f us after %s before %s' % (g[0] / 1e3, g[1], g[2]) for g in s[3])))
PY
rm -rf $out/prof_eg

"""Per-kernel timeline of ONE training step from a rocprofv3 --kernel-trace CSV.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace -d gpurun_out/tl -o tl --output-format csv -- python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline
    python scripts/step_timeline.py gpurun_out/tl/*kernel_trace.csv [--steps 20] [--group]

A step starts at every `pack_weights_kernel` (the first launch of the forward; `prep_masks_kernel` in older traces); --steps complete steps from the middle of the run are reduced (median) position by position
(duration and the idle gap in front of the kernel).
"""
import argparse
import csv
import glob
import re
import sys
from collections import Counter, defaultdict


def short(name):
    name = re.sub(r'\(.*$', '', name)
    name = re.sub(r'^void ', '', name)
    name = name.replace('hual::', '').replace('(anonymous namespace)::', '')
    return name[:48]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('csv')
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--group', action='store_true', help='only the per-kernel-name totals')
    ap.add_argument('--stats', default=None, help='write per-kernel launch statistics of the selected step shape to this CSV')
    a = ap.parse_args()
    paths = glob.glob(a.csv)
    rows = []
    for p in paths:
        with open(p) as f:
            for r in csv.DictReader(f):
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    first = 'pack_weights_kernel' if any('pack_weights_kernel' in r[2] for r in rows) else 'prep_masks_kernel'
    starts = [i for i, r in enumerate(rows) if first in r[2]]
    if len(starts) < 3:
        sys.exit('no steps found')
    steps = [rows[starts[i]:starts[i + 1]] for i in range(len(starts) - 1)]
    # a run may hold steps of several shapes (bench.py's gpu_at_cpu_shape leg runs the c1 shape after the timed c2 steps):
    # template arguments differ between shapes (attn_fwd_kernel<8> vs <4>), so the most frequent launch sequence is the
    # timed workload
    sigs = Counter(tuple(k[2] for k in s) for s in steps)
    sig = sigs.most_common(1)[0][0]
    n = len(sig)
    same = [s for s in steps if tuple(k[2] for k in s) == sig]
    print('%d kernels per step, %d steps with this launch sequence in the trace (of %d), median of %d from the middle of the run'
          % (n, len(same), len(steps), min(a.steps, len(same))))
    if a.stats:
        # per-kernel launch statistics over ALL steps of that sequence (the profiler's own --stats file mixes shapes)
        acc = defaultdict(list)
        for s in same:
            for k in s:
                acc[k[2]].append(k[1] - k[0])
        with open(a.stats, 'w') as f:
            f.write('"Name","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs"\n')
            for nm, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
                f.write('"%s",%d,%d,%.3f,%d,%d\n' % (nm, len(v), sum(v), sum(v) / len(v), min(v), max(v)))
    # from the MIDDLE of the run: the trace of a whole bench.py run ends with the legs behind the timed region (the roofline leg
    # launches the same sequence eagerly, with events: ~2 us of host gap per launch) - the middle is the timed, replayed workload
    mid = len(same) // 2
    steps = same[max(0, mid - a.steps // 2):max(0, mid - a.steps // 2) + a.steps] if len(same) > 2 * a.steps else same[-a.steps:]
    tot = defaultdict(lambda: [0, 0.0, 0.0])
    t_sum = g_sum = 0.0
    def med(v):      # (median over the selected steps: one profiler-buffer flush inside a step is a 4 ms gap in a 1.2 ms step)
        v = sorted(v)
        return (v[(len(v) - 1) // 2] + v[len(v) // 2]) / 2.0
    for k in range(n):
        dur = med([s[k][1] - s[k][0] for s in steps]) / 1e3
        gap = med([(s[k][0] - s[k - 1][1]) for s in steps]) / 1e3 if k else 0.0
        nm = short(steps[0][k][2])
        tot[nm][0] += 1
        tot[nm][1] += dur
        tot[nm][2] += gap
        t_sum += dur
        g_sum += gap
        if not a.group:
            print('%4d %-50s %8.2f us  gap %6.2f' % (k, nm, dur, gap))
    span = med([s[-1][1] - s[0][0] for s in steps]) / 1e3
    print('--- totals: kernel time %.1f us, gaps %.1f us, first-to-last span %.1f us' % (t_sum, g_sum, span))
    for nm, (cnt, d, g) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print('%-50s x%-3d %9.1f us  (avg %6.2f)  gaps %7.1f' % (nm, cnt, d, d / cnt, g))


if __name__ == '__main__':
    main()

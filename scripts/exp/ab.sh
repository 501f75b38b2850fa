#!/bin/bash
# same-box A/B of two builds of the library: the rocprofv3 step timeline of each (twice, interleaved), per-kernel totals side by side
#   scripts/exp/ab.sh <tag> <base.so> <new.so> [bench args]        (paths relative to the repo root; "cur" = the in-tree library)
tag=$1; A=$2; B=$3; shift; shift; shift
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out
[ "$A" = cur ] && A=hual_amd/libhual_seqpan.so
[ "$B" = cur ] && B=hual_amd/libhual_seqpan.so
cd /tmp && export TMPDIR=/tmp
run() {   # name lib
  rm -rf $out/prof_$1
  HUAL_LIB_PATH=$R/$2 timeout -k 10 300 rocprofv3 --kernel-trace -d $out/prof_$1 -o p --output-format csv -- python3 $R/bench.py --steps 60 --warmup 5 --prewarm 60 --no-cpu-baseline --no-roofline --no-epoch-loop "${@:3}" > /dev/null 2> $out/${1}_rocprof.err || { tail -5 $out/${1}_rocprof.err; exit 2; }
  tr=$(ls $out/prof_$1/*/*kernel_trace.csv $out/prof_$1/*kernel_trace.csv 2>/dev/null | head -1)
  python3 $R/scripts/step_timeline.py "$tr" --group > $out/${1}_tl.txt 2>&1
  rm -rf $out/prof_$1
}
run ${tag}_a1 $A "$@" && run ${tag}_b1 $B "$@" && run ${tag}_a2 $A "$@" && run ${tag}_b2 $B "$@" || exit 2
python3 - $out/${tag} <<'PY'
import re, sys
p = sys.argv[1]
def load(f):
    d, tot = {}, None
    for line in open(f):
        m = re.match(r'^(\S.*?)\s+x(\d+)\s+([\d.]+) us', line)
        if m: d[m.group(1)] = (int(m.group(2)), float(m.group(3)))
        m = re.match(r'^--- totals: kernel time ([\d.]+) us', line)
        if m: tot = float(m.group(1))
    return d, tot
a1, ta1 = load(p + '_a1_tl.txt'); a2, ta2 = load(p + '_a2_tl.txt'); b1, tb1 = load(p + '_b1_tl.txt'); b2, tb2 = load(p + '_b2_tl.txt')
print('kernel time per step: A %.1f / %.1f us   B %.1f / %.1f us   (B - A = %+.1f us)' % (ta1, ta2, tb1, tb2, (tb1 + tb2 - ta1 - ta2) / 2))
keys = sorted(set(a1) | set(b1), key=lambda k: -(a1.get(k, (0, 0))[1] + b1.get(k, (0, 0))[1]))
for k in keys:
    xa = [d.get(k, (0, 0.0)) for d in (a1, a2)]; xb = [d.get(k, (0, 0.0)) for d in (b1, b2)]
    da = (xb[0][1] + xb[1][1] - xa[0][1] - xa[1][1]) / 2
    flag = '   <--' if abs(da) >= 1.0 else ''
    print('%-46s x%-2d A %7.1f %7.1f   x%-2d B %7.1f %7.1f   %+6.1f%s' % (k, xa[0][0], xa[0][1], xa[1][1], xb[0][0], xb[0][1], xb[1][1], da, flag))
PY

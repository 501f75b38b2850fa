#!/bin/bash
# same-box A/B of the in-tree library (old) against a variant library (new): scripts/exp/ab_var.sh <tag> <variant.so relative to the repo> [bench args]
tag=$1; var=$2; shift; shift
R=$GRAFT_REPO_ROOT
bash $R/scripts/exp/tl_shape.sh ${tag}_old "$@" > /dev/null || exit 2
HUAL_LIB_PATH=$R/$var bash $R/scripts/exp/tl_shape.sh ${tag}_new "$@" > /dev/null || exit 2
bash $R/scripts/exp/tl_shape.sh ${tag}_old2 "$@" > /dev/null || exit 2
HUAL_LIB_PATH=$R/$var bash $R/scripts/exp/tl_shape.sh ${tag}_new2 "$@" > /dev/null || exit 2
python3 - $R/gpurun_out/$tag <<'PY'
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
a1, ta1 = load(p + '_old_step_timeline.txt'); a2, ta2 = load(p + '_old2_step_timeline.txt'); b1, tb1 = load(p + '_new_step_timeline.txt'); b2, tb2 = load(p + '_new2_step_timeline.txt')
print('kernel time per step: old %.1f / %.1f us   new %.1f / %.1f us   (new - old = %+.1f us)' % (ta1, ta2, tb1, tb2, (tb1 + tb2 - ta1 - ta2) / 2))
keys = sorted(set(a1) | set(b1), key=lambda k: -(a1.get(k, (0, 0))[1] + b1.get(k, (0, 0))[1]))
for k in keys:
    xa = [d.get(k, (0, 0.0)) for d in (a1, a2)]; xb = [d.get(k, (0, 0.0)) for d in (b1, b2)]
    da = (xb[0][1] + xb[1][1] - xa[0][1] - xa[1][1]) / 2
    flag = '   <--' if abs(da) >= 1.0 else ''
    print('%-46s x%-2d old %7.1f %7.1f   x%-2d new %7.1f %7.1f   %+6.1f%s' % (k, xa[0][0], xa[0][1], xa[1][1], xb[0][0], xb[0][1], xb[1][1], da, flag))
PY

R=$GRAFT_REPO_ROOT; out=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $out/prof_ep
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/prof_ep -o p --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --prewarm 10 --no-cpu-baseline --no-roofline > $out/ep_bench.json 2> $out/ep.err || { tail -5 $out/ep.err; exit 2; }
st=$(ls $out/prof_ep/*/*kernel_stats.csv $out/prof_ep/*kernel_stats.csv 2>/dev/null | head -1)
python3 - "$st" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:70]:
    n = r['Name']
    if any(k in n for k in ('assemble', 'Memcpy', 'memcpy', 'copy', 'elementwise', 'fill', 'Fill', 'index', 'gather', 'label', 'span', 'cat', 'vectorized')):
        print('%-90s calls %6s  avg %9.1f us  total %9.1f us' % (n[:90], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3))
PY
rm -rf $out/prof_ep
python3 -c "
import json;d=json.load(open('$out/ep_bench.json'));print(d['ms_per_step'], d['epoch_loop']['ms_per_step'], d['epoch_loop']['steps'])"

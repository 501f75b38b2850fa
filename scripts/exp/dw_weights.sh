# GPU box: balanced weight-gradient launch by tile weights (plain,prod,drop)
for w in $DW_WEIGHTS_LIST; do
  HUAL_DW_WEIGHTS=$w python bench.py --steps 300 --warmup 20 --no-cpu-baseline > gpurun_out/dbg.json 2>/dev/null
  python - "$w" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/dbg.json').read().strip().splitlines()[-1])
r=d['roofline']
k=[x for x in r['families'] if 'dw_bf16' in x['kernel']]
print('weights', sys.argv[1], 'step', d['ms_per_step'], k[0]['kernel'], k[0]['us_per_step'])
PY
done

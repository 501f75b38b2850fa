# GPU box: weight-gradient launch time with parts of the kernel switched off (HUAL_DW_DBG bits: 1 no products, 2 no
# global loads, 4 no atomics, 16 no barrier) and by workgroup count
for cfg in $DW_DBG_CFGS; do
  HUAL_DW_DBG=${cfg%%:*} HUAL_DW_BLOCKS=${cfg##*:} python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/dbg.json 2>/dev/null
  python - "${cfg%%:*}" "${cfg##*:}" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/dbg.json').read().strip().splitlines()[-1])
r=d['roofline']
k=[x for x in r['families'] if 'dw_bf16' in x['kernel']]
print('dbg', sys.argv[1], 'blocks', sys.argv[2], 'step', d['ms_per_step'], k[0]['kernel'], k[0]['us_per_step'])
PY
done

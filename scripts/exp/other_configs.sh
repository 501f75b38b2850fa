# GPU step time at the other BASELINE.json shapes (per-GPU batch): c1 B16 T64, c4 B32 T256 (max_vlen 256), and vdim 512
for cfg in "--batch 16 --T 64" "--batch 32 --T 256" "--batch 64 --T 128 --vdim 512" "--batch 64 --T 100 --L 30"; do
  echo "$cfg: $(python bench.py --steps 50 --warmup 10 --prewarm 200 --no-cpu-baseline --no-roofline --no-epoch-loop $cfg 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step", d["value"], "clips/s")')"
done

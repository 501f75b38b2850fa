for d in 0.0 0.2; do
  echo "drop=$d $(python bench.py --steps 50 --warmup 10 --prewarm 200 --no-cpu-baseline --no-roofline --drop $d 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
done

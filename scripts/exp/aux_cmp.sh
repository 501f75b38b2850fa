for f in "" "--no-aux" "" "--no-aux"; do
  echo "flags=[$f] $(python bench.py --steps 100 --warmup 10 --prewarm 300 --no-cpu-baseline --no-roofline $f 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
done

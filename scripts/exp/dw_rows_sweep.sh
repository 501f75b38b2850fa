for cfg in "0 64" "0 160" "0 320" "512 0" "768 0" "1024 0" "384 0"; do
  set -- $cfg
  if [ "$1" = "0" ]; then export HUAL_DW_FIXED=$2; unset HUAL_DW_ROWS; else export HUAL_DW_ROWS=$1; fi
  echo "rows=$1 fixed=$2: $(HUAL_DEBUG_DW=0 python bench.py --steps 30 --warmup 5 --prewarm 150 --no-cpu-baseline --no-roofline 2>/dev/null | python -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_step"])')"
done

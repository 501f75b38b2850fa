for rt in 0 2 3; do
  export HUAL_GEMM_RT=$rt
  echo "HUAL_GEMM_RT=$rt $(python bench.py --steps 50 --warmup 10 --prewarm 200 --no-cpu-baseline --no-roofline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
done

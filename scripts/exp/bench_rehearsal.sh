#!/bin/bash
# bench.py --gpus N as N processes on ONE GPU (gloo collectives staged through the host; <= 6 processes may share the card): the N > 1 code
# path with more than two ranks - shard plans, gathers, the one-shot all-reduce among N peers.  The numbers mean nothing.
#   scripts/exp/bench_rehearsal.sh 4
N=${1:-4}
R=$GRAFT_REPO_ROOT
port=$((29300 + RANDOM % 200))
pids=()
for r in $(seq 0 $((N - 1))); do
  HUAL_BENCH_ONE_DEVICE=1 HUAL_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 WORLD_SIZE=$N RANK=$r LOCAL_RANK=$r MASTER_ADDR=127.0.0.1 MASTER_PORT=$port \
    timeout -k 10 500 python $R/bench.py --gpus $N --batch 8 --T 32 --L 8 --C 5 --vdim 256 --steps 4 --warmup 1 --prewarm 2 --no-cpu-baseline \
    --epoch-samples 256 --anet-samples 512 > $R/gpurun_out/rehearsal_$r.out 2> $R/gpurun_out/rehearsal_$r.err &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
echo "exit $rc"
python3 - $R/gpurun_out/rehearsal_0.out <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith('{')][-1]
o = json.loads(line)
print('n_gpus', o['n_gpus'], 'parallelism', o['config']['parallelism'], 'ms/step', o['ms_per_step'])
print('rccl', {k: v for k, v in o['rccl'].items() if k != 'custom_allreduce'})
print('custom', o['rccl'].get('custom_allreduce'))
for k in ('epoch_loop',):
    e = o[k]; print(k, e.get('error') or {x: e[x] for x in ('n_gpus', 'steps', 'ms_per_step', 'step_launch_modes')})
for e in o['epoch_loop_anet']:
    print('anet', e.get('error') or {x: e[x] for x in ('n_gpus', 'steps', 'ms_per_step', 'step_launch_modes', 'distinct_padded_shapes')})
PY
exit $rc

for flags in "" "--no-graph"; do
  echo "flags=[$flags] $(python bench.py --steps 50 --warmup 10 --prewarm 200 --no-cpu-baseline --no-roofline $flags 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"]["launch"])')"
done
HUAL_CHAIN=0 python bench.py --steps 50 --warmup 10 --prewarm 200 --no-cpu-baseline --no-roofline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("nochain", d["ms_per_step"])'
python - <<'PY'
# the data-parallel code path on one rank (force_dp): forward, gather, align, backward, all-reduce(1 rank = no-op), AdamWD - eager
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from hual_amd import lib
from hual_amd.model import SeqPAN
from hual_amd.train import Trainer
cfg = lib.make_cfg(vdim=1024, max_vlen=128, num_words=1000, num_chars=40)
wv = np.random.default_rng(777).normal(0, 0.4, size=(998, 300)).astype(np.float32)
m = SeqPAN(cfg, wv)
b = bench.synth_batch(64, 128, 20, 8, 1024, 1000, 40, 12345)
tr = Trainer(m, world=1, use_graph=False, force_dp=True)
tr.set_batch(b['video'], b['lens'], b['word_ids'], b['char_ids'], b['y1'], b['y2'], b['match'], b['inner'])
for _ in range(200): tr.step(1e-4, 0.2)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): tr.step(1e-4, 0.2)
torch.cuda.synchronize(); print('force_dp eager ms/step', (time.perf_counter() - t0) / 50 * 1e3)
PY

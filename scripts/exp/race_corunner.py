"""WHICH of our kernels must run beside the matching head for its packed-fp32 op_sel instructions to lose terms?  The victim is the whole forward of a
library built WITH those instructions (build_exp/packed_heads.so: HUAL_BUILD_NO_ISA_CHECK=1 python -m hual_amd.build --out build_exp/packed_heads.so
--file-flags "heads.hip=-Xclang -target-feature -Xclang +packed-fp32-ops"); a second stream runs ONE block of a second model back to back.
   HUAL_LIB_PATH=$PWD/build_exp/packed_heads.so python scripts/exp/race_corunner.py N"""
import sys, os, threading, time, ctypes
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import parity_util as pu
import test_gpu_blocks as TB
from hual_amd import lib
shape = dict(B=16, T=64, L=20, C=8, seed=12345, max_vlen=64, vdim=256)
cfg, p, wv, b, labels = pu.make_case(**shape)
m = pu.hip_model(cfg, p, wv); m.ws_poison = None
dv = [torch.as_tensor(x).cuda() for x in (b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy())]
blk = TB.Block(**shape); blk.m.ws_poison = None
l = blk.l
x = blk.rand(blk.R, 1).to(blk.dev); y = torch.empty_like(x)
feats = torch.empty(blk.R, 128, device=blk.dev)
s_log = torch.empty(blk.B, blk.T, device=blk.dev); e_log = torch.empty_like(s_log)
si = torch.empty(blk.B, dtype=torch.int64, device=blk.dev); ei = torch.empty_like(si)
xv = blk.rand(blk.Nv, 3).to(blk.dev)
co = {
    'nothing': None,
    'whole forward': lambda: blk.m.forward(*dv, drop_rate=0.0),
    'video_proj_ln (feature load + LN)': lambda: lib.check(l.hual_video_proj_ln_fwd(ctypes.byref(blk.m.cfg), lib.ptr(blk.m.params), lib.ptr(blk.m.word_table), ctypes.byref(blk.bt), ctypes.byref(blk.opts), lib.ptr(y), *blk.tail())),
    'conv_block': lambda: lib.check(l.hual_conv_block_fwd(*blk.args(), lib.ptr(x), lib.ptr(y), *blk.tail())),
    'dual_attn layer 0': lambda: lib.check(l.hual_dual_attn_fwd(*blk.args(), 0, lib.ptr(x), lib.ptr(y), *blk.tail())),
    'cq_attn': lambda: lib.check(l.hual_cq_attn_fwd(*blk.args(), lib.ptr(x), lib.ptr(feats), *blk.tail())),
    'predictor': lambda: lib.check(l.hual_predictor_fwd(*blk.args(), lib.ptr(xv), lib.ptr(s_log), lib.ptr(e_log), lib.ptr(si), lib.ptr(ei), *blk.tail())),
}
def one():
    o = m.forward(*dv, drop_rate=0.0)
    torch.cuda.synchronize()
    return [o[k].cpu().numpy().copy() for k in ('start_logits', 'end_logits', 'match_scores')]
ref = one()
n = int(sys.argv[1])
only = sys.argv[2:] 
for name, fn in co.items():
    if only and not any(o in name for o in only): continue
    stop = [False]; err = []
    def load():
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                while not stop[0]:
                    for _ in range(20): fn()
                    s.synchronize()
        except Exception as e:
            err.append(e)
    th = None
    if fn is not None:
        th = threading.Thread(target=load); th.start(); time.sleep(0.3)
    bad = sum(any(not np.array_equal(a, c) for a, c in zip(ref, one())) for _ in range(n))
    stop[0] = True
    if th: th.join()
    print('second stream runs %-36s: %3d of %d forwards differ%s' % (name, bad, n, '   (co-runner failed: %s)' % err[0] if err else ''), flush=True)

"""measured errors of the attention entry points against float64 autograd (the cases of tests/test_gpu_kernels.py)"""
import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import test_gpu_kernels as K
from hual_amd import lib
dev = torch.device('cuda:0'); l = lib.load()
for scale in (1.0, 1e-3):
  for (B, Tq, Tk) in [(2, 16, 16), (3, 37, 9), (2, 128, 128), (1, 20, 256), (2, 128, 20), (1, 256, 20), (2, 200, 32), (2, 100, 100), (1, 256, 256), (3, 1, 1)]:
    for rate in (0.0, 0.2):
        Q, Kk, V, qm, km = K._attn_case(dev, B, Tq, Tk, 77 + B * 1000 + Tq + Tk)
        g = torch.Generator().manual_seed(5)
        dO = (torch.randn(B * Tq, 128, generator=g) * scale).to(dev)
        seed, offset, site = 0x1234567800000042, 3, 9
        rng_state = torch.tensor(np.array([seed & 0xffffffff, seed >> 32, offset], dtype=np.uint32).view(np.int32)).to(dev)
        ldm = l.hual_attention_keep_row_bytes(Tk)
        keep = torch.zeros(B * Tq * 8, ldm, dtype=torch.uint8, device=dev)
        stats = torch.zeros(2, B * Tq * 8, device=dev)
        O = torch.empty(B * Tq, 128, device=dev)
        lib.check(l.hual_attention_fwd_save(lib.ptr(Q), 128, lib.ptr(Kk), lib.ptr(V), 128, lib.ptr(O), 128, B, Tq, Tk, lib.ptr(qm), lib.ptr(km),
                                            lib.ptr(stats), lib.ptr(keep), ldm, lib.ptr(rng_state), rate, site, lib.stream_ptr()))
        dQ, dK, dV = torch.full_like(Q, 7.0), torch.full_like(Kk, 7.0), torch.full_like(V, 7.0)
        lib.check(l.hual_attention_bwd(lib.ptr(Q), 128, lib.ptr(Kk), lib.ptr(V), 128, lib.ptr(O), 128, lib.ptr(stats), lib.ptr(keep), ldm,
                                       lib.ptr(dO), 128, lib.ptr(dQ), 128, lib.ptr(dK), lib.ptr(dV), 128, B, Tq, Tk, lib.ptr(qm), lib.ptr(km),
                                       lib.ptr(rng_state), rate, site, lib.stream_ptr()))
        Qd, Kd, Vd = (t.double().cpu().requires_grad_(True) for t in (Q, Kk, V))
        ref = K._attn_ref(Qd, Kd, Vd, qm.double().cpu(), km.double().cpu(), B, Tq, Tk, (seed, offset, rate, site) if rate > 0 else None)
        eo = (O.double().cpu() - ref.detach()).abs().max().item()
        ref.backward(dO.double().cpu())
        # float32 torch reference of the same function for scale
        Qf, Kf, Vf = (t.float().cpu().requires_grad_(True) for t in (Q, Kk, V))
        r32 = K._attn_ref(Qf, Kf, Vf, qm.float().cpu(), km.float().cpu(), B, Tq, Tk, (seed, offset, rate, site) if rate > 0 else None)
        r32.backward(dO.float().cpu())
        e = []
        for got, want, w32 in ((dQ, Qd.grad, Qf.grad), (dK, Kd.grad, Kf.grad), (dV, Vd.grad, Vf.grad)):
            sc = want.abs().max().item()
            e.append(((got.double().cpu() - want).abs().max().item() / max(sc, 1e-30), (w32.double() - want).abs().max().item() / max(sc, 1e-30)))
        print('dO x%-6g B%d Tq%-3d Tk%-3d rate %.1f  O err %.1e (f32 %.1e) | dQ %.1e (f32 %.1e) dK %.1e (%.1e) dV %.1e (%.1e)' %
              (scale, B, Tq, Tk, rate, eo, (r32.detach().double() - ref.detach()).abs().max().item(), e[0][0], e[0][1], e[1][0], e[1][1], e[2][0], e[2][1]))

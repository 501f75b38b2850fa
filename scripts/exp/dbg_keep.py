"""keep words written by hual_attention_fwd_save against oracle/philox.py mask_attn (layout: csrc/attn.h)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np, torch
from hual_amd import lib
from oracle import philox as px
l = lib.load()
dev = torch.device('cuda:0')
for (B, Tq, Tk) in ((2, 16, 16), (2, 37, 40), (1, 128, 128)):
    g = torch.Generator().manual_seed(1)
    Q = torch.randn(B * Tq, 128, generator=g).to(dev); K = torch.randn(B * Tk, 128, generator=g).to(dev); V = torch.randn(B * Tk, 128, generator=g).to(dev)
    qm = torch.ones(B * Tq, device=dev); km = torch.ones(B * Tk, device=dev)
    seed, offset, site, rate = 0x1234567800000042, 3, 9, 0.2
    rs = torch.tensor(np.array([seed & 0xffffffff, seed >> 32, offset], dtype=np.uint32).view(np.int32)).to(dev)
    ldm = l.hual_attention_keep_row_bytes(Tk)
    keep = torch.zeros(B * Tq * 8, ldm, dtype=torch.uint8, device=dev)
    stats = torch.zeros(2, B * Tq * 8, device=dev)
    O = torch.empty(B * Tq, 128, device=dev)
    lib.check(l.hual_attention_fwd_save(lib.ptr(Q), 128, lib.ptr(K), lib.ptr(V), 128, lib.ptr(O), 128, B, Tq, Tk, lib.ptr(qm), lib.ptr(km),
                                        lib.ptr(stats), lib.ptr(keep), ldm, lib.ptr(rs), rate, site, lib.stream_ptr()))
    torch.cuda.synchronize()
    nqt, nkt = (Tq + 15) // 16, (Tk + 15) // 16
    w = keep.cpu().numpy().reshape(-1)[:B * 8 * nqt * nkt * 32].view(np.uint64).reshape(B, 8, nqt, nkt, 4)
    rng = px.DropoutRNG(seed, offset, rate)
    rid = (np.arange(B * Tq).reshape(B, 1, Tq) * 8 + np.arange(8).reshape(1, 8, 1)).reshape(-1)
    m = (rng.mask_attn(site, rid, Tk) > 0).reshape(B, 8, Tq, Tk)
    bad = 0; tot = 0
    for b in range(B):
        for h in range(8):
            for q in range(Tq):
                for k in range(Tk):
                    qt, jq, kt, gg, r = q >> 4, q & 15, k >> 4, (k >> 2) & 3, k & 3
                    bit = (int(w[b, h, qt, kt, r]) >> (16 * gg + jq)) & 1
                    bad += int(bit != int(m[b, h, q, k])); tot += 1
    print('B %d Tq %d Tk %d: %d of %d keep bits differ from the oracle; word[0,0,0,0] = %s' % (B, Tq, Tk, bad, tot, [hex(int(x)) for x in w[0, 0, 0, 0]]))

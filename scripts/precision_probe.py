#!/usr/bin/env python3
"""GPU box: where does the HIP-vs-oracle difference come from?  For one case prints, per tensor, the difference of
  (a) HIP split-bf16 path vs float32 oracle   (b) HIP exact-fp32 path vs float32 oracle
  (c) float32 oracle vs float64 oracle        (d) HIP split-bf16 path vs float64 oracle
all relative to max|ref| (the test metric), worst first.  Usage: precision_probe.py B T L C vdim drop seed"""
import os
import sys

sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import torch  # noqa: E402
import parity_util as pu  # noqa: E402


def run_hip(case, drop, env):
    for k, v in env.items():
        os.environ[k] = v
    rows, idx, o, h, m = pu.compare(*case, drop_rate=drop)
    for k in env:
        del os.environ[k]
    return {(r[0], r[1]): r for r in rows}, m, o


def main():
    a = sys.argv[1:]
    B, T, L, C, vdim = [int(x) for x in a[:5]]
    drop, seed = float(a[5]), int(a[6])
    case = pu.make_case(B=B, T=T, L=L, C=C, seed=seed, max_vlen=max(T, L), vdim=vdim)
    ra, m, o32 = run_hip(case, drop, {})
    rb, _, _ = run_hip(case, drop, {'HUAL_GEMM_BF16': '0', 'HUAL_DW_IMPL': '0'})
    # float64 oracle with the same active sets
    pins = pu.relu_pins(m, B, T, L)
    cfg, p, wv, b, labels = case
    o64, g64 = pu.oracle_run(cfg, p, wv, b, labels, drop, 5, 7, dtype=torch.float64, relu_pin=pins)
    o32b, g32 = pu.oracle_run(cfg, p, wv, b, labels, drop, 5, 7, dtype=torch.float32, relu_pin=pins)
    hg = m.grads_dict()
    print('%-62s %10s %10s %10s %10s' % ('tensor (metric = maxabs diff / max(1, maxabs ref))', 'bf16x3', 'fp32mfma', 'o32-o64', 'bf16x3-o64'))
    out = []
    for k in g32:
        ref = g64[k].double()
        sc = max(1.0, float(ref.abs().max()))
        d_o = float((g32[k].double() - ref).abs().max()) / sc
        d_h = float((torch.from_numpy(hg[k]).double().reshape(ref.shape) - ref).abs().max()) / sc
        r = ra[('grad', k)]
        r2 = rb[('grad', k)]
        out.append((min(r[2], r[2] / max(r[3], 1e-30)), 'grad ' + k, min(r2[2], r2[2] / max(r2[3], 1e-30)), d_o, d_h))
    for k in ('start_logits', 'end_logits', 'match_scores'):
        ref = o64[k].double()
        sc = max(1.0, float(ref.abs().max()))
        r, r2 = ra[('out', k)], rb[('out', k)]
        out.append((min(r[2], r[2] / max(r[3], 1e-30)), 'out ' + k, min(r2[2], r2[2] / max(r2[3], 1e-30)),
                    float((o32b[k].double() - ref).abs().max()) / sc, float('nan')))
    for key, r in ra.items():
        if key[0] in ('tap', 'loss'):
            r2 = rb[key]
            out.append((min(r[2], r[2] / max(r[3], 1e-30)), key[0] + ' ' + key[1], min(r2[2], r2[2] / max(r2[3], 1e-30)), float('nan'), float('nan')))
    out.sort(key=lambda t: -t[0])
    for t in out[:40]:
        print('%-62s %10.2e %10.2e %10.2e %10.2e' % (t[1][:62], t[0], t[2], t[3], t[4]))


if __name__ == '__main__':
    main()

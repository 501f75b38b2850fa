"""Does ANY result depend on what the workspace held before?  NaN poison (0xFF bytes) is invisible to fmaxf - a scale taken from a maximum
over uninitialised entries would pass every poison test and still change the rounding.  Forward (+ backward) with the workspace
pre-filled with 0, 1e30, -1e30, 3e4 and NaN; outputs, taps and gradients compared bit for bit against the zero-filled run.
   python scripts/exp/dbg_ws_garbage.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import parity_util as pu


def run(case, fill, drop, train):
    cfg, p, wv, b, labels = case
    m = pu.hip_model(cfg, p, wv)
    m.ws_poison = None
    B, T = b['video'].shape[:2]
    L, C = b['word_ids'].shape[1], b['char_ids'].shape[2]
    ws = m._workspace(B, T, L, C)
    if fill == 'nan':
        ws.fill_(0xFF)
    else:
        ws.view(torch.float32)[:ws.numel() // 4].fill_(fill)
    m.set_rng(5, 7)
    m.debug_taps = True
    o = m.forward(b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy(), drop_rate=drop,
                  labels=tuple(x.numpy() for x in labels) if train else None)
    out = {k: v.detach().cpu().numpy().copy() for k, v in o.items()}
    if train:
        m.backward()
        torch.cuda.synchronize()
        out['grads'] = m.grads.detach().cpu().numpy().copy()
    taps = {}
    for name, (off, rows, cols) in m._ws_table.items():
        if rows * cols > 0 and not name.startswith(('params.', 'dw.table')) and '.rb' not in name and '.kb' not in name and 'keep' not in name:
            try:
                taps[name] = m.tap(name).detach().cpu().numpy().copy()
            except Exception:
                pass
    return out, taps


for shape in (dict(B=6, T=24, L=9, C=8, seed=8, max_vlen=24, vdim=64), dict(B=3, T=100, L=45, C=13, seed=92, max_vlen=100, vdim=64)):
    case = pu.make_case(**shape)
    for drop, train in ((0.0, False), (0.5, False), (0.2, True)):
        ref, rtaps = run(case, 0.0, drop, train)
        for fill in (1e30, -1e30, 3e4, 'nan'):
            got, gtaps = run(case, fill, drop, train)
            bad = [k for k in ref if not np.array_equal(ref[k], got[k], equal_nan=True) and k != 'grads']
            gd = float(np.abs(ref['grads'] - got['grads']).max() / np.abs(ref['grads']).max()) if train else 0.0
            print('%s drop %.1f train %s fill %-6s outputs differing: %s   grads max rel diff %.2e' % (shape['T'], drop, train, fill, bad, gd), flush=True)
            if bad:
                for k in bad:
                    print('    %s max abs diff %.3e' % (k, float(np.nanmax(np.abs(ref[k].astype(np.float64) - got[k].astype(np.float64))))))

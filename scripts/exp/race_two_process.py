"""which tensor differs first when the forward is perturbed by a second process on the GPU?   python dbg_fwd_race2.py detect|load"""
import sys, os, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import parity_util as pu
mode = sys.argv[1]
case = pu.make_case(B=16, T=64, L=20, C=8, seed=12345, max_vlen=64, vdim=256)
cfg, p, wv, b, labels = case
m = pu.hip_model(cfg, p, wv)
m.ws_poison = None
m.debug_taps = True
args = (b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy())


def snap():
    m.set_rng(3, 11)
    o = m.forward(*args, drop_rate=0.0)
    torch.cuda.synchronize()
    t = {}
    for name, (off, rows, cols) in m._ws_table.items():
        if rows * cols > 0 and not name.startswith(('params.', 'dw.table')):
            try:
                t[name] = (off, m._ws[off:off + rows * cols * 4].cpu().numpy().copy())
            except Exception:
                pass
    t['out.start_logits'] = (1 << 60, o['start_logits'].cpu().numpy().view(np.uint8).ravel().copy())
    t['out.match_scores'] = ((1 << 60) + 1, o['match_scores'].cpu().numpy().view(np.uint8).ravel().copy())
    return t


if mode == 'load':
    t0 = time.time()
    while time.time() - t0 < float(sys.argv[2]):
        m.forward(*args, drop_rate=0.0)
    torch.cuda.synchronize()
    sys.exit(0)
ref = snap()
found = 0
for it in range(int(sys.argv[2])):
    cur = snap()
    diff = [(off, name, int((cur[name][1] != ref[name][1]).sum())) for name, (off, a) in ref.items() if not np.array_equal(a, cur[name][1])]
    if diff:
        found += 1
        diff.sort()
        print('iteration %d: %d tensors differ; in workspace order (= dataflow order of the bump allocator): %s' % (it, len(diff), [(n, c) for _, n, c in diff[:8]]), flush=True)
        real = [d for d in diff if not d[1].startswith(('cq.sr', 'cq.sc'))]
        if real:
            off, name, cnt = real[0]
            a = ref[name][1].view(np.float32); c = cur[name][1].view(np.float32)
            rows, cols = m._ws_table[name][1], m._ws_table[name][2]
            a = a.reshape(rows, cols); c = c.reshape(rows, cols)
            rr, cc = np.nonzero(a != c)
            if name == 'outputs':
                f = ref['fuse'][1].view(np.float32).reshape(rows, cols)
                r0 = int(rr[0]); cs = sorted(set(cc[rr == r0].tolist()))
                print('   row %d differing cols %s' % (r0, cs[:40]))
                print('   ref  ', np.round(a[r0, cs[:8]], 4).tolist())
                print('   cur  ', np.round(c[r0, cs[:8]], 4).tolist())
                print('   fuse ', np.round(f[r0, cs[:8]], 4).tolist())
                print('   cur - ref', np.round((c - a)[r0, cs[:8]], 4).tolist(), ' ref - fuse', np.round((a - f)[r0, cs[:8]], 4).tolist(), ' cur - fuse', np.round((c - f)[r0, cs[:8]], 4).tolist())
                msr = ref['out.match_scores'][1].view(np.float32).reshape(-1, 4); msc = cur['out.match_scores'][1].view(np.float32).reshape(-1, 4)
                E = m.state_dict()['label_emb']
                mk = (np.arange(64)[None, :] < b['lens'].numpy()[:, None]).reshape(-1).astype(np.float32)
                print('   match_scores row: ref', msr[r0].tolist(), 'cur', msc[r0].tolist(), ' rows with differing scores:', int((msr != msc).any(1).sum()))
                exp_ref = (f[r0] + msr[r0] @ E) * mk[r0]; exp_cur = (f[r0] + msc[r0] @ E) * mk[r0]
                print('   |ref - expected(ref scores)| %.2e   |cur - expected(cur scores)| %.2e   |cur - expected(ref scores)| %.2e' % (
                    float(np.abs(a[r0] - exp_ref).max()), float(np.abs(c[r0] - exp_cur).max()), float(np.abs(c[r0] - exp_ref).max())))
                for other in range(rows):
                    if other != r0 and np.array_equal(c[r0], a[other]):
                        print('   cur row equals ref row', other)
            print('   first real difference: %s [%d x %d]: rows %s cols %d..%d  max abs %.3e (values up to %.3e)  B=16 T=64 L=20: Nv=%d'
                  % (name, rows, cols, sorted(set(rr.tolist()))[:24], cc.min(), cc.max(), float(np.abs(a - c).max()), float(np.abs(a).max()), 16 * 64), flush=True)
        if found >= 40:
            break
print('detect done, %d differing runs' % found)
print('runs with a real difference:', sum(1 for _ in []))

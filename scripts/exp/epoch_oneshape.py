#!/usr/bin/env python3
"""Where does the epoch loop's distance to the resident-batch rate come from?  The same loop (Trainer.run_epoch, one graph launch per
step) over sets with (a) ONE padded shape for every batch, (b) two alternating shapes, (c) the ActivityNet lengths - next to the
resident-batch replay of shape (a).   python scripts/exp/epoch_oneshape.py [--bs 16]"""
import argparse, os, sys, time
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))


def make_set(N, vdim, shapes, seed, dev):
    """N samples; sample i has the lengths shapes[(i // bs_block) % len(shapes)] - with an unshuffled order every batch is one shape"""
    from hual_amd import al
    from hual_amd.dataset import DeviceDataset
    g = np.random.default_rng(seed)
    recs, vlens, gt = [], {}, []
    for i in range(N):
        T, L, C = shapes[i % len(shapes)]
        name = 'v%d' % i
        vlens[name] = T
        w = [int(x) for x in g.integers(2, 1000, size=L)]
        recs.append(dict(vid=name, duration=60.0, v_len=T, words=['w'] * L, w_ids=w, c_ids=[[int(x) for x in g.integers(1, 40, size=C)] for _ in range(L)]))
        s = float(g.uniform(0, 40)); gt.append([name, 60.0, [s, s + 10.0], 'x'])
    total = sum(vlens[v] for v in sorted(vlens))
    bank = torch.randn(total, vdim, device=dev, generator=torch.Generator(device=dev).manual_seed(seed))
    ds = DeviceDataset(recs, vlens, device=dev, feat_bank=bank)
    s0, e0 = al.labels_from_times(gt, ds.vlen_h)
    ds.set_labels(s0, e0)
    return ds


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--bs', type=int, default=16)
    ap.add_argument('--steps', type=int, default=512)
    ap.add_argument('--only-one', action='store_true', help='the one-shape set only, no resident timing (profiling runs)')
    args = ap.parse_args()
    from hual_amd import lib
    from hual_amd.model import SeqPAN
    from hual_amd.train import Trainer
    sys.path.insert(0, ROOT)
    import bench
    dev = torch.device('cuda:0')
    bs, N = args.bs, args.bs * args.steps
    cfg = lib.make_cfg(vdim=1024, max_vlen=100, num_words=1000, num_chars=40)
    wv = np.random.default_rng(777).normal(0, 0.4, size=(998, 300)).astype(np.float32)
    res = bench._resident_ms(dev, cfg, wv, bs, 100, 30, 11, 1024, 0.2) if not args.only_one else 1.0
    print('resident batch B%d T100 L30 C11: %.4f ms/step' % (bs, res), flush=True)
    sets = (('one shape', [(100, 30, 11)], 1), ('two alternating shapes', [(100, 30, 11), (100, 24, 9)], bs),
            ('eight alternating shapes', [(100, 30 - k, 11) for k in range(8)], bs),
                                ('64 alternating shapes', [(100, 30 - (k % 8), 8 + k // 8) for k in range(64)], bs),
                                ('256 alternating shapes', [(100, 32 - (k % 16), 4 + k // 16) for k in range(256)], bs))
    for name, shapes, block in (sets[:1] if args.only_one else sets):
        # samples are laid out so that consecutive blocks of `bs` samples share a shape: sample i takes shapes[(i // bs) % n]
        sh = [shapes[(i // bs) % len(shapes)] for i in range(N)]
        ds = make_set(N, 1024, [sh[i] for i in range(N)], 3, dev) if False else None
        from hual_amd import al
        # (make_set indexes shapes by i % len: pass the expanded list)
        ds = make_set(N, 1024, sh, 3, dev)
        model = SeqPAN(cfg, wv, device=dev)
        tr = Trainer(model, world=1, use_graph=True)
        order = np.arange(N)                     # unshuffled: batch k = samples [k bs, (k + 1) bs) = one shape
        for ep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            tr.run_epoch(ds, order, bs, lr=1e-4, drop_rate=0.2, min_chars=4)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print('%-26s: %.4f ms/step (third epoch; %s) = %.3f of the resident rate' % (name, dt / args.steps * 1e3, dict(tr.stats), res / (dt / args.steps * 1e3)), flush=True)
        del tr, model, ds
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()

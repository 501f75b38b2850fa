#!/usr/bin/env python3
"""The device-fed epoch loop (Trainer.run_epoch) on a training set whose padded shapes follow the reference's OWN annotations
(tests/golden/lengths_*.npz): ms/step per epoch, launch modes, distinct padded shapes - for several graph-cache policies.

    python scripts/exp/epoch_real.py --task anet --bs 16 --max-vlen 100 --samples 8192 --epochs 5 --modes default,eager
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def build_set(task, N, vdim, max_vlen, seed, dev):
    import al_synth
    from hual_amd import al
    from hual_amd.dataset import DeviceDataset
    recs, vlens, data_gt, _ = al_synth.make_trainset_from_lengths(task, N, vdim, max_vlen, seed, feats=False)
    total = sum(vlens[v] for v in sorted(vlens))
    bank = torch.randn(total, vdim, device=dev, generator=torch.Generator(device=dev).manual_seed(seed))
    ds = DeviceDataset(recs, vlens, device=dev, feat_bank=bank)
    s0, e0 = al.labels_from_times(data_gt, ds.vlen_h)
    ds.set_labels(s0, e0)
    return ds


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--task', default='anet')
    ap.add_argument('--bs', type=int, default=16)
    ap.add_argument('--max-vlen', type=int, default=100)
    ap.add_argument('--vdim', type=int, default=1024)
    ap.add_argument('--samples', type=int, default=8192)
    ap.add_argument('--epochs', type=int, default=5)
    ap.add_argument('--drop', type=float, default=0.2)
    ap.add_argument('--modes', default='default,eager')
    args = ap.parse_args()
    from hual_amd import lib
    from hual_amd.model import SeqPAN
    from hual_amd.train import Trainer
    dev = torch.device('cuda:0')
    t0 = time.perf_counter()
    ds = build_set(args.task, args.samples, args.vdim, args.max_vlen, 11, dev)
    N = len(ds)
    print('set: %s %d samples, max shape %s, built in %.1f s' % (args.task, N, ds.max_shape(4), time.perf_counter() - t0), flush=True)
    Tm, Lm, Cm = ds.max_shape(4)
    cfg = lib.make_cfg(vdim=args.vdim, max_vlen=max(Tm, Lm, args.max_vlen), num_words=1000, num_chars=40)
    wv = np.random.default_rng(777).normal(0, 0.4, size=(998, 300)).astype(np.float32)
    for mode in args.modes.split(','):
        model = SeqPAN(cfg, wv, device=dev, seed=12345, rng_seed=12345)
        tr = Trainer(model, world=1, use_graph=(mode != 'eager'))
        if mode.startswith('limit'):
            tr.cache_limit = int(mode[5:])
        if mode.startswith('after'):
            tr.capture_after = int(mode[5:])
        g = np.random.default_rng(0)
        shapes = set()
        for ep in range(args.epochs):
            order = g.permutation(N)
            before = dict(tr.stats)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tr.run_epoch(ds, order, args.bs, lr=1e-4, drop_rate=args.drop, min_chars=4)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ns = (N + args.bs - 1) // args.bs
            Ts = []
            for lo in range(0, N, args.bs):
                shp = ds.batch_shape(order[lo:lo + args.bs])
                shapes.add((len(order[lo:lo + args.bs]),) + (shp[0], shp[1], max(4, shp[2])))
                Ts.append(shp[0])
            d = {k: round(tr.stats[k] - before.get(k, 0), 4) for k in tr.stats}
            print('%-10s epoch %d: %.4f ms/step (%d steps, %.0f clips/s) mean T %.1f  distinct shapes so far %d  %s loss %.4f'
                  % (mode, ep, dt / ns * 1e3, ns, N / dt, float(np.mean(Ts)), len(shapes), d, float(tr.last_loss())), flush=True)
        del tr, model
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()

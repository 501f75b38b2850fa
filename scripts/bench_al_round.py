#!/usr/bin/env python3
"""Times one active-learning round (update_label -> train -> infer_trainset) on a synthetic, HBM-resident training set
and the CPU oracle's update_label on the same inputs (BASELINE.json configs[4], scaled to one GPU).
    python scripts/bench_al_round.py [--n 4096] [--epochs 1] [--task anet] [--cpu-baseline]
Prints one JSON line."""
import argparse
import copy
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=4096)
    ap.add_argument('--videos', type=int, default=1024)
    ap.add_argument('--vdim', type=int, default=1024)
    ap.add_argument('--max-vlen', type=int, default=100)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--epochs', type=int, default=1)
    ap.add_argument('--task', default='anet')
    ap.add_argument('--cpu-baseline', action='store_true',
                    help='also time the CPU oracle (oracle/al_ref.py) on the same update_label inputs and compare the results - the\n                    checker / baseline leg, off by default: the measured path never touches oracle/')
    args = ap.parse_args()
    import al_synth
    from hual_amd import al, lib
    from hual_amd.dataset import DeviceDataset
    from hual_amd.model import SeqPAN
    recs, vis, data_gt, data_old = al_synth.make_trainset(args.n, args.videos, args.vdim, args.max_vlen, seed=11, num_words=1000,
                                                          num_chars=40, max_words=20)
    cfg = lib.make_cfg(vdim=args.vdim, max_vlen=args.max_vlen, num_words=1000, num_chars=40)
    wv = np.random.default_rng(777).normal(0, 0.4, size=(998, 300)).astype(np.float32)
    model = SeqPAN(cfg, wv)
    t0 = time.perf_counter()
    ds = DeviceDataset(recs, vis)
    s0, e0 = al.labels_from_times(data_old, ds.vlen_h)
    ds.set_labels(s0, e0)
    for r, a, b in zip(recs, s0, e0):
        r['s_ind'], r['e_ind'] = int(a), int(b)
    t_upload = time.perf_counter() - t0

    def batches():
        for lo in range(0, args.n, args.batch):
            sel = np.arange(lo, min(args.n, lo + args.batch))
            f = ds.assemble(sel, labels=False, min_chars=4)
            yield [recs[i] for i in sel], f['video'], f['video_seq_len'], f['word_ids'], f['char_ids']
    prop0, _ = al.infer_trainset(model, batches(), mc_dropout=0.5)
    out = dict(n_samples=args.n, max_vlen=args.max_vlen, vdim=args.vdim, batch=args.batch, epochs=args.epochs, task=args.task,
               dataset_upload_s=round(t_upload, 3), feature_bank_gb=round(ds.feat_bank.numel() * 4 / 1e9, 3))
    if args.cpu_baseline:
        from oracle import al_ref as A
        t0 = time.perf_counter()
        ref = A.update_labels(copy.deepcopy(data_old), data_gt, prop0, A.get_coff(args.task, 1))
        out['oracle_update_label_s'] = round(time.perf_counter() - t0, 3)
    # warm (first-use module load), then the timed round
    al.update_labels(copy.deepcopy(data_old), data_gt, prop0, al.get_coff(args.task, 1))
    new_data, prop1, m = al.run_round(model, ds, copy.deepcopy(data_old), data_gt, prop0, args.task, 1, epochs=args.epochs,
                                      batch_size=args.batch, lr=1e-4, drop_rate=0.2, log=lambda s: print(s, file=sys.stderr))
    if args.cpu_baseline:
        out['update_label_equal_to_oracle'] = all(a[2] == b[2] and a[4] == b[4] for a, b in zip(new_data, ref))
    out.update({k: (round(v, 4) if isinstance(v, float) else v) for k, v in m.items()})
    print(json.dumps(out))


if __name__ == '__main__':
    main()

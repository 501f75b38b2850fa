#!/usr/bin/env python3
"""Times one active-learning round (update_label -> train -> infer_trainset) on a synthetic, HBM-resident training set
and the CPU oracle's update_label on the same inputs (BASELINE.json configs[4], scaled to one GPU).
    python scripts/bench_al_round.py [--n 4096] [--epochs 1] [--task anet] [--lengths anet] [--cpu-baseline]
Data parallel (BASELINE.json configs[4] as written: one process per GPU; --batch is then the batch PER GPU):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P scripts/bench_al_round.py ...
--lengths anet|charades draws (v_len, words, longest word) from the reference's own annotations (tests/golden/lengths_*.npz).
Prints one JSON line (rank 0)."""
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
    ap.add_argument('--lengths', default='synthetic', choices=['synthetic', 'anet', 'charades'])
    ap.add_argument('--cpu-baseline', action='store_true',
                    help='also time the CPU oracle (oracle/al_ref.py) on the same update_label inputs and compare the results - the\n                    checker / baseline leg, off by default: the measured path never touches oracle/')
    args = ap.parse_args()
    import torch
    world, rank, local = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        torch.distributed.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    import al_synth
    from hual_amd import al, lib
    from hual_amd import dist as hdist
    from hual_amd.dataset import DeviceDataset
    from hual_amd.model import SeqPAN
    if args.lengths == 'synthetic':
        recs, vis, data_gt, data_old = al_synth.make_trainset(args.n, args.videos, args.vdim, args.max_vlen, seed=11, num_words=1000,
                                                              num_chars=40, max_words=20)
        bank = None
    else:
        recs, vis, data_gt, data_old = al_synth.make_trainset_from_lengths(args.lengths, args.n, args.vdim, args.max_vlen, 11, feats=False)
        bank = torch.randn(sum(vis[v] for v in vis), args.vdim, device=dev, generator=torch.Generator(device=dev).manual_seed(11))
    Lm = max(len(r['w_ids']) for r in recs)
    cfg = lib.make_cfg(vdim=args.vdim, max_vlen=max(args.max_vlen, Lm), num_words=1000, num_chars=40)
    wv = np.random.default_rng(777).normal(0, 0.4, size=(998, 300)).astype(np.float32)
    model = SeqPAN(cfg, wv, device=dev, rng_seed=12345 + 1000003 * rank)
    t0 = time.perf_counter()
    ds = DeviceDataset(recs, vis, device=dev, feat_bank=bank)
    s0, e0 = al.labels_from_times(data_old, ds.vlen_h)
    ds.set_labels(s0, e0)
    for r, a, b in zip(recs, s0, e0):
        r['s_ind'], r['e_ind'] = int(a), int(b)
    t_upload = time.perf_counter() - t0

    prop0, _ = al.infer_trainset_sharded(model, ds, args.batch, mc_dropout=0.5)      # (records on rank 0)
    out = dict(n_samples=len(ds), n_gpus=world, lengths_from=args.lengths, max_vlen=args.max_vlen, vdim=args.vdim, batch=args.batch, epochs=args.epochs, task=args.task,
               dataset_upload_s=round(t_upload, 3), feature_bank_gb=round(ds.feat_bank.numel() * 4 / 1e9, 3))
    if args.cpu_baseline and rank == 0:
        from oracle import al_ref as A
        t0 = time.perf_counter()
        ref = A.update_labels(copy.deepcopy(data_old), data_gt, prop0, A.get_coff(args.task, 1))
        out['oracle_update_label_s'] = round(time.perf_counter() - t0, 3)
    # warm (first-use module load), then the timed round
    if rank == 0:
        al.update_labels(copy.deepcopy(data_old), data_gt, prop0, al.get_coff(args.task, 1))
    new_data, prop1, m = al.run_round(model, ds, copy.deepcopy(data_old), data_gt, prop0, args.task, 1, epochs=args.epochs,
                                      batch_size=args.batch, lr=1e-4, drop_rate=0.2, log=lambda s: print(s, file=sys.stderr))
    if args.cpu_baseline and rank == 0:
        out['update_label_equal_to_oracle'] = all(a[2] == b[2] and a[4] == b[4] for a, b in zip(new_data, ref))
    out.update({k: (round(v, 4) if isinstance(v, float) else v) for k, v in m.items()})
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()

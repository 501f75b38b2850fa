"""Thin driver with the loop contract of /root/reference/main.py:50-111 and utils/runner_utils.py:139-176 on the HIP path.

    reference                                               here
    ----------------------------------------------------    -------------------------------------------------------
    main.py --mode train   (epochs, lr decay :61, best       Runner.train(epochs) -> per-epoch "TRAIN:\\t..\\nTEST:\\t.."
      R1@0.7 checkpoint :71-75)                               lines, best checkpoint = <ckpt_dir>/best_SeqPAN.npz
    main.py --mode test                                      Runner.test()
    main.py --mode infer_trainset -> results/<task>/<suffix>.pkl   Runner.infer_trainset(path)
    runner_utils.train_epoch / test_epoch (IoU bookkeeping)  Runner.train_epoch / Runner.test_epoch
    tf.train.Saver                                           numpy .npz keyed by the TF variable names

Data: the records the reference's dataset_gen leaves (utils/data_gen.py:98-125: vid, duration, v_len, words, w_ids, c_ids,
s_ind, e_ind) + {vid: float32 [n, vdim]} features; both go to HBM once (hual_amd/dataset.py).  Python `random` is seeded here
(the reference leaves the epoch order unseeded).  There is no CPU fallback.
"""
import logging
import os
import pickle
import random
import time

import numpy as np
import torch

from . import al, data, lib
from . import dist as hdist
from .dataset import DeviceDataset
from .model import SeqPAN
from .train import Trainer


class Runner:
    def __init__(self, configs, word_vectors, train_records, test_records, visual_feats, device='cuda:0', seed=12345,
                 ckpt_dir=None, logger=None, feed='device'):
        """configs: dict with the keys of configs/<task>/SeqPAN.yaml (+ num_chars, num_words like main.py:35-36).
        feed='host': the training features stay in host memory and every batch is padded on the host and uploaded through the pinned
        pipeline of hual_amd/feeder.py (the reference's own data path, for feature sets beyond the HBM); default: the training set
        lives in HBM and batches are assembled there."""
        assert feed in ('device', 'host')
        self.configs = configs
        self.feed = feed
        self._host_train = (train_records, visual_feats) if feed == 'host' else None
        self._feeder = None
        # (same parameter seed on every rank of a data-parallel job, its own dropout stream per rank)
        self.model = SeqPAN(configs, word_vectors, device=device, seed=seed, rng_seed=seed + 1000003 * hdist.rank())
        self.train_set = DeviceDataset(train_records, visual_feats, device=device) if feed == 'device' else None
        self.test_set = DeviceDataset(test_records, visual_feats, device=device) if test_records else None
        self.batch_size = int(configs['train']['batch_size'])
        self.droprate = float(configs['train']['droprate'])
        self.lr = float(configs['train']['lr'])
        self.ckpt_dir = ckpt_dir or os.path.join('ckpt', '%s_' % configs.get('task', 'task'))      # main.py:42 (sic)
        self.log = logger or logging.getLogger('hual_amd')
        self.rand = random.Random(seed)
        # data parallel when torch.distributed is initialised (one process per GPU, every rank constructs the same Runner with the
        # same seed): configs.train.batch_size is then the batch PER RANK, the global batch is world times that
        self.world, self.rank = hdist.world_size(), hdist.rank()
        self.trainer = Trainer(self.model, world=self.world, use_graph=True)     # per-shape step graphs (Trainer.set_batch_device)
        self.clips_per_s = 0.0

    # ------------------------------------------------------------------ runner_utils.train_epoch (:139-159)
    @staticmethod
    def _ious(records, sidx, eidx):
        return al.ious_of_spans(records, sidx, eidx)

    def _train_epoch_host(self, cur_lr):
        """runner_utils.train_epoch with the reference's own data path: process_batch on the host (hual_amd/data.py), one upload per
        step behind the previous step (hual_amd/feeder.py)"""
        from .feeder import HostFeeder
        recs, feats = self._host_train
        if self.world > 1:
            raise lib.HualError('feed=\'host\' is single-process: the data-parallel loop shards the device-resident set (Trainer.run_epoch)')
        N, bs = len(recs), self.batch_size
        if self._feeder is None:
            T = max(int(r['v_len']) for r in recs)
            L = max(len(r['w_ids']) for r in recs)
            C = max(4, max(len(w) for r in recs for w in r['c_ids']))
            self._feeder = HostFeeder(self.trainer, capacity=(min(bs, N), T, L, C), vdim=int(next(iter(feats.values())).shape[1]))
        order = list(range(N))
        self.rand.shuffle(order)                                   # data_loader.py:24

        t0 = time.perf_counter()
        # every batch is padded straight into a pinned slot (HostFeeder.feed_records: process_batch without the intermediate arrays; words
        # padded to >= 4 characters - the char CNN's widest filter, modules.py:19-38) while the device runs the previous step
        for lo in range(0, N, bs):
            self._feeder.feed_records([recs[i] for i in order[lo:lo + bs]], feats, cur_lr, self.droprate, min_chars=4)
        spans = self._feeder.collect()
        self.clips_per_s = N / max(time.perf_counter() - t0, 1e-9)
        st = np.concatenate([s for s, _ in spans])
        en = np.concatenate([e for _, e in spans])
        if (st < 0).any() or not np.isfinite(float(self.trainer.last_loss())):
            raise lib.HualError('training diverged out of the split-fp16 operand range: %d clip(s) of this epoch came back with span -1 '
                                '(loss %s).  Lower the learning rate or clip_norm; the last good checkpoint is %s'
                                % (int((st < 0).sum()), float(self.trainer.last_loss()), os.path.join(self.ckpt_dir, 'best_SeqPAN.npz')))
        return al.iou_metrics(self._ious([recs[i] for i in order], st, en))

    def train_epoch(self, cur_lr):
        if self.feed == 'host':
            return self._train_epoch_host(cur_lr)
        ds, N = self.train_set, len(self.train_set)
        order = list(range(N))
        self.rand.shuffle(order)                                   # data_loader.py:24
        t0 = time.perf_counter()
        # the whole epoch is enqueued without a host wait; the spans come back in one transfer (Trainer.run_epoch)
        st, en = self.trainer.run_epoch(ds, order, self.batch_size, lr=cur_lr, drop_rate=self.droprate, min_chars=4)
        self.clips_per_s = N / max(time.perf_counter() - t0, 1e-9)
        # spans of -1 are the kernels' overflow marker (include/hual_seqpan.h: a weight beyond the scaled fp16 image's range, |w| >= 63, or
        # an activation beyond the fp16 operand range, |x| >= 4094, turned a clip's logits into NaN): the float32 reference would still
        # be finite there, so the run stops and says so instead of training on
        if (np.asarray(st) < 0).any() or not np.isfinite(float(self.trainer.last_loss())):
            bad = int((np.asarray(st) < 0).sum())
            raise lib.HualError('training diverged out of the split-fp16 operand range: %d clip(s) of this epoch came back with span -1 '
                                '(loss %s).  Lower the learning rate or clip_norm; the last good checkpoint is %s'
                                % (bad, float(self.trainer.last_loss()), os.path.join(self.ckpt_dir, 'best_SeqPAN.npz')))
        ious = self._ious([ds.records[i] for i in self.trainer.last_epoch_ids], st, en)       # (= order unless world > 1 dropped a tail)
        return al.iou_metrics(ious)

    # ------------------------------------------------------------------ runner_utils.test_epoch (:161-176)
    def test_epoch(self, dataset=None):
        ds = dataset or self.test_set
        ious = []
        for k, lo in enumerate(range(0, len(ds), self.batch_size)):
            if k % self.world != self.rank:                        # batches dealt round-robin to the ranks, IoUs gathered below
                continue
            sel = np.arange(lo, min(len(ds), lo + self.batch_size))
            f = ds.assemble(sel, labels=False, min_chars=4)
            o = self.model.forward(f['video'], f['video_seq_len'], f['word_ids'], f['char_ids'], drop_rate=0.0)
            ious += self._ious([ds.records[i] for i in sel], o['start_index'].cpu().numpy(), o['end_index'].cpu().numpy())
        if self.world > 1:
            parts = hdist.gather_objects(ious)
            ious = hdist.broadcast_object([x for p in parts for x in p] if self.rank == 0 else None)
        return al.iou_metrics(ious)

    # ------------------------------------------------------------------ main.py --mode train (:50-78)
    def train(self, epochs=None):
        epochs = int(epochs if epochs is not None else self.configs['train']['epochs'])
        best, best_lines = -1.0, None
        for epoch in range(epochs):
            self.log.info('Epoch {}|{}:'.format(epoch, epochs))
            cur_lr = self.lr * (1.0 - epoch / epochs)             # main.py:61
            r = self.train_epoch(cur_lr)
            train_line = 'TRAIN:\t{:.2f}\t{:.2f}\t{:.2f}\t{:.2f}\t'.format(*r)
            self.log.info(train_line + '({:.0f} clips/s)'.format(self.clips_per_s))
            test_line = ''
            r1i7 = r[2]
            if self.test_set is not None:
                t = self.test_epoch()
                test_line = 'TEST:\t{:.2f}\t{:.2f}\t{:.2f}\t{:.2f}\t'.format(*t)
                self.log.info(test_line)
                r1i7 = t[2]
            if r1i7 > best:                                       # main.py:71-75
                best = r1i7
                if self.rank == 0:
                    self.save(os.path.join(self.ckpt_dir, 'best_SeqPAN.npz'))
                best_lines = '\n' + train_line + '\n' + test_line
        self.log.info('\n\nHighest R1i7 epoch\n')
        self.log.info(best_lines)
        return best

    def test(self):
        hdist.barrier()                                            # rank 0 wrote the checkpoint
        self.load(os.path.join(self.ckpt_dir, 'best_SeqPAN.npz'))
        t = self.test_epoch()
        self.log.info('TEST:\t{:.2f}\t{:.2f}\t{:.2f}\t{:.2f}\t'.format(*t))
        return t

    # ------------------------------------------------------------------ main.py --mode infer_trainset (:99-111)
    def infer_trainset(self, path=None, mc_dropout=None, load_best=True):
        """results/<task>/<suffix>.pkl of runner_utils.py:103-104.  mc_dropout=None: as the reference runs."""
        if load_best:
            hdist.barrier()
            self.load(os.path.join(self.ckpt_dir, 'best_SeqPAN.npz'))
        if self.train_set is None:          # feed='host': the inference pass works on a device-resident set, built on first use
            self.train_set = DeviceDataset(*self._host_train, device=self.model.device)
        records, ious = al.infer_trainset_sharded(self.model, self.train_set, self.batch_size, mc_dropout=mc_dropout, min_chars=4)
        if self.rank != 0:
            return None, hdist.broadcast_object(None)
        if path:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            with open(path, 'wb') as f:
                pickle.dump(records, f)
        m = hdist.broadcast_object(al.iou_metrics(ious))
        self.log.info('predict train set:\t{:.2f}\t{:.2f}\t{:.2f}\t{:.2f}\t'.format(*m))
        return records, m

    # ------------------------------------------------------------------ checkpoints by TF variable name
    def save(self, path):
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        np.savez(path, **{k.replace('/', '|'): v for k, v in self.model.state_dict().items()})

    def load(self, path):
        with np.load(path) as z:
            self.model.load_state_dict({k.replace('|', '/'): z[k] for k in z.files})

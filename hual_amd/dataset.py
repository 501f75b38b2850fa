"""Training set resident in HBM + device-side batch assembly (csrc/assemble.hip through hual_assemble_batch).

Replaces the per-step numpy work of the reference's loaders (TrainLoader.process_batch / TestLoader.process_batch,
/root/reference/utils/data_loader.py:30-98,145-164) and the feed_dict upload of runner_utils.py:145-147: the features
of every video (Charades: 3.2 GB fp32, ActivityNet at max_vlen 100: ~6 GB - small against 288 GB) are uploaded once,
each step only sends the B sample ids.  Pseudo labels (s_ind / e_ind) are re-uploaded once per active-learning round.
"""
import ctypes

import numpy as np
import torch

from . import lib


class DeviceDataset:
    def __init__(self, records, visual_feats, device='cuda:0', feat_bank=None):
        """records: dicts with vid, w_ids, c_ids (and s_ind, e_ind for training sets), as produced by the reference's
        dataset_gen (utils/data_gen.py:98-125); visual_feats: {vid: float32 [n_clips, vdim]} (data_utils.py:56-67).
        feat_bank: the features already on the device as ONE float32 [total_clips, vdim] tensor, videos in sorted-vid order
        (synthetic sets generated on the GPU); visual_feats then maps vid -> n_clips."""
        if not torch.cuda.is_available():
            raise lib.HualError('DeviceDataset needs a GPU: the HIP path has no CPU fallback')
        self._lib = lib.load()
        self.dev = torch.device(device)
        self.records = records
        vids = sorted({r['vid'] for r in records})
        vid_id = {v: i for i, v in enumerate(vids)}
        if feat_bank is not None:
            nclips = np.array([int(visual_feats[v]) for v in vids], dtype=np.int64)
            self.vdim = int(feat_bank.shape[1])
        else:
            nclips = np.array([visual_feats[v].shape[0] for v in vids], dtype=np.int64)
            self.vdim = int(visual_feats[vids[0]].shape[1])
        feat_off = np.zeros(len(vids) + 1, dtype=np.int64)
        feat_off[1:] = np.cumsum(nclips)
        if feat_bank is not None:
            if feat_bank.dtype != torch.float32 or tuple(feat_bank.shape) != (int(feat_off[-1]), self.vdim) or not feat_bank.is_contiguous():
                raise ValueError('feat_bank must be a contiguous float32 [%d, vdim] tensor' % int(feat_off[-1]))
            bank = feat_bank.to(self.dev)
        else:
            bank = torch.empty(int(feat_off[-1]), self.vdim, dtype=torch.float32, device=self.dev)
            for i, v in enumerate(vids):
                bank[int(feat_off[i]):int(feat_off[i + 1])] = torch.from_numpy(np.ascontiguousarray(visual_feats[v], dtype=np.float32))
        self.feat_bank = bank
        self.sample_vid_h = np.array([vid_id[r['vid']] for r in records], dtype=np.int32)
        self.vlen_h = nclips[self.sample_vid_h].astype(np.int32)
        self.nwords_h = np.array([len(r['w_ids']) for r in records], dtype=np.int32)
        word_off = np.zeros(len(records) + 1, dtype=np.int32)
        word_off[1:] = np.cumsum(self.nwords_h)
        word_bank = np.array([w for r in records for w in r['w_ids']] + [0], dtype=np.int32)
        nchars = np.array([len(c) for r in records for c in r['c_ids']], dtype=np.int32)
        char_off = np.zeros(len(nchars) + 1, dtype=np.int32)
        char_off[1:] = np.cumsum(nchars)
        char_bank = np.array([x for r in records for c in r['c_ids'] for x in c] + [0], dtype=np.int32)
        # longest word of each sample: the batch's C is the maximum over its samples (pad_char_seq, data_utils.py:143-155)
        self.maxchars_h = np.array([max(len(c) for c in r['c_ids']) for r in records], dtype=np.int32)
        up = lambda a: torch.from_numpy(a).to(self.dev)
        self.feat_off, self.sample_vid = up(feat_off), up(self.sample_vid_h)
        self.word_off, self.word_bank, self.char_off, self.char_bank = up(word_off), up(word_bank), up(char_off), up(char_bank)
        self.s_ind = self.e_ind = None
        if 's_ind' in records[0]:
            self.set_labels([r['s_ind'] for r in records], [r['e_ind'] for r in records])
        else:
            self._struct()

    def __len__(self):
        return len(self.records)

    def set_labels(self, s_ind, e_ind):
        """new pseudo labels (frame indices) after an update_label round"""
        self.s_ind = torch.from_numpy(np.ascontiguousarray(s_ind, dtype=np.int32)).to(self.dev)
        self.e_ind = torch.from_numpy(np.ascontiguousarray(e_ind, dtype=np.int32)).to(self.dev)
        self._struct()

    def _struct(self):
        p = lambda t: None if t is None else lib.ptr(t).value
        self.ds = lib.hual_dataset(p(self.feat_bank), p(self.feat_off), self.vdim, p(self.sample_vid), p(self.word_off),
                                   p(self.word_bank), p(self.char_off), p(self.char_bank), p(self.s_ind), p(self.e_ind))

    def batch_shape(self, sel):
        """(T, L, C) of the batch: maxima of its lengths, as the reference's padding produces them"""
        sel = np.asarray(sel)
        return int(self.vlen_h[sel].max()), int(self.nwords_h[sel].max()), int(self.maxchars_h[sel].max())

    def max_shape(self, min_chars=None):
        """(T, L, C) no batch of this set exceeds"""
        C = int(self.maxchars_h.max())
        return int(self.vlen_h.max()), int(self.nwords_h.max()), max(C, min_chars) if min_chars else C

    def feed_buffers(self, batch_size, min_chars=None, labels=True):
        """ONE set of feed buffers sized for the largest batch of the set.  assemble(..., buffers=) hands out views of them in
        the batch's own padded shape, so the device addresses of the feeds never change from step to step (what a per-shape
        hipGraph of the train step needs) and nothing is allocated inside the epoch loop (runner_utils.py:139-159)."""
        T, L, C = self.max_shape(min_chars)
        f32, i32, d, B = torch.float32, torch.int32, self.dev, int(batch_size)
        buf = dict(video=torch.empty(B * T * self.vdim, dtype=f32, device=d), video_seq_len=torch.empty(B, dtype=i32, device=d),
                   word_ids=torch.empty(B * L, dtype=i32, device=d), char_ids=torch.empty(B * L * C, dtype=i32, device=d),
                   sel=torch.empty(B, dtype=i32, device=d), shape=(B, T, L, C))
        if labels and self.s_ind is not None:
            buf.update(y1=torch.empty(B * T, dtype=f32, device=d), y2=torch.empty(B * T, dtype=f32, device=d),
                       match_labels=torch.empty(B * T, dtype=i32, device=d), inner_labels=torch.empty(B * T, dtype=f32, device=d))
        return buf

    def feed_views(self, B, T, L, C, buffers, labels=True):
        """views of feed_buffers() in the padded shape (B, T, L, C): the tensors assemble(..., buffers=) fills, without the launch"""
        Bm, Tm, Lm, Cm = buffers['shape']
        if B > Bm or T > Tm or L > Lm or C > Cm:
            raise ValueError('batch (%d,%d,%d,%d) exceeds the feed buffers %s' % (B, T, L, C, (buffers['shape'],)))
        out = dict(video=buffers['video'][:B * T * self.vdim].view(B, T, self.vdim), video_seq_len=buffers['video_seq_len'][:B],
                   word_ids=buffers['word_ids'][:B * L].view(B, L), char_ids=buffers['char_ids'][:B * L * C].view(B, L, C))
        if labels and 'y1' in buffers:
            out.update(y1=buffers['y1'][:B * T].view(B, T), y2=buffers['y2'][:B * T].view(B, T),
                       match_labels=buffers['match_labels'][:B * T].view(B, T), inner_labels=buffers['inner_labels'][:B * T].view(B, T))
        return out

    def enqueue_assemble_cursor(self, views, ids_dev, cursor):
        """the assembly launch of an epoch loop whose position lives on the device (hual_assemble_batch_cursor): fills `views`
        (feed_views) with the batch ids_dev[cursor[0] .. + B).  The same arguments every step of a padded shape: it is captured
        into the shape's step graph (Trainer.run_epoch)."""
        B, T, _ = views['video'].shape
        L, C = views['word_ids'].shape[1], views['char_ids'].shape[2]
        labels = 'y1' in views
        p = lib.ptr
        lib.check(self._lib.hual_assemble_batch_cursor(
            ctypes.byref(self.ds), p(ids_dev), p(cursor), B, T, L, C, p(views['video']), p(views['video_seq_len']), p(views['word_ids']),
            p(views['char_ids']), p(views['y1']) if labels else None, p(views['y2']) if labels else None,
            p(views['match_labels']) if labels else None, p(views['inner_labels']) if labels else None, lib.stream_ptr()))

    def assemble(self, sel, out=None, labels=True, min_chars=None, buffers=None, sel_dev=None, carry=None, shape=None):
        """Gather the batch `sel` (sample ids) on the device.  Returns a dict of device tensors named like the feeds of
        model.py:16-27.  out: a dict from a previous call with the same shape to write into (static buffers).
        buffers: feed_buffers() - the returned tensors are views of them (no allocation).  sel_dev: the same ids already on the
        device (a slice of the epoch's permutation): nothing is uploaded for this batch.  carry: (src, dst) int64 device tensors of
        equal size - the launch also copies src to dst (hual_assemble_batch_carry: the previous step's spans into the epoch's bank).
        shape: (T, L, C) to pad to instead of the batch's own maxima - a data-parallel shard is padded to the maxima of the GLOBAL
        batch (hual_amd/dist.py shard_plan)."""
        sel = np.ascontiguousarray(sel, dtype=np.int32)
        B = len(sel)
        T, L, C = self.batch_shape(sel)
        if min_chars:
            C = max(C, min_chars)
        if shape is not None:
            if shape[0] < T or shape[1] < L or shape[2] < C:
                raise ValueError('padded shape %s is smaller than the batch needs (%d, %d, %d)' % (tuple(shape), T, L, C))
            T, L, C = (int(x) for x in shape)
        labels = labels and self.s_ind is not None
        if buffers is not None:
            Bm, Tm, Lm, Cm = buffers['shape']
            if B > Bm or T > Tm or L > Lm or C > Cm:
                raise ValueError('batch (%d,%d,%d,%d) exceeds the feed buffers %s' % (B, T, L, C, (buffers['shape'],)))
            out = dict(video=buffers['video'][:B * T * self.vdim].view(B, T, self.vdim), video_seq_len=buffers['video_seq_len'][:B],
                       word_ids=buffers['word_ids'][:B * L].view(B, L), char_ids=buffers['char_ids'][:B * L * C].view(B, L, C),
                       sel=buffers['sel'][:B] if sel_dev is None else sel_dev)
            if labels:
                out.update(y1=buffers['y1'][:B * T].view(B, T), y2=buffers['y2'][:B * T].view(B, T),
                           match_labels=buffers['match_labels'][:B * T].view(B, T), inner_labels=buffers['inner_labels'][:B * T].view(B, T))
        elif out is None or out['video'].shape != (B, T, self.vdim) or out['char_ids'].shape != (B, L, C):
            f32, i32, d = torch.float32, torch.int32, self.dev
            out = dict(video=torch.empty(B, T, self.vdim, dtype=f32, device=d), video_seq_len=torch.empty(B, dtype=i32, device=d),
                       word_ids=torch.empty(B, L, dtype=i32, device=d), char_ids=torch.empty(B, L, C, dtype=i32, device=d),
                       sel=torch.empty(B, dtype=i32, device=d))
            if labels:
                out.update(y1=torch.empty(B, T, dtype=f32, device=d), y2=torch.empty(B, T, dtype=f32, device=d),
                           match_labels=torch.empty(B, T, dtype=i32, device=d), inner_labels=torch.empty(B, T, dtype=f32, device=d))
        if sel_dev is None:
            out['sel'].copy_(torch.from_numpy(sel))
        else:
            assert sel_dev.dtype == torch.int32 and sel_dev.numel() == B and sel_dev.is_cuda
            out['sel'] = sel_dev
        p = lib.ptr
        csrc, cdst, cn = None, None, 0
        if carry is not None:
            src, dst = carry
            assert src.dtype == torch.int64 and dst.dtype == torch.int64 and src.is_contiguous() and dst.is_contiguous() and src.numel() == dst.numel()
            csrc, cdst, cn = p(src), p(dst), src.numel()
        lib.check(self._lib.hual_assemble_batch_carry(
            ctypes.byref(self.ds), p(out['sel']), B, T, L, C, p(out['video']), p(out['video_seq_len']), p(out['word_ids']),
            p(out['char_ids']), p(out['y1']) if labels else None, p(out['y2']) if labels else None,
            p(out['match_labels']) if labels else None, p(out['inner_labels']) if labels else None, csrc, cdst, cn, lib.stream_ptr()))
        return out

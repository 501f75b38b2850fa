"""Host-fed training: the reference's own feed path as a pipeline (second half: "pinned-memory prefetch of [B,T,V]
features").

/root/reference/utils/runner_utils.py:139-159 walks `train_loader.batch_iter()`: every step `TrainLoader.process_batch`
(/root/reference/utils/data_loader.py:30-98) pads a batch in pageable numpy memory and `sess.run(feed_dict=...)` uploads it synchronously
before the step starts - 33.5 MB per step at the bench shape (B64 T128 vdim1024 float32).  The device-resident training set
(hual_amd/dataset.py) removes that traffic altogether and is what the bench times; THIS module is for callers that keep producing host
batches (features that do not fit the HBM, an existing loader): batch k + 1 is written into pinned staging memory and uploaded with ONE
asynchronous copy on a copy stream while step k runs, and the step reads the uploaded slot in place (Trainer.set_batch_device: one step
graph per padded shape and slot).

    feeder = HostFeeder(trainer, capacity=(B, T, L, C), vdim=V)
    spans = feeder.run_epoch(loader.batch_iter(), lr, drop_rate)           # any iterable of host batches; per-step (start, end)
    # or, without the staging copy: let the loader write into pinned memory
    views = feeder.stage_views(B, T, L, C)          # numpy views of the next slot, to be filled by the producer
    feeder.submit(lr, drop_rate)

A host batch is the tuple `TrainLoader.process_batch` returns behind the records - (vfeats, vfeat_lens, word_ids, char_ids, y1, y2,
match_labels, inner_labels) - or a dict with the feed names of hual_amd/dataset.py.  The C ABI below is unchanged: device pointers in,
the pipeline is host code."""
import numpy as np
import torch

FEEDS = ('video', 'y1', 'y2', 'inner_labels', 'video_seq_len', 'word_ids', 'char_ids', 'match_labels')


def _layout(B, T, L, C, V, vbytes):
    """byte offsets of the eight feeds of one batch in a slot, back to back, 256-byte aligned (ONE copy uploads them all)"""
    sizes = dict(video=B * T * V * vbytes, y1=B * T * 4, y2=B * T * 4, inner_labels=B * T * 4, video_seq_len=B * 4,
                 word_ids=B * L * 4, char_ids=B * L * C * 4, match_labels=B * T * 4)
    off, o = {}, 0
    for k in FEEDS:
        off[k] = (o, sizes[k])
        o += (sizes[k] + 255) // 256 * 256
    return off, o


class HostFeeder:
    def __init__(self, trainer, capacity, vdim, video_dtype=torch.float32, depth=2, staging_threads=4):
        """capacity: the largest (B, T, L, C) a batch may have; depth: slots (2 = the next batch uploads while the current one trains);
        staging_threads: feed() copies the clip features of a pageable batch into the pinned slot with that many threads (numpy's copy
        releases the GIL; one thread moves ~17 GB/s on the test box = 2 ms for the bench batch, longer than its train step)"""
        assert video_dtype in (torch.float32, torch.bfloat16) and depth >= 2
        if trainer.dp:
            # a data-parallel shard must be padded to the GLOBAL batch's shape with the global matching denominator (hual_amd/dist.py
            # shard_plan); that loop exists for the device-resident set only (Trainer.run_epoch)
            raise ValueError('HostFeeder is single-process: the data-parallel epoch loop shards the device-resident set (Trainer.run_epoch)')
        self._pool = None
        if staging_threads > 1:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=staging_threads)
        self._nthr = max(1, int(staging_threads))
        self.tr, self.V, self.vdt, self.depth = trainer, int(vdim), video_dtype, depth
        self.vbytes = 4 if video_dtype == torch.float32 else 2
        dev = trainer.m.device
        self.cap = tuple(int(x) for x in capacity)
        _, total = _layout(*self.cap, self.V, self.vbytes)
        self.host = [torch.empty(total, dtype=torch.uint8, pin_memory=True) for _ in range(depth)]
        self.dev = [torch.empty(total, dtype=torch.uint8, device=dev) for _ in range(depth)]
        self.copy_stream = torch.cuda.Stream(device=dev)
        self.uploaded = [torch.cuda.Event() for _ in range(depth)]       # slot's upload finished (copy stream)
        self.consumed = [torch.cuda.Event() for _ in range(depth)]       # the step that read the slot finished (compute stream)
        self._used = [False] * depth
        self.n = 0                       # batches submitted
        self._staged = None              # (slot, shape, layout, bytes) between stage_views() and submit()
        self.spans = []                  # per step: device clone of the [2, B] predicted span indices
        trainer.reserve(*self.cap)       # workspace and fetch tensors for the largest shape: nothing is allocated inside the loop
        self.stats = dict(batches=0, bytes_uploaded=0)

    # -------------------------------------------------------------------------------------------------
    def _views(self, buf, shape, lay, numpy=False):
        B, T, L, C = shape
        shp = dict(video=(B, T, self.V), y1=(B, T), y2=(B, T), inner_labels=(B, T), video_seq_len=(B,), word_ids=(B, L),
                   char_ids=(B, L, C), match_labels=(B, T))
        dt = dict(video=self.vdt, y1=torch.float32, y2=torch.float32, inner_labels=torch.float32, video_seq_len=torch.int32,
                  word_ids=torch.int32, char_ids=torch.int32, match_labels=torch.int32)
        out = {}
        for k in FEEDS:
            o, n = lay[k]
            v = buf[o:o + n].view(dt[k]).view(shp[k])
            if numpy:
                # bfloat16 has no numpy type: the producer writes its bit patterns as uint16
                v = v.view(torch.int16).numpy().view(np.uint16) if dt[k] == torch.bfloat16 else v.numpy()
            out[k] = v
        return out

    def stage_views(self, B, T, L, C):
        """numpy views of the next slot's pinned memory for a batch padded to (B, T, L, C): the producer fills them (every element -
        the slot holds an older batch), then calls submit().  Waits until the slot's previous upload has left the host memory."""
        assert self._staged is None, 'submit() the staged batch first'
        assert B <= self.cap[0] and T <= self.cap[1] and L <= self.cap[2] and C <= self.cap[3], 'batch larger than the feeder capacity'
        k = self.n % self.depth
        if self._used[k]:
            self.uploaded[k].synchronize()
        lay, nbytes = _layout(B, T, L, C, self.V, self.vbytes)
        self._staged = (k, (B, T, L, C), lay, nbytes)
        return self._views(self.host[k], (B, T, L, C), lay, numpy=True)

    def submit(self, lr, drop_rate):
        """upload the staged batch (asynchronously, behind the last reader of its device slot) and enqueue its train step behind the
        upload.  Returns at once: the host is free to stage the next batch while the device works."""
        k, shape, lay, nbytes = self._staged
        self._staged = None
        cur = torch.cuda.current_stream()
        with torch.cuda.stream(self.copy_stream):
            if self._used[k]:
                self.copy_stream.wait_event(self.consumed[k])
            self.dev[k][:nbytes].copy_(self.host[k][:nbytes], non_blocking=True)
            self.uploaded[k].record(self.copy_stream)
        cur.wait_event(self.uploaded[k])
        tr = self.tr
        tr.set_batch_device(self._views(self.dev[k], shape, lay))
        tr.step(lr=lr, drop_rate=drop_rate)
        self.spans.append(tr.spans[:, :shape[0]].clone())
        self.consumed[k].record(cur)
        self._used[k] = True
        self.n += 1
        self.stats['batches'] += 1
        self.stats['bytes_uploaded'] += nbytes

    # -------------------------------------------------------------------------------------------------
    def feed(self, batch, lr, drop_rate):
        """one host batch (pageable numpy arrays, as the reference's loader builds them): staged with one memcpy per feed, submitted"""
        if not isinstance(batch, dict):
            batch = tuple(batch)[-8:]                 # process_batch returns (records, vfeats, vfeat_lens, word_ids, char_ids, ...)
            batch = dict(video=batch[0], video_seq_len=batch[1], word_ids=batch[2], char_ids=batch[3], y1=batch[4], y2=batch[5],
                         match_labels=batch[6], inner_labels=batch[7])
        B, T, V = np.shape(batch['video'])
        L, C = np.shape(batch['word_ids'])[1], np.shape(batch['char_ids'])[2]
        if V != self.V:
            raise ValueError('feature width %d, feeder built for %d' % (V, self.V))
        if int(np.max(batch['video_seq_len'])) != T:
            raise ValueError('video T must equal max(video_seq_len) - model.py:31')
        views = self.stage_views(B, T, L, C)
        for k in FEEDS:
            src = batch[k]
            if k == 'video' and self.vdt == torch.bfloat16:
                # float32 features of a bfloat16 feed: rounded once, on the host (round to nearest even like Trainer.set_batch)
                t = torch.as_tensor(np.ascontiguousarray(src, dtype=np.float32)).to(torch.bfloat16)
                views[k][...] = t.view(torch.int16).numpy().view(np.uint16)
            elif k == 'video' and self._pool is not None and B >= self._nthr:
                src = np.asarray(src)
                cuts = [B * i // self._nthr for i in range(self._nthr + 1)]
                list(self._pool.map(lambda ab: np.copyto(views['video'][ab[0]:ab[1]], src[ab[0]:ab[1]], casting='same_kind'),
                                    zip(cuts[:-1], cuts[1:])))
            else:
                np.copyto(views[k], np.asarray(src), casting='same_kind' if views[k].dtype.kind == 'f' else 'unsafe')
        self.submit(lr, drop_rate)

    def feed_records(self, records, visual_feats, lr, drop_rate, min_chars=4):
        """one batch straight from the loader's INPUTS (data_loader.py:30-98: records with vid / w_ids / c_ids / s_ind / e_ind and the
        {vid: float32 [n, V]} features): padded directly into the pinned slot - no intermediate batch, no staging copy; the clips are
        copied by the staging threads.  float32 feed only."""
        from . import data
        if self.vdt != torch.float32:
            return self.feed(data.process_train_batch(records, visual_feats), lr, drop_rate)
        feats = [visual_feats[r['vid']] for r in records]
        B = len(records)
        lens = np.array([f.shape[0] for f in feats], dtype=np.int32)
        T = int(lens.max())
        L = max(len(r['w_ids']) for r in records)
        C = max(int(min_chars), max(len(w) for r in records for w in r['c_ids']))
        if feats[0].shape[1] != self.V:
            raise ValueError('feature width %d, feeder built for %d' % (feats[0].shape[1], self.V))
        v = self.stage_views(B, T, L, C)
        video = v['video']

        def fill(ab):
            for b in range(ab[0], ab[1]):
                n = int(lens[b])
                video[b, :n] = feats[b]
                video[b, n:] = 0.0
        if self._pool is not None and B >= self._nthr:
            cuts = [B * i // self._nthr for i in range(self._nthr + 1)]
            list(self._pool.map(fill, zip(cuts[:-1], cuts[1:])))
        else:
            fill((0, B))
        v['video_seq_len'][:] = lens
        w, c = v['word_ids'], v['char_ids']
        w[...] = 0
        c[...] = 0
        for b, r in enumerate(records):
            w[b, :len(r['w_ids'])] = r['w_ids']
            for l, cw in enumerate(r['c_ids']):
                c[b, l, :len(cw)] = cw
        y1, y2, match, inner = data.make_labels([r['s_ind'] for r in records], [r['e_ind'] for r in records], lens)
        v['y1'][...] = y1
        v['y2'][...] = y2
        v['match_labels'][...] = match
        v['inner_labels'][...] = inner
        self.submit(lr, drop_rate)

    def collect(self):
        """wait for the submitted steps and return the predicted (start, end) indices of every step since the last collect() as numpy
        arrays - the ONE synchronisation of a host-fed epoch"""
        torch.cuda.synchronize()
        out = [s.cpu().numpy() for s in self.spans]
        self.spans = []
        return [(o[0], o[1]) for o in out]

    def run_epoch(self, batches, lr, drop_rate, prefetch=2):
        """runner_utils.py:139-159 over an iterable of host batches; returns collect().  prefetch > 0: the iterable is drained by a
        producer thread that keeps up to that many batches ready (the loader's padding loops overlap the staging copies and the
        launches of this thread; numpy releases the GIL in its large copies)."""
        if prefetch and prefetch > 0:
            import queue
            import threading
            q, end = queue.Queue(maxsize=int(prefetch)), object()

            def produce():
                try:
                    for b in batches:
                        q.put(b)
                    q.put(end)
                except BaseException as e:          # the loader failed: hand the error to the consumer
                    q.put(e)
            th = threading.Thread(target=produce, daemon=True)
            th.start()
            while True:
                b = q.get()
                if b is end:
                    break
                if isinstance(b, BaseException):
                    raise b
                self.feed(b, lr, drop_rate)
            th.join()
        else:
            for b in batches:
                self.feed(b, lr, drop_rate)
        return self.collect()

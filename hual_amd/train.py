"""Training-step driver for the HIP SeqPAN path: static device buffers, the whole step
(forward + backward + clip + AdamWD) enqueued through the C ABI and - single GPU - replayed as ONE hipGraph.

Equivalent of the loop body of /root/reference/utils/runner_utils.py:144-147 (feed_dict upload + sess.run).
"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

from . import dist as hdist
from . import lib


class Trainer:
    def __init__(self, model, world=1, use_graph=True, force_dp=False):
        self.m = model
        self.world = world
        self.dp = world > 1 or force_dp          # force_dp: run the data-parallel code path on one rank (tests)
        self.use_graph = use_graph and not self.dp
        self.dp_graph = use_graph and self.dp     # data-parallel step as a graph (nccl backend only, see _step_dp)
        self._dp_graph_failed = False
        self.dp_launch = 'eager'                  # how the last data-parallel step was launched: eager / segment graphs / one graph
        self._prezero_token = ctypes.c_uint64(0)
        self.graph = None
        self.graph_drop = None
        self.shape = None
        self._tables_ready = False
        self._lib = lib.load()
        # device-fed mode (set_batch_device): per-shape cache of ABI structs and step graphs, outputs at the largest shape seen
        self.dynamic = False
        self.graph_shapes = use_graph and not self.dp
        # per-shape cache of the device-fed mode.  The reference's own annotations give ~340 distinct (L, C) padded shapes per
        # ActivityNet epoch at batch 16 and more with varying T (tests/golden/lengths_anet.npz): the bound is far above that (an entry
        # is a few ctypes structs, a 60 KB job table and - once captured - a hipGraphExec of ~55 kernel nodes), evictions are counted
        self.cache_limit = 8192
        self.capture_after = 1       # sightings of a shape that are launched eagerly before its step graph is captured
        self._cache = {}
        self._entry = None
        self._out_cap = None
        self._dp_B = None
        self._cap_stream = None
        self._host_denom = None      # data-parallel epoch loop: the matching-loss denominator computed on the host (shard_plan)
        self._loop = None            # run_epoch: ids / cursor / span bank of the device-side epoch position, the dataset, the current views
        self._table_bytes = int(self._lib.hual_seqpan_dw_table_bytes())
        self._dw_table = None        # static mode: this trainer's own job table (hual_run_opts.dw_table), never in the shared workspace
        if self.dp and world > 1:
            hdist.enable_custom_allreduce(model.grads)      # collective; a no-op unless HUAL_ALLREDUCE=custom (default: RCCL)
        self.stats = dict(eager=0, captured=0, replayed=0, evicted=0, capture_failed=0)

    # ------------------------------------------------------------------ static batch buffers
    def set_batch(self, video, lens, word_ids, char_ids, y1, y2, match_labels, inner_labels, video_dtype=torch.float32):
        """video_dtype=torch.bfloat16 keeps the clip features in HBM as bfloat16 (hual_batch.video_dtype; the float32 input is
        rounded once here)"""
        m, dev = self.m, self.m.device
        B, T, V = np.shape(video)
        L, C = np.shape(word_ids)[1], np.shape(char_ids)[2]
        # model.py:31 pads to the longest clip of the batch.  A data-parallel shard must be padded to the longest clip of the
        # GLOBAL batch to reproduce the single-process numbers (the reference's conv_block does not mask: rows behind a clip's
        # end feed its last valid rows, so the results depend on the padded length) - hence T >= max(lens) there.
        if int(np.max(lens)) != T and not (self.dp and int(np.max(lens)) < T):
            raise ValueError('video T must equal max(video_seq_len) - model.py:31')
        if self.dp and os.environ.get('HUAL_DP_SKIP_T_CHECK') != '1':
            hdist.check_padded_length(int(np.max(lens)), T)      # raises on EVERY rank together
        shape = (B, T, L, C, video_dtype)
        if shape != self.shape or self.dynamic:
            self.dynamic = False
            self.shape = shape
            self.graph = None
            self._tables_ready = False
            f32, i32 = torch.float32, torch.int32
            self.video = torch.empty(B, T, V, device=dev, dtype=video_dtype)
            self.lens = torch.empty(B, device=dev, dtype=i32)
            self.word_ids = torch.empty(B, L, device=dev, dtype=i32)
            self.char_ids = torch.empty(B, L, C, device=dev, dtype=i32)
            self.y1 = torch.empty(B, T, device=dev, dtype=f32)
            self.y2 = torch.empty(B, T, device=dev, dtype=f32)
            self.match = torch.empty(B, T, device=dev, dtype=i32)
            self.inner = torch.empty(B, T, device=dev, dtype=f32)
            self.start_logits = torch.empty(B, T, device=dev)
            self.end_logits = torch.empty(B, T, device=dev)
            self.match_scores = torch.empty(B, T, 4, device=dev)
            self.start_index = torch.empty(B, device=dev, dtype=torch.int64)
            self.end_index = torch.empty(B, device=dev, dtype=torch.int64)
            self.loss_terms = torch.zeros(4, device=dev)
            self.ws = m._workspace(B, T, L, C)
            # the job table of the weight-gradient launch is this trainer's own: the workspace is shared with every other user of the
            # model (model.forward at another shape, a second Trainer), static_tables must not depend on what they leave there
            self._dw_table = torch.empty(self._table_bytes, dtype=torch.uint8, device=dev)
            self._cache.clear()       # (entries of an earlier device-fed phase hold the addresses of the old fetch tensors)
            self._out_cap = None
            p = lib.ptr
            self.bt = lib.hual_batch(p(self.video).value, p(self.lens).value, p(self.word_ids).value,
                                     p(self.char_ids).value, B, T, L, C, 1 if video_dtype == torch.bfloat16 else 0)
            self.lab = lib.hual_labels(p(self.y1).value, p(self.y2).value, p(self.match).value, p(self.inner).value)
            self.out = lib.hual_outputs(p(self.start_logits).value, p(self.end_logits).value, p(self.match_scores).value,
                                        p(self.start_index).value, p(self.end_index).value, p(self.loss_terms).value)
            if self.dp:
                self._alloc_dp(B)

        def put(dst, src, dt):
            dst.copy_(torch.as_tensor(np.ascontiguousarray(src), dtype=dt), non_blocking=False)
        put(self.video, video, torch.float32)        # copy_ converts to the buffer's dtype
        put(self.lens, lens, torch.int32)
        put(self.word_ids, word_ids, torch.int32)
        put(self.char_ids, char_ids, torch.int32)
        put(self.y1, y1, torch.float32)
        put(self.y2, y2, torch.float32)
        put(self.match, match_labels, torch.int32)
        put(self.inner, inner_labels, torch.float32)
        self._update_match_denominator()

    def _alloc_dp(self, B):
        """static buffers of the data-parallel exchange: [that | vhat] of the local / all samples, the [Bg,Bg] scratch of the
        alignment loss, its value, and the device scalar holding the matching-loss denominator.  One set per local batch size, kept for
        the trainer's lifetime: the segment graphs of a padded shape hold their addresses (an epoch's ragged last batch comes back
        every epoch)."""
        if not hasattr(self, '_dp_bufs'):
            self._dp_bufs = {}
        if B not in self._dp_bufs:
            dev, Bg = self.m.device, B * self.world
            self._dp_bufs[B] = (torch.empty(2 * Bg * Bg + Bg, device=dev), torch.empty(Bg, 256, device=dev), torch.zeros(1, device=dev),
                                torch.zeros(1, device=dev))
        self.align_scratch, self.feat_all, self.align_loss, self.denom_dev = self._dp_bufs[B]
        self.graph = None

    def _update_match_denominator(self):
        """exact data parallel: the masked matching loss divides by the GLOBAL valid-frame count; every rank
        uses n_global / world (+1e-12 of layers.py:173 on the global count) so that the rank average is the global masked
        mean.  Device side: sum of the lengths -> all-reduce -> scalar the kernels read (hual_run_opts.match_denom_dev);
        no host round trip, so it can sit in front of every batch of a device-fed loop."""
        self.match_denom = 0.0
        if not self.dp:
            return
        if self._host_denom is not None:
            # the epoch loop knows the global batch's lengths on the host: no collective - but the value still travels through the DEVICE
            # scalar (one fill launch): a step's options must not change from batch to batch of a padded shape, they are baked into
            # that shape's segment graphs (by value the first batch's denominator would be replayed for every later one)
            self.denom_dev.fill_(float(self._host_denom))
            return
        torch.sum(self.lens.to(torch.float32), dim=0, keepdim=True, out=self.denom_dev)
        hdist.allreduce_sum_(self.denom_dev)
        self.denom_dev.add_(1e-12).div_(float(self.world))

    def set_batch_device(self, feeds):
        """Point the step at feeds that already live on the device (DeviceDataset.assemble): no host copy, no upload.
        The padded shape changes from batch to batch (runner_utils.py:139-159: T / L / C are maxima over the batch), so this
        mode keeps a small cache keyed by (shape, feed addresses): the structs of the C ABI, and - single GPU, use_graph - a
        hipGraph of the whole step per shape.  A shape is launched eagerly the first time it is seen (first-use attribute calls
        happen outside any capture) and captured the second time; with static feed buffers (DeviceDataset.feed_buffers) and
        the one workspace of SeqPAN.reserve() every later batch of that shape is ONE graph launch.  Nothing is allocated,
        filled or synchronised when the shape changes."""
        m, dev = self.m, self.m.device
        B, T, V = feeds['video'].shape
        L, C = feeds['word_ids'].shape[1], feeds['char_ids'].shape[2]
        self.dynamic = True
        self.video, self.lens, self.word_ids, self.char_ids = (feeds[k] for k in ('video', 'video_seq_len', 'word_ids', 'char_ids'))
        self.y1, self.y2, self.match, self.inner = (feeds[k] for k in ('y1', 'y2', 'match_labels', 'inner_labels'))
        vdt = self.video.dtype
        self.shape = (B, T, L, C, vdt)
        self._ensure_outputs(B, T)
        self.ws = m._workspace(B, T, L, C)
        p = lib.ptr
        lp = self._loop
        key = (B, T, L, C, vdt, self.ws.data_ptr(), self._out_flat.data_ptr(), self.loss_terms.data_ptr(), self.spans.data_ptr(),
               0 if lp is None else (lp['ids'].data_ptr(), lp['cursor'].data_ptr(), lp['bank'].data_ptr())) + tuple(
            t.data_ptr() for t in (self.video, self.lens, self.word_ids, self.char_ids, self.y1, self.y2, self.match, self.inner))
        e = self._cache.get(key)
        if e is None:
            if len(self._cache) >= self.cache_limit:            # bounded: drop the least recently used shape
                self._cache.pop(next(iter(self._cache)))
                self.stats['evicted'] += 1
            # table: the shape's OWN job table of the weight-gradient launch - the workspace is shared between shapes, the table is
            # not, so from the shape's second step on nothing rewrites it (five launches fewer per step, eager or replayed)
            e = dict(seen=0, graph=None, drop=None, nograph=False, tables_ready=False,
                     table=torch.empty(self._table_bytes, dtype=torch.uint8, device=dev),
                     bt=lib.hual_batch(p(self.video).value, p(self.lens).value, p(self.word_ids).value, p(self.char_ids).value,
                                       B, T, L, C, 1 if vdt == torch.bfloat16 else 0),
                     lab=lib.hual_labels(p(self.y1).value, p(self.y2).value, p(self.match).value, p(self.inner).value))
        else:
            self._cache.pop(key)                                 # re-insert: most recently used last
        self._cache[key] = e
        self._entry = e
        self.bt, self.lab = e['bt'], e['lab']
        self.start_logits = self._out_flat[:B * T].view(B, T)
        self.end_logits = self._out_flat[self._out_cap[0] * self._out_cap[1]:][:B * T].view(B, T)
        self.match_scores = self._out_flat[2 * self._out_cap[0] * self._out_cap[1]:][:B * T * 4].view(B, T, 4)
        self.start_index, self.end_index = self.spans[0, :B], self.spans[1, :B]
        self.out = lib.hual_outputs(p(self.start_logits).value, p(self.end_logits).value, p(self.match_scores).value,
                                    p(self.start_index).value, p(self.end_index).value, p(self.loss_terms).value)
        self._dw_table = e['table']
        self._tables_ready = e['tables_ready']
        if self.dp and (self._dp_B != B):
            self._alloc_dp(B)
            self._dp_B = B
        self._update_match_denominator()

    def _ensure_outputs(self, B, T):
        """fetch tensors of the device-fed mode: one allocation for the largest (B, T) seen (growing it drops the cached graphs -
        they hold the old addresses; run_epoch reserves the set's maximum up front, so a loop never grows it); spans as ONE [2, B]
        tensor so a loop can bank both with one copy"""
        cap = self._out_cap
        if cap is not None and B <= cap[0] and T <= cap[1]:
            return
        Bc, Tc = max(B, cap[0] if cap else 0), max(T, cap[1] if cap else 0)
        dev = self.m.device
        self._out_cap = (Bc, Tc)
        self._out_flat = torch.empty(6 * Bc * Tc, device=dev)
        self.spans = torch.zeros(2, Bc, device=dev, dtype=torch.int64)
        self.loss_terms = torch.zeros(4, device=dev)
        self._cache.clear()

    def reserve(self, B, T, L, C):
        """size workspace and fetch tensors for the largest batch of a loop up front (no growth inside it)"""
        self.m.reserve(B, T, L, C)
        self._ensure_outputs(B, T)

    # ------------------------------------------------------------------ one step
    def _opts(self, drop_rate, align_external, defer_loss=True):
        """defer_loss: the forward leaves the closing of the loss to the backward's matching-head launch (one launch fewer per step,
        hual_run_opts.deferred_loss_terms) - for a forward that IS followed by its backward; False for a forward on its own"""
        # static_tables: all buffers of this trainer are static per shape, so after one backward on them the job tables
        # in the workspace stay valid (hual_run_opts.static_tables)
        return lib.hual_run_opts(float(drop_rate), lib.ptr(self.m.rng_state).value, float(self.match_denom),
                                 int(align_external),
                                 1 if (self._tables_ready and not os.environ.get('HUAL_NO_STATIC_TABLES')) else 0,
                                 lib.ptr(self.denom_dev).value if self.dp else None, 0,
                                 # the forward's first launch zeroes the gradient bucket (one launch fewer in backward); the
                                 # host word is the receipt the backward call checks and clears (hual_run_opts.prezero_token)
                                 lib.ptr(self.m.grads).value, ctypes.addressof(self._prezero_token),
                                 lib.ptr(self.loss_terms).value if defer_loss else None,
                                 lib.ptr(self._dw_table).value if self._dw_table is not None else None,
                                 self._table_bytes if self._dw_table is not None else 0)

    def _forward(self, opts):
        m = self.m
        lib.check(self._lib.hual_seqpan_forward(
            ctypes.byref(m.cfg), lib.ptr(m.params), lib.ptr(m.word_table), ctypes.byref(self.bt), ctypes.byref(self.lab),
            ctypes.byref(self.out), ctypes.byref(opts), lib.ptr(self.ws), self.ws.numel(), lib.stream_ptr()))

    def _backward(self, opts):
        m = self.m
        self._tables_ready = True
        if self.dynamic and self._entry is not None:
            self._entry['tables_ready'] = True
        lib.check(self._lib.hual_seqpan_backward(
            ctypes.byref(m.cfg), lib.ptr(m.params), lib.ptr(m.word_table), ctypes.byref(self.bt), ctypes.byref(self.lab),
            ctypes.byref(opts), lib.ptr(m.grads), lib.ptr(self.ws), self.ws.numel(), lib.stream_ptr()))

    def _adam(self, prescale):
        m = self.m
        # the Philox offset of the dropout stream advances in the same launch (rng_state[2] += 1)
        lp = self._loop
        if lp is not None:
            # ... and so does the epoch loop's device-side position: the step's spans go to their place in the bank, the cursor moves on
            nsp = self.spans.numel()
            lib.check(self._lib.hual_adamw_clip_step_loop(
                lib.ptr(m.params), lib.ptr(m.grads), lib.ptr(m.adam_m), lib.ptr(m.adam_v), lib.ptr(m.decay),
                m.params.numel(), lib.ptr(m.lr), float(m.cfg.clip_norm), float(prescale), lib.ptr(m.sqnorm),
                lib.ptr(m.rng_state), lib.ptr(lp['cursor']), lib.ptr(self.spans), lib.ptr(lp['bank']), nsp, int(self.shape[0]), nsp,
                lib.stream_ptr()))
            return
        lib.check(self._lib.hual_adamw_clip_step_rng(
            lib.ptr(m.params), lib.ptr(m.grads), lib.ptr(m.adam_m), lib.ptr(m.adam_v), lib.ptr(m.decay),
            m.params.numel(), lib.ptr(m.lr), float(m.cfg.clip_norm), float(prescale), lib.ptr(m.sqnorm),
            lib.ptr(m.rng_state), lib.stream_ptr()))

    def _enqueue_assembly(self):
        """epoch loop: the batch-assembly launch is the first launch of the step (and of the step's graph)"""
        lp = self._loop
        if lp is not None:
            lp['ds'].enqueue_assemble_cursor(lp['views'], lp['ids'], lp['cursor'])

    def _enqueue_single(self, drop_rate):
        self._enqueue_assembly()
        opts = self._opts(drop_rate, 0)
        self._forward(opts)
        self._backward(opts)
        self._adam(1.0)

    # one data-parallel step = three runs of our own launches with a collective between them
    def _dp_part_a(self, opts):
        self._enqueue_assembly()
        self._forward(opts)

    def _dp_gather(self):
        hdist.allgather_rows_(self.feat_all, self.m.tap('align.tv'))      # [that | vhat] rows as the forward left them: no copy

    def _dp_part_b(self, opts):
        m, B, fa = self.m, self.shape[0], self.feat_all
        lib.check(self._lib.hual_align_loss_rows(
            lib.ptr(fa), ctypes.c_void_p(fa.data_ptr() + 128 * 4), 256, fa.shape[0], hdist.rank() * B if self.world > 1 else 0, B,
            lib.ptr(self.align_scratch), lib.ptr(m.tap('d.align.that')), lib.ptr(m.tap('d.align.vhat')), lib.ptr(self.align_loss),
            float(self.world), lib.stream_ptr()))
        self._backward(opts)

    def _dp_reduce(self):
        hdist.allreduce_sum_(self.m.grads)

    def _dp_part_c(self):
        self._adam(1.0 / self.world)

    def _enqueue_dp(self, drop_rate):
        """one data-parallel step, enqueued without any host synchronisation: forward -> all-gather of the [B,256] alignment
        features -> global [Bg,Bg] alignment loss, gradient rows of the own samples written straight into the backward's
        workspace buffers -> backward -> ONE all-reduce of the flat gradient bucket -> clip + AdamWD on the averaged gradient"""
        opts = self._opts(drop_rate, 1)
        self._dp_part_a(opts)
        self._dp_gather()
        self._dp_part_b(opts)
        self._dp_reduce()
        self._dp_part_c()

    def _capture_dp_segments(self, drop_rate):
        """The three runs of OUR launches as three hipGraphs (records only, nothing executes); the collectives stay OUTSIDE and are
        issued eagerly between the replays - no RCCL call inside a graph (that form, _step_dp's full capture, has never run on two
        devices and stays opt-in), but 57 of the step's 59 host launches collapse into three.  Ranks need not agree on it: a rank on
        segment graphs and a rank on eager launches issue the same two collectives in the same order.  Returns None if a capture is
        refused."""
        if self._cap_stream is None:
            self._cap_stream = torch.cuda.Stream(device=self.m.device)
        cs, cur = self._cap_stream, torch.cuda.current_stream()
        opts = self._opts(drop_rate, 1)
        segs = []
        for part in (lambda: self._dp_part_a(opts), lambda: self._dp_part_b(opts), self._dp_part_c):
            g = torch.cuda.CUDAGraph()
            cs.wait_stream(cur)
            err = None
            with torch.cuda.stream(cs):
                # thread_local: the process group's watchdog thread may touch the device while this thread captures
                g.capture_begin(capture_error_mode='thread_local')
                try:
                    part()
                except BaseException as ex:
                    err = ex
                try:
                    g.capture_end()
                except RuntimeError as ex:
                    err = err or ex
            cur.wait_stream(cs)
            if err is not None:
                if isinstance(err, lib.HualError) or not isinstance(err, RuntimeError):
                    raise err
                print('[hual] data-parallel step: segment capture refused (%s) - eager launches' % str(err).splitlines()[0], file=sys.stderr)
                return None
            segs.append(g)
        return segs

    def _replay_dp_segments(self, segs):
        segs[0].replay()
        self._dp_gather()
        segs[1].replay()
        self._dp_reduce()
        segs[2].replay()

    def step(self, lr, drop_rate):
        m = self.m
        if m.lr_value != float(lr):               # the fed scalar changes once per epoch (main.py:61): no fill launch otherwise
            m.lr.fill_(float(lr))
            m.lr_value = float(lr)
        if self.dynamic:
            self._step_dynamic(drop_rate)
        elif self.dp:
            self._step_dp(drop_rate)
        elif not self.use_graph:
            self._enqueue_single(drop_rate)
        else:
            if self.graph is None or self.graph_drop != drop_rate:
                # warm-up outside capture (first-use hipFuncSetAttribute etc.) on a snapshot of the training state,
                # then capture the whole step; the first replay below is the first real step
                snap = [t.clone() for t in (m.params, m.adam_m, m.adam_v, m.rng_state)]
                self._enqueue_single(drop_rate)
                torch.cuda.synchronize()
                for t, s in zip((m.params, m.adam_m, m.adam_v, m.rng_state), snap):
                    t.copy_(s)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                # thread_local: a capture must not make other threads' HIP calls illegal (a loader thread that synchronises its own stream
                # would fail AND invalidate the capture in the default global mode); the capturing thread itself only launches kernels
                with torch.cuda.graph(g, capture_error_mode='thread_local'):
                    self._enqueue_single(drop_rate)
                self.graph, self.graph_drop = g, drop_rate
            self.graph.replay()
        m.global_step += 1

    def _step_dynamic(self, drop_rate):
        """device-fed mode: eager on the first `capture_after` sightings of a (shape, addresses) key, then captured, replayed after.
        Data parallel: eager launches (the collectives inside a step graph are opt-in, _step_dp)."""
        e = self._entry
        if self.dp:
            # segment graphs per shape (the collectives between them eager): default with world > 1 (HUAL_DP_GRAPH=seg), see _step_dp
            segmented = self.dp_graph and self._dp_mode() == 'seg' and not e['nograph']
            if segmented and e.get('segs') is not None and e['drop'] == drop_rate:
                self._replay_dp_segments(e['segs'])
                self.stats['replayed'] += 1
                return
            if segmented and e['seen'] >= self.capture_after:
                segs = self._capture_dp_segments(drop_rate)
                if segs is not None:
                    e['segs'], e['drop'] = segs, drop_rate
                    self._replay_dp_segments(segs)
                    self.stats['captured'] += 1
                    return
                e['nograph'] = True
                self.stats['capture_failed'] += 1
            self._enqueue_dp(drop_rate)
            e['seen'] += 1
            self.stats['eager'] += 1
            return
        if e['graph'] is not None and e['drop'] == drop_rate:
            e['graph'].replay()
            self.stats['replayed'] += 1
            return
        if not self.graph_shapes or e['nograph'] or e['seen'] < self.capture_after:
            self._enqueue_single(drop_rate)
            e['seen'] += 1
            self.stats['eager'] += 1
            return
        # capture on a side stream by hand (records, executes nothing: the replay below IS the step).  Not `with torch.cuda.graph(g)`:
        # that context synchronises the device and runs the garbage collector on entry - tens of milliseconds per shape.  A capture
        # costs ~1 ms of host time (measured over 250 shapes of the ActivityNet length distribution, scripts/exp/epoch_real.py).
        g = torch.cuda.CUDAGraph()
        if self._cap_stream is None:
            self._cap_stream = torch.cuda.Stream(device=self.m.device)
        cs, cur = self._cap_stream, torch.cuda.current_stream()
        cs.wait_stream(cur)
        err = None
        with torch.cuda.stream(cs):
            g.capture_begin(capture_error_mode='thread_local')      # (other threads stay free to use the GPU: see step())
            try:
                self._enqueue_single(drop_rate)
            except BaseException as ex:          # our own launch failed under capture: end the capture, keep THIS error
                err = ex
            try:
                g.capture_end()
            except RuntimeError as ex:           # the runtime refused the capture
                err = err or ex
        cur.wait_stream(cs)
        if err is not None:
            # nothing was executed.  A refused capture leaves the shape on eager launches for good; an error of one of our own
            # launches (lib.HualError: bad arguments) would fail eagerly too and is raised
            self.stats['capture_failed'] += 1
            e['nograph'] = True
            if isinstance(err, lib.HualError) or not isinstance(err, RuntimeError):
                raise err
            print('[hual] step graph capture refused for shape %s (%s): eager launches' % (self.shape[:4], str(err).splitlines()[0]),
                  file=sys.stderr)
            self._enqueue_single(drop_rate)
            e['seen'] += 1
            self.stats['eager'] += 1
            return
        e['graph'], e['drop'] = g, drop_rate
        g.replay()
        self.stats['captured'] += 1

    def _dp_mode(self):
        """HUAL_DP_GRAPH: '1' = the whole step incl. its RCCL collectives as ONE hipGraph (default on a one-rank group: the rehearsal the
        GPU suite runs; opt-in with more ranks - never validated on two devices), 'seg' = three graphs of our own launches with the two
        collectives eager between them (default with more than one rank), '0' = eager launches"""
        return os.environ.get('HUAL_DP_GRAPH', '1' if self.world == 1 else 'seg')

    def _step_dp(self, drop_rate):
        """Three launch modes (_dp_mode).  With more than one rank the default is SEGMENT graphs: forward | backward | optimizer as three
        hipGraphs of our own launches, the all-gather and the all-reduce issued eagerly between the replays - no collective inside a
        graph, and a rank whose capture is refused simply launches eagerly (same collective sequence).  Measured on a one-rank RCCL
        group: eager 1.253 ms/step, one graph with the collectives inside 1.210, the single-GPU graph 1.195.
        With the nccl backend (RCCL: its collectives are stream operations) the data-parallel step, collectives included, CAN also be
        captured into ONE hipGraph and replayed.  That is the default only on ONE rank (the forced-collectives rehearsal the GPU
        suite runs); with more than one rank it is opt-in (HUAL_DP_GRAPH=1) until a multi-GPU run has validated capture and replay
        of the all-gather / all-reduce pair - no such run exists yet.  There the graph-or-eager decision is COLLECTIVE: after
        the capture attempt the ranks all-reduce(MIN) an ok flag, so either every rank replays or every rank launches eagerly.  Only a
        refused capture (a RuntimeError from torch / HIP) is treated as "no graph"; an error raised by one of our own launches
        (lib.HualError) propagates.  Other backends (gloo: host collectives) always launch eagerly."""
        mode = self._dp_mode()
        if mode == 'seg' and self.dp_graph and not self._dp_graph_failed:
            m = self.m
            if self.graph is None or self.graph_drop != drop_rate:
                snap = [t.clone() for t in (m.params, m.adam_m, m.adam_v, m.rng_state)]
                self._enqueue_dp(drop_rate)          # warm-up outside capture (first-use attributes, RCCL channel setup)
                torch.cuda.synchronize()
                for t, sn in zip((m.params, m.adam_m, m.adam_v, m.rng_state), snap):
                    t.copy_(sn)
                torch.cuda.synchronize()
                segs = self._capture_dp_segments(drop_rate)
                if segs is None:
                    self._dp_graph_failed = True
                    self._enqueue_dp(drop_rate)
                    return
                self.graph, self.graph_drop = segs, drop_rate
            self._replay_dp_segments(self.graph)
            self.dp_launch = 'three hipGraphs of the launches, the two collectives eager between them'
            return
        want_graph = (mode == '1' and self.dp_graph and not self._dp_graph_failed and hdist.backend() == 'nccl')
        if not want_graph:
            self._enqueue_dp(drop_rate)
            self.dp_launch = 'eager'
            return
        self.dp_launch = 'hipGraph with the collectives captured'
        m = self.m
        if self.graph is None or self.graph_drop != drop_rate:
            snap = [t.clone() for t in (m.params, m.adam_m, m.adam_v, m.rng_state)]
            self._enqueue_dp(drop_rate)              # warm-up outside capture (first-use attributes, RCCL channel setup)
            torch.cuda.synchronize()
            for t, sn in zip((m.params, m.adam_m, m.adam_v, m.rng_state), snap):
                t.copy_(sn)
            torch.cuda.synchronize()
            g, err, fatal = None, None, None
            try:
                g = torch.cuda.CUDAGraph()
                # thread_local: the process group's watchdog thread may touch the device while this thread captures
                with torch.cuda.graph(g, capture_error_mode='thread_local'):
                    self._enqueue_dp(drop_rate)
            except lib.HualError as e:               # one of our launches failed: not a capture refusal - but the peers are about to
                g, fatal = None, e                   # enter the collective below, so it is raised only after this rank has joined it
            except RuntimeError as e:                # capture of the collectives refused
                g, err = None, e
                torch.cuda.synchronize()
            except BaseException as e:               # anything else: same rule - join the collective first
                g, fatal = None, e
            # the same decision on every rank: 2 = captured, 1 = capture refused (eager launches), 0 = a launch failed (everybody raises)
            state = hdist.global_min(2 if g is not None else (0 if fatal is not None else 1))
            if fatal is not None:
                raise fatal
            if state == 0:
                raise lib.HualError('data-parallel step: a launch failed on another rank during graph capture')
            if state == 1:
                self._dp_graph_failed = True
                self.graph = None
                print('[hual] data-parallel step: graph capture %s - eager launches on every rank'
                      % ('refused here (%s)' % str(err).splitlines()[0] if err is not None else 'refused on another rank'), file=sys.stderr)
                self._enqueue_dp(drop_rate)
                self.dp_launch = 'eager'
                return
            self.graph, self.graph_drop = g, drop_rate
        self.graph.replay()

    # ------------------------------------------------------------------ one epoch on a device-resident training set
    def run_epoch(self, dataset, order, batch_size, lr, drop_rate, min_chars=4, want_spans=True):
        """The loop of runner_utils.py:139-159 (train_epoch) on a DeviceDataset: for every batch of `order` (sample ids, already
        shuffled - data_loader.py:23-28) assemble the feeds on the device, run the train step, bank the predicted spans.

        Nothing in the loop waits for the device and nothing is uploaded per step: the permutation goes up ONCE, a batch's ids
        are a slice of it; the feeds are views of one set of max-shape buffers; the workspace and the fetch tensors are sized
        once; a step is one hipGraph launch for every padded shape seen before (set_batch_device); the spans of each step are
        copied (device to device, by the NEXT step's assembly launch) into an epoch-long bank that is fetched with ONE transfer after
        the last step - the reference's IoU bookkeeping (runner_utils.py:150-156) only needs them at the end of the epoch.

        Data parallel (world > 1; `batch_size` = clips per rank): every rank calls this with the SAME dataset and order.  A global
        batch is batch_size * world consecutive ids, rank r trains on its r-th slice, padded to the GLOBAL batch's (T, L, C); the
        matching-loss denominator comes from the global batch's lengths on the host; plan: hual_amd/dist.py shard_plan (the < world
        clips that do not fill the last round are dropped from the epoch).  The spans of all ranks are gathered once, after the
        last step.

        Returns (start, end) int64 numpy arrays for the ids of `self.last_epoch_ids` (= `order` unless clips were dropped), in that
        order - (None, None) with want_spans False."""
        m = self.m
        order = np.ascontiguousarray(order, dtype=np.int32)
        world, rank = (self.world, hdist.rank()) if self.world > 1 else (1, 0)
        if world > 1:
            hdist.check_same(int(np.dot(order.astype(np.int64) % 1000003, np.arange(1, len(order) + 1) % 1009) % (1 << 40)) + len(order),
                             "the epoch's sample order")
        steps, dropped = hdist.shard_plan(order, batch_size, world, dataset.vlen_h, dataset.nwords_h, dataset.maxchars_h, min_chars)
        nsteps, bs = len(steps), int(batch_size)
        self.last_epoch_ids = np.concatenate([st['ids'] for st in steps]) if steps else order[:0]
        self.stats['dropped'] = self.stats.get('dropped', 0) + dropped
        if nsteps == 0:
            return (order[:0].astype(np.int64),) * 2 if want_spans else (None, None)
        Bmax = max(st['B'] for st in steps)
        if getattr(self, '_feed_owner', None) is not dataset or self._feeds['shape'][0] < Bmax:
            self._feeds = dataset.feed_buffers(Bmax, min_chars=min_chars)
            self._feed_owner = dataset
        Bm, Tm, Lm, Cm = self._feeds['shape']
        self.reserve(Bm, Tm, Lm, Cm)
        # the ids this rank trains on, step after step, uploaded ONCE; the epoch's position is a device cursor that the step's own
        # last launch advances (hual_adamw_clip_step_loop), and the step's first launch is its batch assembly at that position
        # (hual_assemble_batch_cursor): a step of a known padded shape is ONE graph launch and nothing else - no eager launch between
        # two step graphs (each cost ~10 us of idle device on either side), nothing uploaded, nothing fetched
        mine = np.concatenate([st['ids'][rank * st['B']:(rank + 1) * st['B']] for st in steps])
        lp = self._ensure_loop(dataset, len(mine), nsteps)
        lp['ids'][:len(mine)].copy_(torch.from_numpy(np.ascontiguousarray(mine)))
        lp['cursor'].zero_()
        self._loop = lp
        t_host = time.perf_counter()
        try:
            for st in steps:
                T, L, C = st['shape']
                lp['views'] = dataset.feed_views(st['B'], T, L, C, self._feeds)
                self._host_denom = (st['frames'] + 1e-12) / world if self.dp else None
                self.set_batch_device(lp['views'])
                self.step(lr=lr, drop_rate=drop_rate)
        finally:
            self._host_denom = None
            self._loop = None
        self.stats['host_enqueue_s'] = self.stats.get('host_enqueue_s', 0.0) + (time.perf_counter() - t_host)
        if not want_spans:
            return None, None
        Bcap = self.spans.shape[1]
        bank = lp['bank'][:nsteps * 2 * Bcap].view(nsteps, 2, Bcap)
        host = hdist.allgather_cat(bank).cpu().numpy() if world > 1 else bank.cpu().numpy()[None]      # the epoch's only device -> host transfer
        st_, en_ = [], []
        for i, stp in enumerate(steps):
            for r in range(world):
                st_.append(host[r, i, 0, :stp['B']])
                en_.append(host[r, i, 1, :stp['B']])
        return np.concatenate(st_), np.concatenate(en_)

    def _ensure_loop(self, dataset, n_ids, nsteps):
        """device state of the epoch loop: the rank's id list, the cursor {ids consumed, bank words written} and the span bank
        [steps, 2, B capacity].  Allocated once per (dataset size, fetch capacity) - the step graphs hold these addresses."""
        lp = getattr(self, '_loop_state', None)
        Bcap = self.spans.shape[1]
        if lp is None or lp['ds'] is not dataset or lp['ids'].numel() < n_ids or lp['bank'].numel() < nsteps * 2 * Bcap or lp['Bcap'] != Bcap:
            dev = self.m.device
            # capacity for ANY epoch over this set (a batch size of one: as many steps as samples; 34 MB for ActivityNet at B 64), so
            # that a later epoch with another batch size or order never moves the buffers
            cap = max(n_ids, nsteps, len(dataset), 1)
            lp = dict(ds=dataset, Bcap=Bcap, ids=torch.zeros(cap, dtype=torch.int32, device=dev),
                      cursor=torch.zeros(2, dtype=torch.int64, device=dev),
                      bank=torch.zeros(cap * 2 * Bcap, dtype=torch.int64, device=dev), views=None)
            self._loop_state = lp
        return lp

    def last_loss(self):
        """total loss of the last step (device sync).  DP: local loc/match terms + the global alignment loss."""
        l = self.loss_terms[0]
        if self.dp:
            l = l + self.align_loss[0]
        return l

"""Active-learning round pieces around the SeqPAN hot path, MI355X side.

    reference                                                    here
    ---------------------------------------------------------    -------------------------------------------------
    runner_utils.eval_test_save (utils/runner_utils.py:69-110)   infer_trainset(model, batches): ONE deterministic
      five sess.run per batch + results/<task>/<suffix>.pkl        forward (all five fetches) + two stochastic forwards
    update_label.main (update_label.py:173-208)                  update_labels(data_old, data_gt, last_prop, coff):
      python loop over samples, sorted() inside the loop           two launches over the whole training set
    update_label.get_coff / F_renew (update_label.py:11-37,212)  get_coff / F_RENEW

The scoring and the pseudo-label re-derivation run in libhual_seqpan.so (csrc/al.hip, through hual_al_score /
hual_al_renew); ranking, ground-truth lookup (the "annotator") and time<->index conversion are a few numpy lines on the
host.  There is no CPU fallback: without the library or a GPU this module raises.

Reference quirk kept selectable: eval-mode get_feed_dict (runner_utils.py:61-65) drops drop_rate, so the
reference's two "dropout 0.5" passes run WITHOUT dropout and prop_logits1 == prop_logits2 == prop_logits;
mc_dropout=None reproduces that, mc_dropout=0.5 is what the code intends.
"""
import ctypes
import math

import numpy as np
import torch

from . import lib

# update_label.py:11-37 (index = active-learning round I; entry 0 unused)
F_RENEW = {
    'charades': {'pos': {'old': [None, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0], 'model': [None, 0.8, 0.8, 0.8, 0.8, 0.8, 0.8],
                         'distance': [None, 4.0, 0.2, 0.2, 0.2, 0.2, 0.2]},
                 'neg': {'old': [None, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0], 'model': [None, 2.4, 0.2, 0.2, 0.2, 0.2, 0.2],
                         'distance': [None, 2.0, 0.2, 0.2, 0.2, 0.2, 0.2]},
                 'uncert': [None, 0.25, 0.25, 0.25, 0.25, 0.25, 0.25]},
    'anet': {'pos': {'old': [None, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0], 'model': [None, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0],
                     'distance': [None, 2.0, 1.8, 1.6, 1.5, 1.5, 1.5]},
             'neg': {'old': [None, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0], 'model': [None, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0],
                     'distance': [None, 2.0, 1.8, 1.6, 1.5, 1.5, 1.5]},
             'uncert': [None, 0.25, 0.25, 0.25, 0.25, 0.25, 0.25]},
}


def get_coff(task, I):
    """(pos.distance, pos.model, pos.old, neg.distance, neg.model, neg.old, uncert) of round I (update_label.py:212-218)"""
    t = F_RENEW[task]
    return (t['pos']['distance'][I], t['pos']['model'][I], t['pos']['old'][I],
            t['neg']['distance'][I], t['neg']['model'][I], t['neg']['old'][I], t['uncert'][I])


# ---------------------------------------------------------------- infer_trainset ----------------
def infer_trainset(model, batches, mc_dropout=None, batch_ids=None, rng=None):
    """eval_test_save (runner_utils.py:69-110) without the file write: returns (records, ious).

    batches: iterable of (raw_records, video, video_seq_len, word_ids, char_ids) as TestLoader.test_iter yields them
    (data_loader.py:131-143).  Each record of the result has the keys of runner_utils.py:90-100; logits are the raw
    [T_b] rows of the batch (unmasked beyond v_len), m_score is [T_b, 4].
    batch_ids (with mc_dropout): the position of each yielded batch in the whole pass - the two stochastic forwards of batch i use the
    Philox offsets base + 2 i and base + 2 i + 1, whichever rank runs the batch and whatever ran before it (infer_trainset_sharded);
    rng = (seed, base) of that stream (default: the model's own state).
    """
    from . import data
    records, ious = [], []
    batch_ids = iter(batch_ids) if batch_ids is not None else None
    rng_base = rng_seed = None
    if batch_ids is not None and mc_dropout is not None:
        if rng is not None:
            rng_seed, rng_base = int(rng[0]), int(rng[1])
        else:
            st = model.rng_state.cpu().numpy().view(np.uint32)
            rng_seed, rng_base = int(st[0]) | (int(st[1]) << 32), int(st[2])

    def enqueue(batch):
        """all forwards of a batch (one deterministic + two stochastic), nothing fetched: the device runs them while the host
        writes the records of the batch before"""
        raw, video, lens, word_ids, char_ids = batch
        o = model.forward(video, lens, word_ids, char_ids, drop_rate=0.0)
        dev = [o['start_logits'], o['end_logits'], o['match_scores'], o['start_index'], o['end_index']]
        if mc_dropout is not None:
            if rng_base is not None:
                model.set_rng(rng_seed, rng_base + 2 * int(next(batch_ids)))
            o1 = model.forward(video, lens, word_ids, char_ids, drop_rate=mc_dropout)
            model.rng_state[2] += 1                              # a fresh Philox offset for the second stochastic pass
            o2 = model.forward(video, lens, word_ids, char_ids, drop_rate=mc_dropout)
            model.rng_state[2] += 1
            dev += [o1['start_logits'], o1['end_logits'], o2['start_logits'], o2['end_logits']]
        # device -> pinned host, asynchronously on the compute stream; the event marks their arrival
        host = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in dev]
        for h, t in zip(host, dev):
            h.copy_(t, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return raw, host, ev

    def emit(job):
        raw, host, ev = job
        ev.synchronize()
        # (copies: the records outlive the loop, the pinned staging buffers should not)
        s0, e0, ms, si, ei = (h.numpy().copy() for h in host[:5])
        if mc_dropout is None:
            s1, e1, s2, e2 = s0, e0, s0, e0                     # as written: drop_rate never reaches the graph (F8)
        else:
            s1, e1, s2, e2 = (h.numpy().copy() for h in host[5:9])
        ious.extend(ious_of_spans(raw, si, ei))
        for i, r in enumerate(raw):
            records.append({'vid': r['vid'], 'duration': r['duration'], 'psuedo_idx': [r['s_ind'], r['e_ind']],
                            'sentence': ' '.join(r['words']), 'v_len': int(r['v_len']),
                            'prop_idx': [int(si[i]), int(ei[i])], 'prop_logits': [s0[i], e0[i]],
                            'prop_logits1': [s1[i], e1[i]], 'prop_logits2': [s2[i], e2[i]], 'm_score': ms[i]})

    pending = None
    for batch in batches:
        job = enqueue(batch)                                     # batch n is on the device ...
        if pending is not None:
            emit(pending)                                        # ... while the records of batch n - 1 are written
        pending = job
    if pending is not None:
        emit(pending)
    return records, ious


def infer_trainset_sharded(model, dataset, batch_size, mc_dropout=None, min_chars=4):
    """infer_trainset over a DeviceDataset in the reference's order (TrainNoSuffleLoader.test_iter, data_loader.py:167-206), the
    batches dealt round-robin to the ranks of the process group (batch i -> rank i % world; every rank holds the whole set):
    rank 0 returns (records, ious) of the WHOLE set in sample order, the other ranks (None, None).  One rank: the plain pass.
    The batches are the single-process ones (own padded shape each), so the records equal a single-process pass; the Philox offsets
    of the stochastic forwards depend on the batch index only."""
    from . import dist as hdist
    world, rank = hdist.world_size(), hdist.rank()
    N = len(dataset)
    los = list(range(0, N, batch_size))
    own = [i for i in range(len(los)) if i % world == rank]

    def batches():
        for i in own:
            sel = np.arange(los[i], min(N, los[i] + batch_size))
            f = dataset.assemble(sel, out=None, labels=False, min_chars=min_chars)
            yield [dataset.records[k] for k in sel], f['video'], f['video_seq_len'], f['word_ids'], f['char_ids']
    st0 = model.rng_state.cpu().numpy().view(np.uint32).copy()
    own_rng = (int(st0[0]) | (int(st0[1]) << 32), int(st0[2]))
    rng = hdist.broadcast_object(own_rng)                       # rank 0's stream for the stochastic passes: the records do not depend on `world`
    records, ious = infer_trainset(model, batches(), mc_dropout=mc_dropout, batch_ids=own, rng=rng)
    # every rank returns to its OWN dropout stream, behind the whole pass
    model.set_rng(own_rng[0], own_rng[1] + (2 * len(los) if mc_dropout is not None else 0))
    parts = hdist.gather_objects((own, records, ious))
    if parts is None:
        return None, None
    out_r, out_i = [None] * N, [None] * N
    for own_r, recs_r, ious_r in parts:
        k = 0
        for i in own_r:
            n = min(N, los[i] + batch_size) - los[i]
            out_r[los[i]:los[i] + n] = recs_r[k:k + n]
            out_i[los[i]:los[i] + n] = ious_r[k:k + n]
            k += n
    return out_r, out_i


def calculate_iou(i0, i1):
    """runner_utils.py:34-38"""
    union = (min(i0[0], i1[0]), max(i0[1], i1[1]))
    inter = (max(i0[0], i1[0]), min(i0[1], i1[1]))
    return max(0.0, 1.0 * (inter[1] - inter[0]) / (union[1] - union[0]))


def ious_of_spans(records, sidx, eidx):
    """IoU of the predicted spans against the records' own (s_ind, e_ind), both through index_to_time (data_utils.py:121-128) and
    calculate_iou (runner_utils.py:34-38): the per-sample loop of runner_utils.py:149-156 for a whole list at once, in the reference's
    own arithmetic (float32 unit grid times, float32 ratio).  Equal to the scalar functions element for element
    (tests/test_data_golden.py)."""
    n = len(records)
    if n == 0:
        return []
    f32 = np.float32
    vl = np.array([r['v_len'] for r in records], dtype=f32)
    du = np.array([float(r['duration']) for r in records], dtype=f32)
    gs = np.array([r['s_ind'] for r in records], dtype=np.int64)
    ge = np.array([r['e_ind'] for r in records], dtype=np.int64)
    ps = np.asarray(sidx, dtype=np.int64)[:n]
    pe = np.asarray(eidx, dtype=np.int64)[:n]

    def t0(i):      # s_times[i] = float32(i) * duration / num_units
        return i.astype(f32) * du / vl

    def t1(i):      # e_times[i] = float32(i + 1) * duration / num_units
        return (i + 1).astype(f32) * du / vl
    st, et, g0, g1 = t0(ps), t1(pe), t0(gs), t1(ge)
    union = np.maximum(et, g1) - np.minimum(st, g0)
    inter = np.minimum(et, g1) - np.maximum(st, g0)
    with np.errstate(divide='ignore', invalid='ignore'):
        iou = inter / union
    return np.where(iou > 0, iou, f32(0.0)).tolist()


def iou_metrics(ious):
    """R@1 IoU={0.3,0.5,0.7} and mIoU in percent (runner_utils.py:25-31,106-109)"""
    a = np.asarray(ious, dtype=np.float64)
    return tuple(float(np.mean(a >= t) * 100.0) for t in (0.3, 0.5, 0.7)) + (float(np.mean(a) * 100.0),)


# ---------------------------------------------------------------- label update -------------------
def _round_half_even_index(t, duration, vlen):
    """time_to_index_v2 (update_label.py:41-48): python round() of t / duration * (vlen - 1)"""
    return round(t / duration * (vlen - 1))


class LabelUpdater:
    """Device-side state of one update_label round: the logits of the results pkl as [N, ld] matrices, the active
    points as CSR.  score() and renew() are one launch each."""

    def __init__(self, last_prop, aps, device='cuda:0'):
        if not torch.cuda.is_available():
            raise lib.HualError('LabelUpdater needs a GPU: the HIP path has no CPU fallback')
        self._lib = lib.load()
        self.dev = torch.device(device)
        N = len(last_prop)
        tlen = np.array([len(p['prop_logits'][0]) for p in last_prop], dtype=np.int32)
        vlen = np.array([p['v_len'] for p in last_prop], dtype=np.int32)
        ld = int(tlen.max())
        if int(tlen.min()) < 2 or ld > 1024 or (vlen < 1).any() or (vlen > tlen).any():
            raise ValueError('need 2 <= len(logits) <= 1024 and 1 <= v_len <= len(logits)')
        lg = np.zeros((6, N, ld), dtype=np.float32)
        for n, p in enumerate(last_prop):
            for k, key in enumerate(('prop_logits', 'prop_logits1', 'prop_logits2')):
                lg[2 * k, n, :tlen[n]] = p[key][0]
                lg[2 * k + 1, n, :tlen[n]] = p[key][1]
        self.N, self.ld = N, ld
        self.tlen_h, self.vlen_h = tlen, vlen
        self.logits = torch.from_numpy(lg).to(self.dev)
        self.tlen = torch.from_numpy(tlen).to(self.dev)
        self.vlen = torch.from_numpy(vlen).to(self.dev)
        self.sprob = torch.zeros(N, ld, device=self.dev)
        self.eprob = torch.zeros(N, ld, device=self.dev)
        self.uncert_frame = torch.zeros(N, ld, device=self.dev, dtype=torch.float64)
        self.uncert_video = torch.zeros(N, device=self.dev)
        self.observe = torch.zeros(N, device=self.dev, dtype=torch.int32)
        self.set_active_points(aps)

    def set_active_points(self, aps):
        """aps: per sample (list of (frame, is_pos)) in annotation order"""
        off = np.zeros(self.N + 1, dtype=np.int32)
        off[1:] = np.cumsum([len(a) for a in aps])
        idx = np.array([f for a in aps for f, _ in a] + [0], dtype=np.int32)      # never empty: a valid pointer
        pos = np.array([1 if p else 0 for a in aps for _, p in a] + [0], dtype=np.int8)
        self.ap_off = torch.from_numpy(off).to(self.dev)
        self.ap_idx = torch.from_numpy(idx).to(self.dev)
        self.ap_pos = torch.from_numpy(pos).to(self.dev)
        p = lib.ptr
        self.set = lib.hual_al_set(self.N, self.ld, p(self.vlen).value, p(self.tlen).value, p(self.ap_off).value,
                                   p(self.ap_idx).value, p(self.ap_pos).value)

    def score(self, coff_uncert):
        p, lg = lib.ptr, self.logits
        lib.check(self._lib.hual_al_score(ctypes.byref(self.set), p(lg[0]), p(lg[1]), p(lg[2]), p(lg[3]), p(lg[4]), p(lg[5]),
                                          float(coff_uncert), p(self.sprob), p(self.eprob), p(self.uncert_frame),
                                          p(self.uncert_video), p(self.observe), lib.stream_ptr()))

    def renew(self, sel, old_idx, coff):
        """sel: sample ids (numpy); old_idx: int [N,2]; returns new_idx int32 [N,2] (valid for the selected rows)"""
        sel_d = torch.from_numpy(np.ascontiguousarray(sel, dtype=np.int32)).to(self.dev)
        old_d = torch.from_numpy(np.ascontiguousarray(old_idx, dtype=np.int32)).to(self.dev)
        new_d = torch.full((self.N, 2), -1, device=self.dev, dtype=torch.int32)
        c6 = (ctypes.c_double * 6)(*[float(x) for x in coff[:6]])
        lib.check(self._lib.hual_al_renew(ctypes.byref(self.set), lib.ptr(sel_d), int(len(sel)), lib.ptr(self.sprob),
                                          lib.ptr(self.eprob), lib.ptr(old_d), c6, lib.ptr(new_d), lib.stream_ptr()))
        return new_d.cpu().numpy()


def update_labels(data_old, data_gt, last_prop, coff, device='cuda:0', return_debug=False):
    """update_label.main (update_label.py:173-208) without the file IO.

    data_old / data_gt: lists [vid, duration, [start_time, end_time], sentence(, active points)] as in
    data/<task>_re<I>/train.json; last_prop: the records of results/<task>/re<I-1>.pkl; coff: get_coff(task, I).
    Mutates and returns data_old exactly as the reference writes it to data/<task>_re<I>/train.json.
    """
    if len(data_old[0]) == 4:
        for r in data_old:
            r.append({'pos_idx': [], 'neg_idx': []})
    N = len(data_old)
    for i in range(N):
        assert data_old[i][0] == last_prop[i]['vid'] and data_old[i][0] == data_gt[i][0]
    # active points in one list per sample; the reference keeps two lists and only ever asks for min / max / membership
    # of each, so the relative order between the two kinds does not matter
    aps = [[(f, True) for f in r[4]['pos_idx']] + [(f, False) for f in r[4]['neg_idx']] for r in data_old]
    up = LabelUpdater(last_prop, aps, device=device)
    up.score(coff[6])
    uv = up.uncert_video.cpu().numpy()
    observe = up.observe.cpu().numpy()
    order = np.argsort(uv, kind='stable')                       # sorted(key=uncert_video), ties in sample order
    sel = order[:math.ceil(N / 2)]
    vlen = up.vlen_h
    gt_idx = np.array([[_round_half_even_index(t, data_gt[i][1], int(vlen[i])) for t in data_gt[i][2]] for i in range(N)])
    old_idx = np.array([[_round_half_even_index(t, data_old[i][1], int(vlen[i])) for t in data_old[i][2]] for i in range(N)])
    # append_AP (utils_hual.py:133-139): the annotator answers "is the observed frame inside the ground-truth span?"
    for i in sel:
        p = int(observe[i])
        is_pos = gt_idx[i, 0] <= p <= gt_idx[i, 1]
        data_old[i][4]['pos_idx' if is_pos else 'neg_idx'].append(p)
        aps[i].append((p, bool(is_pos)))
    up.set_active_points(aps)
    new_idx = up.renew(sel, old_idx, coff)
    for i in sel:
        dur, vl = data_old[i][1], int(vlen[i])
        data_old[i][2] = [round(int(t) / (vl - 1) * dur, 2) for t in new_idx[i]]          # index_to_time, update_label.py:50-57
    if return_debug:
        return data_old, dict(order=order, uncert_video=uv, observe=observe, uncert_frame=up.uncert_frame.cpu().numpy(),
                              sprob=up.sprob.cpu().numpy(), eprob=up.eprob.cpu().numpy(), new_idx=new_idx, gt_idx=gt_idx,
                              old_idx=old_idx, updater=up)
    return data_old


# ---------------------------------------------------------------- one whole round ----------------
def labels_from_times(data, vlens):
    """pseudo-label frame indices of train.json entries: dataset_gen's time_to_index (utils/data_gen.py:98-125 ->
    data_utils.py:110-118) on every [vid, duration, [start, end], ...] record"""
    from . import data as hdata
    s, e = [], []
    for r, n in zip(data, vlens):
        a, b = hdata.time_to_index(r[2][0], r[2][1], int(n), r[1])
        s.append(a)
        e.append(b)
    return np.array(s, dtype=np.int32), np.array(e, dtype=np.int32)


def run_round(model, dataset, data_old, data_gt, last_prop, task, I, epochs, batch_size, lr, drop_rate, mc_dropout=0.5,
              shuffle_seed=0, log=None, trainer=None):
    """One active-learning round of run_charades.py:9-41 on device-resident data:
         update_label.py <task> I   ->  main.py --mode train (epochs)   ->  main.py --mode infer_trainset
    dataset: DeviceDataset over the training records in the SAME order as data_old / data_gt / last_prop.
    Data parallel (torch.distributed initialised, one process per GPU): every rank calls this with the same dataset and train lists;
    `last_prop` is needed on rank 0 only (the other ranks may pass None).  Rank 0 renews the labels and broadcasts the new train
    list, the epochs run data parallel (`batch_size` clips per rank, Trainer.run_epoch), infer_trainset is sharded by batch and its
    records are gathered on rank 0.
    Returns (new train list, new results records - rank 0 only, else None -, metrics dict)."""
    import time
    from . import dist as hdist
    from .train import Trainer
    world, rank = hdist.world_size(), hdist.rank()
    t0 = time.perf_counter()
    new_data = update_labels(data_old, data_gt, last_prop, get_coff(task, I), device=model.device) if rank == 0 else None
    new_data = hdist.broadcast_object(new_data)
    torch.cuda.synchronize()
    t1a = time.perf_counter()
    # the pseudo-label frame indices of the new train list: the reference regenerates its dataset cache for this (dataset_gen,
    # utils/data_gen.py:98-125 - one time_to_index per record, a T x T overlap table each), outside its training loop
    s_ind, e_ind = labels_from_times(new_data, dataset.vlen_h)
    dataset.set_labels(s_ind, e_ind)
    for r, a, b in zip(dataset.records, s_ind, e_ind):
        r['s_ind'], r['e_ind'] = int(a), int(b)
    t1 = time.perf_counter()
    N = len(dataset)
    tr = trainer if trainer is not None else Trainer(model, world=world, use_graph=True)
    rng = np.random.default_rng(shuffle_seed)                   # the same permutations on every rank
    steps = 0
    for ep in range(epochs):
        cur_lr = lr * (1.0 - ep / epochs)                       # main.py:61
        order = rng.permutation(N)                              # random.shuffle(self.dataset), data_loader.py:24
        tr.run_epoch(dataset, order, batch_size, lr=cur_lr, drop_rate=drop_rate, min_chars=4)      # spans fetched: train_epoch's IoU log
        steps += (N + batch_size * world - 1) // (batch_size * world)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    records, ious = infer_trainset_sharded(model, dataset, batch_size, mc_dropout=mc_dropout, min_chars=4)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    met = hdist.broadcast_object(iou_metrics(ious) if rank == 0 else None)
    r3, r5, r7, mi = met
    m = dict(update_s=t1a - t0, relabel_s=t1 - t1a, train_s=t2 - t1, infer_s=t3 - t2, train_steps=steps, clips_per_s=N * epochs / max(t2 - t1, 1e-9),
             step_launch_modes=dict(tr.stats), world=world,
             r1i3=r3, r1i5=r5, r1i7=r7, miou=mi)
    if log and rank == 0:
        log('round %d: update_label %.3f s | train %d steps %.3f s (%.0f clips/s) | infer_trainset %.3f s | pseudo-label '
            'R1@0.5 %.2f mIoU %.2f' % (I, m['update_s'], steps, m['train_s'], m['clips_per_s'], m['infer_s'], r5, mi))
    return new_data, records, m

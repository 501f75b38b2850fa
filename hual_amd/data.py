"""Host-side batch assembly for the SeqPAN hot path (numpy; feeds the HIP forward/backward).

Mirrors the interface of the reference's loaders so the runner code reads the same:
  TrainLoader.process_batch   /root/reference/utils/data_loader.py:30-98
  TestLoader.process_batch    /root/reference/utils/data_loader.py:145-164
  pad_seq / pad_char_seq / pad_video_seq   /root/reference/utils/data_utils.py:130-172
  time_to_index / index_to_time / visual_feature_sampling   /root/reference/utils/data_utils.py:70-128
The label synthesis is written array-at-a-time (one pass per label kind) instead of the reference's
per-sample Python loop; tests/test_data_golden.py pins it against fixtures produced by the reference code.
"""
import numpy as np


# ---------------------------------------------------------------- padding ----------------
def pad_word_ids(seqs):
    """list of list[int] -> int32 [B, Lmax], 0 = PAD (data_utils.py:130-141)."""
    L = max(len(s) for s in seqs)
    out = np.zeros((len(seqs), L), dtype=np.int32)
    for b, s in enumerate(seqs):
        out[b, :len(s)] = s
    return out


def pad_char_ids(seqs):
    """list of list of list[int] -> int32 [B, Lmax, Cmax] (data_utils.py:143-155)."""
    L = max(len(s) for s in seqs)
    C = max(max(len(w) for w in s) for s in seqs)
    out = np.zeros((len(seqs), L, C), dtype=np.int32)
    for b, s in enumerate(seqs):
        for l, w in enumerate(s):
            out[b, l, :len(w)] = w
    return out


def pad_video(feats):
    """list of [n_i, V] -> float32 [B, Tmax, V] zero padded, int32 lens (data_utils.py:158-172)."""
    lens = np.array([f.shape[0] for f in feats], dtype=np.int32)
    T = int(lens.max())
    out = np.zeros((len(feats), T, feats[0].shape[1]), dtype=np.float32)
    for b, f in enumerate(feats):
        out[b, :f.shape[0]] = f
    return out, lens


# ---------------------------------------------------------------- labels -----------------
def make_labels(s_inds, e_inds, lens, max_len=None, ext_len=2):
    """Soft start/end labels, 4-class match labels and inner labels (data_loader.py:49-94).

    Returns (y1 f32[B,T], y2 f32[B,T], match_labels i32[B,T], inner_labels i32[B,T]).
    """
    s_inds = np.asarray(s_inds, dtype=np.int64)
    e_inds = np.asarray(e_inds, dtype=np.int64)
    lens = np.asarray(lens, dtype=np.int64)
    B = len(lens)
    T = int(lens.max()) if max_len is None else int(max_len)
    pos = np.arange(T)[None, :]
    valid = pos < lens[:, None]

    def soft(idx):
        lab = np.where(valid, np.float32(1e-10), np.float32(0.0)).astype(np.float32)
        # the reference computes y in python double precision and stores it into a float32 array
        y = ((1.0 - lens.astype(np.float64) * 1e-10 - 0.5) / 2.0)
        rows = np.arange(B)
        lab[rows, idx] = lab[rows, idx] + np.float32(0.5)
        for b in range(B):
            i, n = int(idx[b]), int(lens[b])
            if i > 0:
                lab[b, i - 1] = y[b]
            else:
                lab[b, i] = lab[b, i] + y[b]
            if i < n - 1:
                lab[b, i + 1] = y[b]
            else:
                lab[b, i] = lab[b, i] + y[b]
        return lab

    y1 = soft(s_inds)
    y2 = soft(e_inds)
    st_l = np.maximum(0, s_inds - ext_len)
    st_r = np.minimum(s_inds + ext_len, lens - 1)
    et_l = np.maximum(0, e_inds - ext_len)
    et_r = np.minimum(e_inds + ext_len, lens - 1)
    st_r = np.where(st_r >= et_l, np.maximum(s_inds, et_l - 1), st_r)
    match = np.zeros((B, T), dtype=np.int32)
    inner = np.zeros((B, T), dtype=np.int32)
    # same write order as the reference: B-M, then I-M, then E-M (later writes win)
    match[(pos >= st_l[:, None]) & (pos <= st_r[:, None])] = 1
    mid = (pos > st_r[:, None]) & (pos < et_l[:, None])
    match[mid] = 2
    inner[mid] = 1
    match[(pos >= et_l[:, None]) & (pos <= et_r[:, None])] = 3
    return y1, y2, match, inner


class Batch(dict):
    __getattr__ = dict.__getitem__


def process_train_batch(records, visual_feats):
    """records: dicts with vid, w_ids, c_ids, s_ind, e_ind  (data_loader.py:30-98)."""
    video, lens = pad_video([visual_feats[r['vid']] for r in records])
    y1, y2, match, inner = make_labels([r['s_ind'] for r in records], [r['e_ind'] for r in records], lens)
    return Batch(video=video, video_seq_len=lens, word_ids=pad_word_ids([r['w_ids'] for r in records]),
                 char_ids=pad_char_ids([r['c_ids'] for r in records]), y1=y1, y2=y2, match_labels=match,
                 inner_labels=inner)


def pad_batch_to(batch, T, L, C):
    """A processed batch zero-padded to (T, L, C) >= its own maxima, labels re-derived for the padded length: what a data-parallel
    shard looks like when it is padded to the GLOBAL batch's shape (hual_amd/dist.py shard_plan) - the rows a longer clip / query /
    word of ANOTHER shard forces onto it are exactly the rows the single-process batch holds for these samples."""
    B, T0, V = batch['video'].shape
    L0, C0 = batch['word_ids'].shape[1], batch['char_ids'].shape[2]
    assert T >= T0 and L >= L0 and C >= C0
    out = Batch(batch)
    out['video'] = np.zeros((B, T, V), dtype=np.float32)
    out['video'][:, :T0] = batch['video']
    out['word_ids'] = np.zeros((B, L), dtype=np.int32)
    out['word_ids'][:, :L0] = batch['word_ids']
    out['char_ids'] = np.zeros((B, L, C), dtype=np.int32)
    out['char_ids'][:, :L0, :C0] = batch['char_ids']
    for k in ('y1', 'y2', 'match_labels', 'inner_labels'):
        if k in batch:
            a = np.zeros((B, T), dtype=batch[k].dtype)
            a[:, :T0] = batch[k]
            out[k] = a
    return out


def process_test_batch(records, visual_feats):
    """data_loader.py:145-164."""
    video, lens = pad_video([visual_feats[r['vid']] for r in records])
    return Batch(video=video, video_seq_len=lens, word_ids=pad_word_ids([r['w_ids'] for r in records]),
                 char_ids=pad_char_ids([r['c_ids'] for r in records]))


# ---------------------------------------------------------------- time <-> index ---------
def time_to_index(start_time, end_time, num_units, duration):
    """Best-IoU (start,end) unit indices (data_utils.py:110-118)."""
    duration, start_time, end_time = float(duration), float(start_time), float(end_time)
    s_times = np.arange(0, num_units).astype(np.float32) / float(num_units) * duration
    e_times = np.arange(1, num_units + 1).astype(np.float32) / float(num_units) * duration
    # candidates are python floats in the reference (tolist) -> float64 arithmetic on float32 grid values
    s = s_times.astype(np.float64)[:, None]
    e = e_times.astype(np.float64)[None, :]
    inter = np.maximum(0.0, np.minimum(e, end_time) - np.maximum(s, start_time))
    union = np.maximum(1e-12, np.maximum(e, end_time) - np.minimum(s, start_time))
    ov = inter / union
    k = int(np.argmax(ov))
    return k // num_units, k % num_units


def index_to_time(st, num_units, duration):
    """data_utils.py:121-128."""
    duration = float(duration)   # python float keeps the float32 grid (NEP 50), as in the reference
    s_times = np.arange(0, num_units).astype(np.float32) * duration / float(num_units)
    e_times = np.arange(1, num_units + 1).astype(np.float32) * duration / float(num_units)
    return s_times[st[0]], e_times[st[1]]


def visual_feature_sampling(feat, max_num_clips):
    """Mean-pool a long video down to max_num_clips (data_utils.py:70-85)."""
    n = feat.shape[0]
    if n <= max_num_clips:
        return feat
    idxs = np.round(np.arange(0, max_num_clips + 1, 1.0) / max_num_clips * n).astype(np.int32)
    idxs[idxs > n - 1] = n - 1
    out = np.empty((max_num_clips, feat.shape[1]), dtype=feat.dtype)
    for i in range(max_num_clips):
        a, b = idxs[i], idxs[i + 1]
        out[i] = feat[a:b].mean(axis=0) if a < b else feat[a]
    return out

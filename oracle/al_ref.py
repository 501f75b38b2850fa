"""CPU oracle for the active-learning label update (TEST INFRASTRUCTURE - only tests/, smoke() and bench.py's
cpu_baseline may import this; the product path is hual_amd/al.py + csrc/al.hip).

numpy restatement of the reference's per-round label update, one function per reference function:

    /root/reference/utils/utils_hual.py   fill_isactivate :37-59, get_segment :63-76, center_width_gauss :79-89,
                                          get_distance_score :92-103, get_distance_score_shift :107-124, sigmoid :128,
                                          append_AP :133-139, get_uncert_model :144-161, calculate_iou :13-19
    /root/reference/update_label.py       time_to_index_v2 :41-48, index_to_time :50-57, mask_activepoints :62-83,
                                          renew_label :85-123, get_uncert_rank :125-169, main :173-208, get_coff :212-218,
                                          F_renew :11-37
    /root/reference/utils/runner_utils.py calculate_iou :34-38 (no zero-union guard), calculate_iou_accuracy :25-31

Pinned by tests/golden/al.npz (scripts/gen_golden.py imports the reference's own update_label.py / utils_hual.py with
stand-ins for the absent easydict / omegaconf modules).  dtype notes matter for the argmax results and are kept: the
gaussians are float32 (np.linspace(dtype=float32) with python-scalar operands), the score mixes are float64 (they start
from np.zeros), the model uncertainty is float32 (torch.sigmoid of float32 logits).
"""
import math

import numpy as np

# update_label.py:11-37 (index = active-learning round I; entry 0 is unused)
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
    """update_label.py:212-218 -> flat tuple (pos.distance, pos.model, pos.old, neg.distance, neg.model, neg.old, uncert)"""
    t = F_RENEW[task]
    return (t['pos']['distance'][I], t['pos']['model'][I], t['pos']['old'][I],
            t['neg']['distance'][I], t['neg']['model'][I], t['neg']['old'][I], t['uncert'][I])


def sigmoid(x):
    """utils_hual.py:128-129 (stays in the dtype of x: float32 for the logits of the results pkl)"""
    return 1 / (1 + np.exp(-x))


def time_to_index_v2(t, duration, vlen):
    """update_label.py:41-48 (python round = half to even)"""
    if isinstance(t, (list, tuple)):
        return [time_to_index_v2(x, duration, vlen) for x in t]
    return round(t / duration * (vlen - 1))


def index_to_time(t, duration, vlen):
    """update_label.py:50-57"""
    if isinstance(t, (list, tuple)):
        return [index_to_time(x, duration, vlen) for x in t]
    return round(t / (vlen - 1) * duration, 2)


def calculate_iou(i0, i1):
    """utils_hual.py:13-19"""
    lo, hi = min(i0[0], i1[0]), max(i0[1], i1[1])
    if hi - lo == 0.0:
        return 0.0
    return max(0.0, 1.0 * (min(i0[1], i1[1]) - max(i0[0], i1[0])) / (hi - lo))


def fill_isactivate(pos_idx, neg_idx, vlen, max_vlen):
    """utils_hual.py:37-59: 1 inside the hull of the positive points, -1 outside the nearest negatives (or at the
    negative points themselves when there is no positive), 0 = unknown, -100 = padding."""
    act = np.zeros(max_vlen)
    if len(pos_idx) > 0:
        lo, hi = min(pos_idx), max(pos_idx)
        act[lo:hi + 1] = 1
        left = [i for i in neg_idx if i < lo]
        right = [i for i in neg_idx if i > hi]
        if left:
            act[:max(left) + 1] = -1
        if right:
            act[min(right):] = -1
    else:
        for i in neg_idx:
            act[i] = -1
    act[vlen:] = -100
    return act


def get_segment(isactive):
    """utils_hual.py:63-76: maximal runs of 0 as [first, last]"""
    segs, start = [], None
    for i, v in enumerate(list(isactive) + [-100]):
        if v == 0 and start is None:
            start = i
        elif v != 0 and start is not None:
            segs.append([start, i - 1])
            start = None
    return segs


def center_width_gauss(center, width, vlen, max_vlen):
    """utils_hual.py:79-89.  float32 throughout (python scalars are weak operands); peak normalised to width/vlen."""
    x = np.linspace(-1, 1, num=max_vlen, dtype=np.float32)
    sig = vlen / max_vlen
    sig *= width / vlen * 0.4
    u = (center / (max_vlen - 1)) * 2 - 1
    w = np.exp(-(x - u) ** 2 / (2 * sig ** 2)) / (math.sqrt(2 * math.pi) * sig)
    w /= np.max(w)
    w *= width / vlen
    w[vlen:] = 0.0
    return w


def _distance(pos_idx, neg_idx, vlen, max_vlen, shift):
    out = np.zeros(max_vlen)
    for a, b in get_segment(fill_isactivate(pos_idx, neg_idx, vlen, max_vlen)):
        width = b - a + 1
        center = (b - a) / 2 + a + width * shift / 2
        out[a:b + 1] = center_width_gauss(center, width, vlen, max_vlen)[a:b + 1]
    return out


def get_distance_score(pos_idx, neg_idx, vlen, max_vlen):
    """utils_hual.py:92-103: inside every unknown run, a gaussian bump centred on the run (far from known points = high)"""
    return _distance(pos_idx, neg_idx, vlen, max_vlen, 0.0)


def get_distance_score_shift(pos_idx, neg_idx, vlen, max_vlen, shift):
    """utils_hual.py:107-124: start score shifted by -width*shift/2, end score by +width*shift/2"""
    return _distance(pos_idx, neg_idx, vlen, max_vlen, -shift), _distance(pos_idx, neg_idx, vlen, max_vlen, shift)


def get_uncert_model(prop_logits1, prop_logits2, vlen):
    """utils_hual.py:144-161: |sigmoid(S1)-sigmoid(S2)| + |sigmoid(E1)-sigmoid(E2)| on the valid frames (float32)"""
    import torch
    s1, e1 = [torch.sigmoid(torch.from_numpy(np.asarray(a))) for a in prop_logits1]
    s2, e2 = [torch.sigmoid(torch.from_numpy(np.asarray(a))) for a in prop_logits2]
    for t in (s1, e1, s2, e2):
        t[vlen:] = 0
    return torch.abs(s1 - s2).numpy() + torch.abs(e1 - e2).numpy()


def mask_activepoints(start_prob, end_prob, pos_idx, neg_idx, vlen):
    """update_label.py:62-83"""
    if len(pos_idx) == 0:
        for i in neg_idx:
            m = 1 - center_width_gauss(i, 0.3 * vlen, vlen=vlen, max_vlen=len(start_prob))
            start_prob = m * start_prob
            end_prob = m * end_prob
        return start_prob, end_prob
    lo, hi = min(pos_idx), max(pos_idx)
    start_prob[lo + 1:] = 0
    left = [i for i in neg_idx if i < lo]
    if left:
        start_prob[:max(left) + 1] = 0
    end_prob[:hi] = 0
    right = [i for i in neg_idx if i > hi]
    if right:
        end_prob[min(right):] = 0
    return start_prob, end_prob


def renew_label(old_idx, ap, sprob, eprob, vlen, max_vlen, coff):
    """update_label.py:85-123.  coff = get_coff(...) tuple.  Returns [start, end] frame indices of the new pseudo label."""
    pos_idx, neg_idx = list(ap['pos_idx']), list(ap['neg_idx'])
    old_s = center_width_gauss(old_idx[0], 0.5 * vlen, vlen=vlen, max_vlen=max_vlen)
    old_e = center_width_gauss(old_idx[1], 0.5 * vlen, vlen=vlen, max_vlen=max_vlen)
    if pos_idx:
        a1, a2, a3 = coff[0], coff[1], coff[2]
        ds, de = get_distance_score_shift(pos_idx, neg_idx, vlen, max_vlen, shift=-0.3)
        ss = ds * a1 + sprob * a2 + old_s * a3
        es = de * a1 + eprob * a2 + old_e * a3
        ss, es = mask_activepoints(ss, es, pos_idx, neg_idx, vlen)
        return [int(np.argmax(ss)), int(np.argmax(es))]
    a1, a2, a3 = coff[3], coff[4], coff[5]
    ds, de = get_distance_score_shift(pos_idx, neg_idx, vlen, max_vlen, shift=0.9)
    ss = ds * a1 + sprob * a2 + old_s * a3
    es = de * a1 + eprob * a2 + old_e * a3
    ss, es = mask_activepoints(ss, es, pos_idx, neg_idx, vlen)
    outer = np.outer(ss, es)
    keep = np.zeros_like(outer)
    cuts = sorted(neg_idx + [-1, vlen])
    for lo, hi in zip(cuts[:-1], cuts[1:]):            # the span may not contain a negative point
        keep[lo + 1:hi, lo + 1:hi] = outer[lo + 1:hi, lo + 1:hi]
    keep = np.triu(keep)
    return [int(np.argmax(keep.max(axis=1))), int(np.argmax(keep.max(axis=0)))]


def get_uncert_rank(data_old, data_gt, last_prop, coff):
    """update_label.py:125-169: per-sample scores, sorted by the video-level model uncertainty (ascending, stable)."""
    res = []
    for idx, sample in enumerate(data_old):
        vid, duration, _, _, old_ap = sample
        assert vid == last_prop[idx]['vid'] and vid == data_gt[idx][0]
        vlen = last_prop[idx]['v_len']
        s_logit, e_logit = last_prop[idx]['prop_logits']
        sprob, eprob = sigmoid(s_logit), sigmoid(e_logit)
        max_vlen = len(sprob)
        um = get_uncert_model(last_prop[idx]['prop_logits1'], last_prop[idx]['prop_logits2'], vlen)
        ud = get_distance_score(old_ap['pos_idx'], old_ap['neg_idx'], vlen=vlen, max_vlen=max_vlen)
        res.append(dict(idx=idx, gt_idx=time_to_index_v2(data_gt[idx][2], duration, vlen),
                        old_idx=time_to_index_v2(sample[2], duration, vlen), old_ap=old_ap, vlen=vlen, max_vlen=max_vlen,
                        duration=duration, uncert_frame=ud + um * coff[6], uncert_video=np.sum(um), sprob=sprob,
                        eprob=eprob))
    return sorted(res, key=lambda r: r['uncert_video'])


def update_labels(data_old, data_gt, last_prop, coff):
    """update_label.py:173-208 without the file IO: annotate the most uncertain frame of the lower-uncertainty half
    against the ground truth and re-derive those samples' pseudo spans.  Mutates and returns data_old."""
    if len(data_old[0]) == 4:
        for r in data_old:
            r.append({'pos_idx': [], 'neg_idx': []})
    rank = get_uncert_rank(data_old, data_gt, last_prop, coff)
    for rec in rank[:math.ceil(len(rank) / 2)]:
        p = int(np.argmax(rec['uncert_frame']))
        ap = rec['old_ap']
        ap['pos_idx' if rec['gt_idx'][0] <= p <= rec['gt_idx'][1] else 'neg_idx'].append(p)    # utils_hual.py:133-139
        new_idx = renew_label(rec['old_idx'], ap, rec['sprob'], rec['eprob'], rec['vlen'], rec['max_vlen'], coff)
        data_old[rec['idx']][2] = index_to_time(new_idx, rec['duration'], rec['vlen'])
        data_old[rec['idx']][4] = ap
    return data_old

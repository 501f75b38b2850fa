"""Data-parallel plumbing (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI on ROCm,
"gloo" in the CPU tests).  The reference is single-process; data parallelism is the build's addition
and is exact, i.e. N ranks x B clips reproduce the single-process gradient of the N*B-clip batch:

  * gradients: ONE flat fp32 bucket (the whole parameter layout, ~4.75 MB) summed with all_reduce, then scaled by
    1/world inside the fused clip+AdamWD kernel (grad_prescale);
  * matching loss (layers.py:172-173) divides by the GLOBAL valid-frame count: every rank uses the denominator
    n_global/world so that the rank average equals the global masked mean;
  * alignment loss (layers.py:232-247) couples all samples of the batch through [B,B] softmaxes: the per-sample
    l2-normalised features are all-gathered, every rank evaluates the global loss, keeps the gradient rows of its
    own samples and scales them by `world` (they are averaged, not summed, afterwards).
"""
import torch
import torch.distributed as dist


def _forced():
    """HUAL_DP_FORCE_COLLECTIVES=1: issue the collectives on a 1-rank group too (tests of the RCCL code path on one GPU)"""
    import os
    return os.environ.get('HUAL_DP_FORCE_COLLECTIVES') == '1' and dist.is_available() and dist.is_initialized()


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def backend():
    return dist.get_backend() if dist.is_available() and dist.is_initialized() else None


def _host_staged(t):
    """gloo (the CPU / single-GPU rehearsal backend) moves device tensors through the host; RCCL works on them in place"""
    return t.is_cuda and dist.get_backend() == 'gloo'


_custom = {}      # data_ptr of a bucket -> hual_amd.xgmi.OneShotAllReduce (HUAL_ALLREDUCE=custom)


def custom_allreduce_wanted():
    import os
    return os.environ.get('HUAL_ALLREDUCE', 'rccl') == 'custom'


def enable_custom_allreduce(flat):
    """COLLECTIVE.  With HUAL_ALLREDUCE=custom the sum of this bucket runs as the one-shot peer-mapped all-reduce (SURVEY.md 8f #4,
    hual_amd/xgmi.py) instead of RCCL's; default off.  Returns the OneShotAllReduce or None."""
    if not custom_allreduce_wanted() or not flat.is_cuda or world_size() < 2:
        return None
    if flat.data_ptr() not in _custom:
        from .xgmi import OneShotAllReduce
        _custom[flat.data_ptr()] = OneShotAllReduce(flat)
    return _custom[flat.data_ptr()]


def disable_custom_allreduce():
    """COLLECTIVE: unmap and free everything enable_custom_allreduce set up"""
    for ar in list(_custom.values()):
        ar.close()
    _custom.clear()


def allreduce_sum_(flat):
    """in-place sum of the flat gradient bucket over ranks"""
    if world_size() > 1 or _forced():
        ar = _custom.get(flat.data_ptr()) if flat.is_cuda else None
        if ar is not None and ar.flat is flat:
            ar()
        elif _host_staged(flat):
            h = flat.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            flat.copy_(h)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def global_count(local_count, device):
    """sum of a python/0-d count over ranks -> float"""
    t = torch.tensor([float(local_count)], device=device, dtype=torch.float64)
    if world_size() > 1:
        if _host_staged(t):
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def check_padded_length(local_longest, T):
    """Data-parallel shape contract of Trainer.set_batch: every shard is padded to the longest clip of the GLOBAL batch
    (model.py:31 pads to the batch maximum and the reference's conv_block does not mask, modules.py:59-70, so the results depend on the
    padded length).  ONE host collective returns the longest clip and the largest / smallest padded length over ranks: every rank
    sees the same verdict and they raise TOGETHER (a rank raising alone would leave the others waiting in the next collective)."""
    t = torch.tensor([int(local_longest), int(T), -int(T)], dtype=torch.int64)
    if world_size() > 1:
        if dist.get_backend() == 'nccl':
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    gmax, tmax, tmin = (int(x) for x in t.tolist())
    if tmax != -tmin or gmax != tmax:
        raise ValueError('data parallel: every shard must be padded to the longest clip of the GLOBAL batch (%d); padded lengths '
                         'over ranks span %d..%d, this rank has T = %d - model.py:31' % (gmax, -tmin, tmax, T))
    return gmax


def global_min(local_value):
    """min of a python integer over ranks (the collective graph-or-eager decision of Trainer._step_dp)"""
    t = torch.tensor([int(local_value)], dtype=torch.int64)
    if world_size() > 1:
        if dist.get_backend() == 'nccl':
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())


def match_denominator(local_valid_frames, device):
    """denominator each rank must use for the masked matching loss (+1e-12 of layers.py:173 on the global count)"""
    w = world_size()
    return (global_count(local_valid_frames, device) + 1e-12) / w


def gather_features(that, vhat):
    """all_gather of the [B,128] alignment features -> ([W*B,128], [W*B,128]) in rank order (equal B per rank)"""
    w = world_size()
    if w == 1:
        return that, vhat
    both = torch.cat([that, vhat], dim=1).contiguous()
    if _host_staged(both):
        parts = [torch.empty(both.shape, dtype=both.dtype) for _ in range(w)]
        dist.all_gather(parts, both.cpu())
        out = torch.cat(parts, dim=0).to(both.device)
    else:
        out = torch.empty((w * both.shape[0], both.shape[1]), dtype=both.dtype, device=both.device)
        dist.all_gather_into_tensor(out, both)
    d = that.shape[1]
    return out[:, :d].contiguous(), out[:, d:].contiguous()


def allgather_rows_(out, local):
    """out[w*B:(w+1)*B] = rank w's `local` [B,C] (equal B per rank); in place, no host round trip on RCCL"""
    w = world_size()
    if w == 1 and not _forced():
        out.copy_(local)
    elif _host_staged(local):
        parts = [torch.empty(local.shape, dtype=local.dtype) for _ in range(w)]
        dist.all_gather(parts, local.cpu())
        out.copy_(torch.cat(parts, dim=0))
    else:
        dist.all_gather_into_tensor(out, local)
    return out


def local_rows(t, batch):
    r = rank()
    return t[r * batch:(r + 1) * batch]


# ---------------------------------------------------------------------------------------------------------------------
# the epoch loop, data parallel (runner_utils.py:139-159 + data_loader.py:23-28 over `world` ranks)
def shard_plan(order, batch_size, world, vlen, nwords, maxchars, min_chars=None):
    """Host-side plan of ONE data-parallel epoch; pure numpy, the same on every rank (every rank holds the set's length arrays).

    order: the epoch's shuffled sample ids; batch_size: clips per RANK (a global batch is batch_size * world consecutive ids of
    `order`, rank r takes the r-th slice of batch_size).  Returns a list of steps, each a dict
        ids     int32 [world * B]   the global batch, rank r's shard = ids[r * B : (r + 1) * B]
        B       clips per rank (batch_size; the epoch's last global batch: what is left // world - shards must be equal for the
                all-gather of the alignment features and for `mean over clips` to decompose; the < world clips that do
                not fill a round are dropped from this epoch and reported in `dropped`)
        shape   (T, L, C) of the GLOBAL batch: every shard is padded to it, NOT to its own maxima - model.py:31 pads to the batch
                maximum and the reference's conv_block does not mask (modules.py:59-70), so a shard padded to its own longest clip /
                query / word would not reproduce the single-process numbers
        frames  valid frames of the global batch: the matching loss divides by it (layers.py:172-173), each rank by frames / world
    With world = 1 this is exactly the reference's batching (ragged last batch kept)."""
    import numpy as np
    order = np.ascontiguousarray(order, dtype=np.int32)
    world, bs = int(world), int(batch_size)
    steps, dropped = [], 0
    for lo in range(0, len(order), bs * world):
        ids = order[lo:lo + bs * world]
        B = bs if len(ids) == bs * world else len(ids) // world
        if B == 0:
            dropped += len(ids)
            break
        dropped += len(ids) - B * world
        ids = ids[:B * world]
        C = int(maxchars[ids].max())
        steps.append(dict(ids=ids, B=B, shape=(int(vlen[ids].max()), int(nwords[ids].max()), max(C, min_chars) if min_chars else C),
                          frames=int(vlen[ids].sum())))
    return steps, dropped


def check_same(value, what='value'):
    """every rank must hold the same integer (e.g. a checksum of the epoch's permutation): MIN == MAX over ranks, and the ranks
    raise TOGETHER if not"""
    t = torch.tensor([int(value), -int(value)], dtype=torch.int64)
    if world_size() > 1:
        if dist.get_backend() == 'nccl':
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    hi, lo = int(t[0]), -int(t[1])
    if hi != lo:
        raise ValueError('data parallel: the ranks disagree on %s (%d .. %d)' % (what, lo, hi))


def allgather_cat(t):
    """[world, *t.shape] of every rank's `t` (equal shapes), on t's device"""
    w = world_size()
    if w == 1:
        return t.unsqueeze(0)
    if _host_staged(t) or not t.is_cuda:
        h = t.cpu().contiguous()
        parts = [torch.empty_like(h) for _ in range(w)]
        dist.all_gather(parts, h)
        return torch.stack(parts, dim=0).to(t.device)
    out = torch.empty((w,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t.contiguous())
    return out


def gather_objects(obj, dst=0):
    """python objects of every rank as a list on rank `dst` (None elsewhere); world 1: [obj]"""
    if world_size() == 1:
        return [obj]
    out = [None] * world_size() if rank() == dst else None
    dist.gather_object(obj, out, dst=dst)
    return out


def broadcast_object(obj, src=0):
    if world_size() == 1:
        return obj
    box = [obj if rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def barrier():
    if world_size() > 1:
        dist.barrier()

"""One-shot all-reduce of the flat gradient bucket over peer mappings (csrc/xgmi.hip, hual_xgmi_allreduce).

The reference trains on one pinned GPU (/root/reference/utils/runner_utils.py:11); data parallelism is this build's addition and its ONE
collective of size is the 4.75 MB gradient sum.  At that size a ring is hop-latency bound; xGMI is point to point, so every rank reads
every peer's bucket directly: one launch, two flag barriers (csrc/xgmi.hip).

Setup (collective, once per bucket): every rank exports its bucket, a scratch slice and its flag words with hipIpcGetMemHandle, the 64-byte
handles travel over the process group (all_gather_object), every rank opens its peers' handles.  HSA_ENABLE_IPC_MODE_LEGACY=0 must be in
the environment (dmabuf IPC - the only mode this host driver supports).

OFF by default: hual_amd/dist.py uses it only with HUAL_ALLREDUCE=custom.  It has been validated with two processes on ONE GPU (IPC to
the same device; bit-equal to the host-staged sum, tests/test_gpu_xgmi.py); no multi-GPU box has run it yet, RCCL stays the default.
"""
import ctypes

import torch
import torch.distributed as dist

from . import lib


def _export(ptr):
    """(handle of the allocation `ptr` lies in, ptr's offset inside it): tensors of PyTorch's caching allocator share segments"""
    h, off = (ctypes.c_char * 64)(), ctypes.c_uint64()
    lib.check(lib.load().hual_xgmi_ipc_export(ctypes.c_void_p(ptr), h, ctypes.byref(off)))
    return bytes(h), int(off.value)


def _open(handle):
    p = ctypes.c_void_p()
    lib.check(lib.load().hual_xgmi_ipc_open(handle, ctypes.byref(p)))
    return p.value


class OneShotAllReduce:
    def __init__(self, flat, group=None):
        """flat: this rank's contiguous float32 CUDA bucket (numel % 4 == 0, 16-byte aligned); every rank of `group` constructs its own at
        the same time (the constructor is collective)."""
        assert flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous() and flat.numel() % 4 == 0
        self._lib = lib.load()
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        if self.world > 16:
            raise lib.HualError('one-shot all-reduce: at most 16 ranks')
        self.flat = flat
        n = flat.numel()
        self.chunk = (((n + self.world - 1) // self.world) + 3) // 4 * 4
        dev = flat.device
        self.scratch = torch.empty(self.chunk, dtype=torch.float32, device=dev)
        self.seq = torch.zeros(1, dtype=torch.int32, device=dev)
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        p = ctypes.c_void_p()
        with torch.cuda.device(dev):
            try:
                lib.check(self._lib.hual_xgmi_flags_alloc(ctypes.byref(p)))
                self._flags = p.value
                torch.cuda.synchronize()
                mine = (_export(flat.data_ptr()), _export(self.scratch.data_ptr()), _export(self._flags), flat.numel())
            except Exception as e:                               # (e.g. HSA_ENABLE_IPC_MODE_LEGACY=0 missing) - raised on every rank below
                mine = 'export failed: %s' % e
            everyone = [None] * self.world
            dist.all_gather_object(everyone, mine, group=group)
            bad = [(r, v) for r, v in enumerate(everyone) if isinstance(v, str)]
            if bad:
                if p.value:
                    self._lib.hual_xgmi_flags_free(p)
                raise lib.HualError('one-shot all-reduce: rank %d: %s' % bad[0])
            self._opened = []
            bases = {}                                           # handle -> mapped base: two tensors of one segment are opened once
            ptrs = ([], [], [])
            err = None
            try:
                for r, (hf, hs, hg, nr) in enumerate(everyone):
                    if nr != n:
                        raise lib.HualError('one-shot all-reduce: rank %d holds a bucket of %d floats, this rank %d' % (r, nr, n))
                    if r == self.rank:
                        own = (flat.data_ptr(), self.scratch.data_ptr(), self._flags)
                        for k in range(3):
                            ptrs[k].append(own[k])
                    else:
                        for k, (h, off) in enumerate((hf, hs, hg)):
                            if h not in bases:
                                bases[h] = _open(h)
                                self._opened.append(bases[h])
                            ptrs[k].append(bases[h] + off)
            except Exception as e:                               # the verdict is collective: every rank raises or none does
                err = e
            verdicts = [None] * self.world
            dist.all_gather_object(verdicts, None if err is None else str(err), group=group)
            bad = [(r, v) for r, v in enumerate(verdicts) if v is not None]
            if bad:
                for q in self._opened:
                    self._lib.hual_xgmi_ipc_close(ctypes.c_void_p(q))
                self._opened = None
                dist.barrier(group=group)
                self._lib.hual_xgmi_flags_free(ctypes.c_void_p(self._flags))
                raise lib.HualError('one-shot all-reduce: peer mapping failed on rank %d: %s' % bad[0])
        arr = ctypes.c_void_p * self.world
        self._flat_p, self._scratch_p, self._flags_p = (arr(*[ctypes.c_void_p(x) for x in ptrs[k]]) for k in range(3))
        dist.barrier(group=group)             # every rank has opened every handle before anybody may free or signal

    def __call__(self, stream=None):
        """enqueue the all-reduce (sum, in place) of the bucket on the current stream; no host synchronisation"""
        lib.check(self._lib.hual_xgmi_allreduce(self.rank, self.world, self._flat_p, self._scratch_p, self._flags_p, lib.ptr(self.seq),
                                                lib.ptr(self.status), self.flat.numel(), self.chunk, lib.stream_ptr(stream)))
        return self.flat

    def check(self):
        """host-side look at the status word (synchronises): raises if a peer did not arrive within the kernel's spin limit"""
        if int(self.status.item()) != 0:
            raise lib.HualError('one-shot all-reduce: a peer did not arrive within the spin limit - the gradient bucket is invalid')

    def close(self):
        """collective: unmap the peers' memory, free the flag words"""
        if getattr(self, '_opened', None) is None:
            return
        torch.cuda.synchronize()
        dist.barrier(group=self.group)        # nobody is still reading this rank's memory
        for q in self._opened:
            self._lib.hual_xgmi_ipc_close(ctypes.c_void_p(q))
        self._opened = None
        dist.barrier(group=self.group)        # every mapping of this rank's flags is gone before they are freed
        self._lib.hual_xgmi_flags_free(ctypes.c_void_p(self._flags))
        self._flags = None

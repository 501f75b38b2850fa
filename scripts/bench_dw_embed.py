"""GPU box: time single dw_kernel jobs (plain vs the overlapping-row embed job) with the library's own dispatch events."""
import ctypes
import sys
import torch
sys.path.insert(0, '.')
from hual_amd import lib
l = lib.load()
dev = torch.device('cuda:0')


def run(M, K, lda, tag, rpb=512):
    A = torch.randn(M + 8, lda, device=dev)
    dY = torch.randn(M, 128, device=dev)
    dW = torch.zeros(K, 128, device=dev)
    db = torch.zeros(128, device=dev)
    for _ in range(3):
        lib.check(l.hual_linear_dw(lib.ptr(A), lda, lib.ptr(dY), 128, lib.ptr(dW), 128, lib.ptr(db), M, K, 128, rpb, None, 0, lib.stream_ptr()))
    torch.cuda.synchronize()
    l.hual_prof_begin()
    for _ in range(10):
        lib.check(l.hual_linear_dw(lib.ptr(A), lda, lib.ptr(dY), 128, lib.ptr(dW), 128, lib.ptr(db), M, K, 128, rpb, None, 0, lib.stream_ptr()))
    n = l.hual_prof_end()
    for i in range(n):
        name = ctypes.create_string_buffer(128)
        la, us = ctypes.c_int64(0), ctypes.c_double(0)
        fl, by = ctypes.c_double(0), ctypes.c_double(0)
        l.hual_prof_get(i, name, 128, ctypes.byref(la), ctypes.byref(us), ctypes.byref(fl), ctypes.byref(by))
        print('%-28s %-22s avg %7.1f us  %6.1f TF/s' % (tag, name.value.decode(), us.value / la.value, fl.value / us.value / 1e6))


run(9472, 128, 128, 'plain M9472 K128')
run(10240, 128, 128, 'plain M10240 K128')
run(10240, 256, 256, 'plain M10240 K256 lda256')
run(10240, 256, 64, 'embed M10240 K256 lda64')
run(10240, 256, 64, 'embed rpb256', rpb=256)
run(10240, 256, 64, 'embed rpb128', rpb=128)

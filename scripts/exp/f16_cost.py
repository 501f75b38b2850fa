"""GPU box: cost of the f16x3 per-row scaling in the dense kernel (HUAL_F16_DBG: 1 = no row maximum, 2 = no fold shuffle)."""
import ctypes, os, sys
sys.path.insert(0, '.')
import torch
from hual_amd import lib
dev = torch.device('cuda:0')
l = lib.load()
M, K = 9472, 128
A = torch.randn(M, K, device=dev); W = torch.randn(K, 128, device=dev) / 11; b = torch.randn(128, device=dev)
def run(tag):
    for _ in range(20): lib.linear_bf16x3(A, W, b, act=1)
    torch.cuda.synchronize()
    l.hual_prof_begin()
    for _ in range(200): lib.linear_bf16x3(A, W, b, act=1)
    n = l.hual_prof_end()
    for i in range(n):
        name = ctypes.create_string_buffer(128); la = ctypes.c_int64(0); us = ctypes.c_double(0); f = ctypes.c_double(0); by = ctypes.c_double(0)
        l.hual_prof_get(i, name, 128, ctypes.byref(la), ctypes.byref(us), ctypes.byref(f), ctypes.byref(by))
        if b'gemm' in name.value: print(tag, name.value.decode(), round(us.value / la.value, 2), 'us')
for rep in range(2):
    for dbg in ('0', '1', '2', '3'):
        os.environ['HUAL_F16_DBG'] = dbg
        run('dbg=' + dbg)

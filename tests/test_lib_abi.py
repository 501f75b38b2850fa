"""CPU-side checks of the C ABI: the library builds, loads and exports every symbol include/hual_seqpan.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'hual_seqpan.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(hual_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from hual_amd import build, lib
    build.build()
    l = ctypes.CDLL(lib.LIB_PATH)
    names = _declared()
    assert 'hual_abi_version' in names and len(names) >= 4
    for n in names:
        assert hasattr(l, n), 'missing export ' + n
    assert lib.load().hual_abi_version() == lib.ABI_VERSION


def test_bad_arguments_fail_loudly_without_a_gpu():
    from hual_amd import lib
    import pytest
    l = lib.load()
    rc = l.hual_linear_bf16x3(None, 0, None, 0, None, None, 0, 4, 16, 128, 0, None, 0, None)
    assert rc != 0 and b'null pointer' in l.hual_last_error()
    with pytest.raises(lib.HualError):
        lib.check(rc)

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


def test_shape_and_config_errors_mirror_the_reference():
    """host-side validation, no GPU: what TensorFlow would raise inside sess.run comes back as an error code + message.
    tf.assert_less_equal(seq_len, max_pos_len) (/root/reference/models/modules.py:44) for clips AND queries (one position table,
    model.py:53,56); the VALID width-4 conv of the char CNN (modules.py:19-38) needs four characters per word; the kernels are
    specialised for the YAMLs' dim 128 / 8 heads and say so instead of computing something else"""
    import ctypes
    import pytest
    from hual_amd import lib
    cfg = lib.make_cfg(max_vlen=64)
    assert lib.query_workspace(cfg, 2, 64, 8, 5) > 0
    for shape, msg in (((2, 65, 8, 5), 'longer than max_vlen'), ((2, 64, 65, 5), 'longer than max_vlen'), ((2, 64, 8, 3), 'C >= 4'),
                       ((0, 64, 8, 5), 'empty batch')):
        with pytest.raises(lib.HualError, match=msg):
            lib.query_workspace(cfg, *shape)
    l = lib.load()
    for kw, msg in ((dict(dim=256), b'dim must be 128'), (dict(num_heads=4), b'num_heads must be 8'), (dict(max_vlen=300), b'max_vlen'),
                    (dict(vdim=1000), b'vdim'), (dict(attn_layer=0), b'attn_layer')):
        c = lib.make_cfg(**kw)
        assert l.hual_seqpan_validate(ctypes.byref(c)) != 0 and msg in l.hual_last_error()
    assert l.hual_seqpan_validate(ctypes.byref(lib.make_cfg())) == 0
    assert l.hual_seqpan_dw_table_bytes() > 0 and l.hual_xgmi_flags_bytes() >= 4 * 34

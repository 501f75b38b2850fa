"""ctypes binding of libhual_seqpan.so (include/hual_seqpan.h).  No fallback: a missing or stale
library raises - the product path never routes through a CPU implementation."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HUAL_LIB_PATH: an experiment build of the library (scripts/exp/tl_variant.sh) - the in-tree file is never overwritten
LIB_PATH = os.environ.get('HUAL_LIB_PATH') or os.path.join(_HERE, 'libhual_seqpan.so')
ABI_VERSION = 8

_lib = None


class HualError(RuntimeError):
    pass


class hual_cfg(ctypes.Structure):
    _fields_ = [('vdim', ctypes.c_int32), ('dim', ctypes.c_int32), ('num_heads', ctypes.c_int32),
                ('word_dim', ctypes.c_int32), ('char_dim', ctypes.c_int32), ('max_vlen', ctypes.c_int32),
                ('attn_layer', ctypes.c_int32), ('num_chars', ctypes.c_int32), ('num_words', ctypes.c_int32),
                ('no_gumbel', ctypes.c_int32), ('match_lambda', ctypes.c_float), ('tau', ctypes.c_float),
                ('clip_norm', ctypes.c_float)]


class hual_param_entry(ctypes.Structure):
    _fields_ = [('name', ctypes.c_char * 112), ('offset', ctypes.c_uint64), ('size', ctypes.c_uint64),
                ('ndim', ctypes.c_int32), ('shape', ctypes.c_int32 * 4), ('decay', ctypes.c_int32)]


class hual_batch(ctypes.Structure):
    _fields_ = [('video', ctypes.c_void_p), ('video_seq_len', ctypes.c_void_p), ('word_ids', ctypes.c_void_p),
                ('char_ids', ctypes.c_void_p), ('B', ctypes.c_int32), ('T', ctypes.c_int32), ('L', ctypes.c_int32),
                ('C', ctypes.c_int32), ('video_dtype', ctypes.c_int32)]


class hual_labels(ctypes.Structure):
    _fields_ = [('y1', ctypes.c_void_p), ('y2', ctypes.c_void_p), ('match_labels', ctypes.c_void_p),
                ('inner_labels', ctypes.c_void_p)]


class hual_outputs(ctypes.Structure):
    _fields_ = [('start_logits', ctypes.c_void_p), ('end_logits', ctypes.c_void_p), ('match_scores', ctypes.c_void_p),
                ('start_index', ctypes.c_void_p), ('end_index', ctypes.c_void_p), ('loss_terms', ctypes.c_void_p)]


class hual_run_opts(ctypes.Structure):
    _fields_ = [('drop_rate', ctypes.c_float), ('rng_state', ctypes.c_void_p), ('match_denom_override', ctypes.c_float),
                ('align_external', ctypes.c_int32), ('static_tables', ctypes.c_int32),
                ('match_denom_dev', ctypes.c_void_p), ('debug_taps', ctypes.c_int32), ('grads_prezero', ctypes.c_void_p),
                ('prezero_token', ctypes.c_void_p), ('deferred_loss_terms', ctypes.c_void_p),
                ('dw_table', ctypes.c_void_p), ('dw_table_bytes', ctypes.c_uint64)]


class hual_al_set(ctypes.Structure):
    _fields_ = [('N', ctypes.c_int32), ('ld', ctypes.c_int32), ('vlen', ctypes.c_void_p), ('tlen', ctypes.c_void_p),
                ('ap_off', ctypes.c_void_p), ('ap_idx', ctypes.c_void_p), ('ap_pos', ctypes.c_void_p)]


class hual_dataset(ctypes.Structure):
    _fields_ = [('feat_bank', ctypes.c_void_p), ('feat_off', ctypes.c_void_p), ('vdim', ctypes.c_int32),
                ('sample_vid', ctypes.c_void_p), ('word_off', ctypes.c_void_p), ('word_bank', ctypes.c_void_p),
                ('char_off', ctypes.c_void_p), ('char_bank', ctypes.c_void_p), ('s_ind', ctypes.c_void_p),
                ('e_ind', ctypes.c_void_p)]


class hual_ws_entry(ctypes.Structure):
    _fields_ = [('name', ctypes.c_char * 48), ('offset', ctypes.c_uint64), ('rows', ctypes.c_uint64),
                ('cols', ctypes.c_uint64)]


def load():
    global _lib
    if _lib is not None:
        return _lib
    # torch first: its wheel bundles its own libamdhip64, and a process must end up with ONE HIP runtime - if this library
    # were loaded before torch it would bind /opt/rocm's copy and its launches would not see torch's device context
    # ("no ROCm-capable device is detected")
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise HualError('%s not found: build it with `python -m hual_amd.build` (or __graft_entry__.build())' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.hual_abi_version.restype = ctypes.c_int
    lib.hual_last_error.restype = ctypes.c_char_p
    v = lib.hual_abi_version()
    if v != ABI_VERSION:
        raise HualError('libhual_seqpan.so ABI %d != python binding %d: rebuild' % (v, ABI_VERSION))
    lib.hual_seqpan_dw_table_bytes.restype = ctypes.c_uint64
    vp, i32, u64, f32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_float
    P = ctypes.POINTER
    lib.hual_seqpan_validate.argtypes = [P(hual_cfg)]
    lib.hual_seqpan_param_count.argtypes = [P(hual_cfg), P(u64), P(u64)]
    lib.hual_seqpan_param_table.argtypes = [P(hual_cfg), P(hual_param_entry), i32]
    lib.hual_seqpan_query_workspace.argtypes = [P(hual_cfg), i32, i32, i32, i32, P(u64)]
    lib.hual_seqpan_ws_table.argtypes = [P(hual_cfg), i32, i32, i32, i32, P(hual_ws_entry), i32]
    lib.hual_seqpan_forward.argtypes = [P(hual_cfg), vp, vp, P(hual_batch), P(hual_labels), P(hual_outputs),
                                        P(hual_run_opts), vp, u64, vp]
    lib.hual_seqpan_backward.argtypes = [P(hual_cfg), vp, vp, P(hual_batch), P(hual_labels), P(hual_run_opts), vp, vp,
                                         u64, vp]
    lib.hual_adamw_clip_step.argtypes = [vp, vp, vp, vp, vp, u64, vp, f32, f32, vp, vp]
    lib.hual_adamw_clip_step_rng.argtypes = [vp, vp, vp, vp, vp, u64, vp, f32, f32, vp, vp, vp]
    lib.hual_adamw_clip_step_loop.argtypes = [vp, vp, vp, vp, vp, u64, vp, f32, f32, vp, vp, vp, vp, vp, i32, i32, i32, vp]
    lib.hual_assemble_batch_cursor.argtypes = [P(hual_dataset), vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.hual_xgmi_flags_bytes.restype = ctypes.c_uint64
    lib.hual_xgmi_flags_alloc.argtypes = [P(vp)]
    lib.hual_xgmi_flags_free.argtypes = [vp]
    lib.hual_xgmi_ipc_export.argtypes = [vp, vp, P(u64)]
    lib.hual_xgmi_ipc_open.argtypes = [vp, P(vp)]
    lib.hual_xgmi_ipc_close.argtypes = [vp]
    lib.hual_xgmi_allreduce.argtypes = [i32, i32, P(vp), P(vp), P(vp), vp, vp, u64, u64, vp]
    lib.hual_align_loss.argtypes = [vp, vp, i32, vp, vp, vp, vp, f32, vp]
    lib.hual_align_loss_rows.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, f32, vp]
    blk = [P(hual_cfg), vp, P(hual_batch), P(hual_run_opts)]
    lib.hual_video_proj_ln_fwd.argtypes = [P(hual_cfg), vp, vp, P(hual_batch), P(hual_run_opts), vp, vp, u64, vp]
    lib.hual_conv_block_fwd.argtypes = blk + [vp, vp, vp, u64, vp]
    lib.hual_conv_block_bwd.argtypes = blk + [vp, vp, vp, vp, u64, vp]
    lib.hual_dual_attn_fwd.argtypes = blk + [i32, vp, vp, vp, u64, vp]
    lib.hual_dual_attn_bwd.argtypes = blk + [i32, vp, vp, vp, vp, u64, vp]
    lib.hual_cq_attn_fwd.argtypes = blk + [vp, vp, vp, u64, vp]
    lib.hual_cq_attn_bwd.argtypes = blk + [vp, vp, vp, vp, u64, vp]
    lib.hual_predictor_fwd.argtypes = blk + [vp, vp, vp, vp, vp, vp, u64, vp]
    lib.hual_predictor_bwd.argtypes = blk + [vp, vp, vp, vp, vp, u64, vp]
    lib.hual_prof_get.argtypes = [i32, ctypes.c_char_p, i32, P(ctypes.c_int64), P(ctypes.c_double), P(ctypes.c_double),
                                  P(ctypes.c_double)]
    lib.hual_prof_kernel_pipe.argtypes = [ctypes.c_char_p, P(i32), P(i32)]
    lib.hual_linear_bf16x3.argtypes = [vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp, u64, vp]
    lib.hual_layer_norm_fwd.argtypes = [vp, vp, vp, vp, vp, vp, i32, vp]
    lib.hual_attention_fwd.argtypes = [vp, i32, vp, vp, i32, vp, i32, i32, i32, i32, vp, vp, vp]
    lib.hual_attention_fwd_save.argtypes = [vp, i32, vp, vp, i32, vp, i32, i32, i32, i32, vp, vp, vp, vp, i32, vp, f32, i32, vp]
    lib.hual_attention_bwd.argtypes = [vp, i32, vp, vp, i32, vp, i32, vp, vp, i32, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, vp, vp,
                                       vp, f32, i32, vp]
    lib.hual_attention_keep_row_bytes.argtypes = [i32]
    lib.hual_span_argmax.argtypes = [vp, vp, vp, vp, vp, i32, i32, vp]
    lib.hual_linear_dw.argtypes = [vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, vp, u64, vp]
    lib.hual_al_score.argtypes = [P(hual_al_set), vp, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp]
    lib.hual_al_renew.argtypes = [P(hual_al_set), vp, i32, vp, vp, vp, P(ctypes.c_double), vp, vp]
    lib.hual_assemble_batch.argtypes = [P(hual_dataset), vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.hual_assemble_batch_carry.argtypes = [P(hual_dataset), vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp]
    _lib = lib
    return lib


def check(code):
    if code != 0:
        raise HualError('libhual_seqpan: %s (code %d)' % (load().hual_last_error().decode(), code))


def ptr(t):
    """device pointer of a torch tensor (or None)"""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_ptr(stream=None):
    import torch
    s = torch.cuda.current_stream() if stream is None else stream
    return ctypes.c_void_p(s.cuda_stream)


# ---------------------------------------------------------------- host-only queries (no GPU needed)
def make_cfg(**kw):
    c = hual_cfg()
    d = dict(vdim=1024, dim=128, num_heads=8, word_dim=300, char_dim=50, max_vlen=64, attn_layer=2, num_chars=40,
             num_words=500, no_gumbel=1, match_lambda=1.0, tau=0.3, clip_norm=1.0)
    d.update(kw)
    for k, v in d.items():
        setattr(c, k, v)
    return c


def param_table(cfg):
    lib = load()
    n = lib.hual_seqpan_param_table(ctypes.byref(cfg), None, 0)
    if n < 0:
        check(n)
    arr = (hual_param_entry * n)()
    n2 = lib.hual_seqpan_param_table(ctypes.byref(cfg), arr, n)
    assert n2 == n
    total, count = ctypes.c_uint64(), ctypes.c_uint64()
    check(lib.hual_seqpan_param_count(ctypes.byref(cfg), ctypes.byref(total), ctypes.byref(count)))
    ents = [dict(name=e.name.decode(), offset=int(e.offset), size=int(e.size), shape=list(e.shape[:e.ndim]),
                 decay=bool(e.decay)) for e in arr]
    return ents, int(total.value), int(count.value)


def query_workspace(cfg, B, T, L, C):
    b = ctypes.c_uint64()
    check(load().hual_seqpan_query_workspace(ctypes.byref(cfg), B, T, L, C, ctypes.byref(b)))
    return int(b.value)


def ws_table(cfg, B, T, L, C):
    lib = load()
    n = lib.hual_seqpan_ws_table(ctypes.byref(cfg), B, T, L, C, None, 0)
    if n < 0:
        check(n)
    arr = (hual_ws_entry * n)()
    lib.hual_seqpan_ws_table(ctypes.byref(cfg), B, T, L, C, arr, n)
    return {e.name.decode(): (int(e.offset), int(e.rows), int(e.cols)) for e in arr}


# ---------------------------------------------------------------- per-kernel entry points
def linear_bf16x3(A, W, bias=None, act=0, trans_w=False):
    """split-bf16 dense: W [K,128] (trans_w False) or [N,128] used transposed (dX)"""
    import torch
    M, K = A.shape
    N = W.shape[0] if trans_w else W.shape[1]
    Y = torch.empty(M, N, device=A.device, dtype=torch.float32)
    nbytes = ((N + 127) // 128) * 65536 if trans_w else ((K + 127) // 128) * 65536
    scratch = torch.zeros(nbytes, dtype=torch.uint8, device=A.device)
    check(load().hual_linear_bf16x3(ptr(A), A.stride(0), ptr(W), int(trans_w), ptr(bias), ptr(Y), Y.stride(0), M, K, N, act,
                                    ptr(scratch), nbytes, stream_ptr()))
    return Y


def linear_dw(A, dY, dW, db=None, workgroups=0):
    M, K = A.shape
    N = dY.shape[1]
    import torch
    scratch = torch.empty(256, dtype=torch.float32, device=A.device)
    check(load().hual_linear_dw(ptr(A), A.stride(0), ptr(dY), dY.stride(0), ptr(dW), dW.stride(0), ptr(db), M, K, N,
                                workgroups, ptr(scratch), 1024, stream_ptr()))

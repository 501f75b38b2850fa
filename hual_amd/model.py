"""SeqPAN on MI355X: the Python object the runner code talks to.

Mirrors the attribute surface of /root/reference/models/model.py:7-122 as consumed by
/root/reference/utils/runner_utils.py (get_feed_dict :53-65, train_epoch :139-159, test_epoch :161-176,
eval_test_save :69-110):

    reference                                             here
    ------------------------------------------------      -------------------------------------------------
    SeqPAN(configs, graph, word_vectors)                  SeqPAN(configs, word_vectors)
    sess.run([start_index, end_index], feed)              model.forward(**feeds)            (all 5 fetches, 1 pass)
    sess.run([start_logits, end_logits], feed)            model.forward(...)['start_logits'] ...
    sess.run(match_scores, feed)                          model.forward(...)['match_scores']
    sess.run([train_op, loss, start_index, end_index])    model.train_step(feeds, lr, drop_rate)
    tf.train.Saver                                        model.state_dict() / load_state_dict()  (TF variable names)

All compute runs in libhual_seqpan.so (hand-written HIP for gfx950) through ctypes; PyTorch only owns device
memory and streams.  There is no CPU fallback: without the library or a GPU this module raises.
"""
import ctypes

import numpy as np
import torch

from . import lib
from .params import ParamTable


def _get(cfg, path, default=None):
    cur = cfg
    for k in path.split('.'):
        if isinstance(cur, dict):
            if k not in cur:
                return default
            cur = cur[k]
        else:
            if not hasattr(cur, k):
                return default
            cur = getattr(cur, k)
    return cur


def cfg_from_configs(configs, num_words):
    """configs: the YAML of configs/<task>/SeqPAN.yaml as nested dict / attribute object (+ num_chars)."""
    return lib.make_cfg(
        vdim=int(_get(configs, 'model.vdim')), dim=int(_get(configs, 'model.dim')),
        num_heads=int(_get(configs, 'model.num_heads')), word_dim=int(_get(configs, 'model.word_dim')),
        char_dim=int(_get(configs, 'model.char_dim')), max_vlen=int(_get(configs, 'model.max_vlen')),
        attn_layer=int(_get(configs, 'model.attn_layer')), num_chars=int(_get(configs, 'num_chars')),
        num_words=int(num_words), no_gumbel=1 if _get(configs, 'loss.no_gumbel', True) else 0,
        match_lambda=float(_get(configs, 'loss.match_lambda', 1.0)), tau=float(_get(configs, 'loss.tau', 0.3)),
        clip_norm=float(_get(configs, 'train.clip_norm', 1.0)))


class SeqPAN:
    def __init__(self, configs, word_vectors, device='cuda:0', seed=12345, rng_seed=12345):
        if not torch.cuda.is_available():
            raise lib.HualError('SeqPAN needs a GPU: the HIP path has no CPU fallback')
        lib.load()
        self.device = torch.device(device)
        wv = np.asarray(word_vectors, dtype=np.float32)
        self.cfg = configs if isinstance(configs, lib.hual_cfg) else cfg_from_configs(configs, wv.shape[0] + 2)
        lib.check(lib.load().hual_seqpan_validate(ctypes.byref(self.cfg)))
        assert wv.shape == (self.cfg.num_words - 2, self.cfg.word_dim), 'word_vectors must be [num_words-2, word_dim]'
        self.table = ParamTable(self.cfg)
        self.word_table = torch.from_numpy(wv).to(self.device).contiguous()
        self.params = torch.from_numpy(self.table.init_flat(seed)).to(self.device)
        self.grads = torch.zeros_like(self.params)
        self.adam_m = torch.zeros_like(self.params)
        self.adam_v = torch.zeros_like(self.params)
        self.decay = torch.from_numpy(self.table.decay_flat(0.01)).to(self.device)
        self.lr = torch.zeros(1, device=self.device)
        self.lr_value = 0.0                                       # host copy of self.lr (skips redundant fills)
        self.sqnorm = torch.zeros(256, device=self.device)       # hual_adamw_clip_step scratch
        # Philox state {seed lo, seed hi, offset}; offset advances once per train step (on device, graph friendly)
        st = np.array([rng_seed & 0xFFFFFFFF, (rng_seed >> 32) & 0xFFFFFFFF, 0], dtype=np.uint32).view(np.int32)
        self.rng_state = torch.from_numpy(st.copy()).to(self.device)
        self.global_step = 0
        self._ws = None
        self._ws_shape = None
        self._ws_table = None
        self._ws_tables = {}                                      # (B,T,L,C) -> name table (one dry pass per distinct shape)
        self._ws_need = {}                                        # (B,T,L,C) -> bytes
        self.ws_poison = None                                     # tests: byte the workspace is filled with before each shape change
        self.world = 1

    # ------------------------------------------------------------------ parameters by TF name
    def state_dict(self):
        return self.table.unpack(self.params.detach().cpu().numpy())

    def load_state_dict(self, named):
        self.params.copy_(torch.from_numpy(self.table.pack(named)).to(self.device))

    def grads_dict(self):
        return self.table.unpack(self.grads.detach().cpu().numpy())

    def set_rng(self, seed, offset):
        st = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, offset & 0xFFFFFFFF], dtype=np.uint32).view(np.int32)
        self.rng_state.copy_(torch.from_numpy(st.copy()).to(self.device))

    # ------------------------------------------------------------------ workspace
    def reserve(self, B, T, L, C):
        """Size the workspace ONCE for the largest batch a loop will see (batch_size, max_vlen, longest query, longest word):
        every smaller shape then runs in the same allocation - no allocation, no fill, no synchronisation when the padded
        shape changes from batch to batch (runner_utils.py:139-159 feeds a new T / L / C nearly every step)."""
        need = lib.query_workspace(self.cfg, B, T, L, C) + 256
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            assert self._ws.data_ptr() % 256 == 0
            self._ws_shape = None
        return self._ws

    def _workspace(self, B, T, L, C):
        """The workspace for a (B,T,L,C) batch: the one allocation, grown only if this shape needs more than any before it.
        Its contents never matter: every buffer the kernels accumulate into is zeroed by the step's own prologue launch, so
        the memory is NOT cleared (tests/parity_util.compare poisons it with 0xFF bytes before every run)."""
        shape = (B, T, L, C)
        if self._ws_shape != shape:
            if shape not in self._ws_tables:
                self._ws_need[shape] = lib.query_workspace(self.cfg, B, T, L, C) + 256
                self._ws_tables[shape] = lib.ws_table(self.cfg, B, T, L, C)
            if self._ws is None or self._ws.numel() < self._ws_need[shape]:
                self._ws = torch.empty(self._ws_need[shape], dtype=torch.uint8, device=self.device)
                assert self._ws.data_ptr() % 256 == 0
            if self.ws_poison is not None:
                self._ws.fill_(self.ws_poison)
            self._ws_shape = shape
            self._ws_table = self._ws_tables[shape]
        return self._ws

    def tap(self, name):
        """fp32 view of a named intermediate in the workspace (debugging / parity tests)."""
        if name in ('align.that', 'align.vhat'):      # the two halves of the [B,256] buffer the forward leaves side by side
            tv = self.tap('align.tv')
            return tv[:, :128] if name == 'align.that' else tv[:, 128:]
        off, rows, cols = self._ws_table[name]
        return self._ws[off:off + rows * cols * 4].view(torch.float32).view(rows, cols)

    def tap_bits(self, name):
        """bool [rows, 128] view of a bit plane of the workspace (relu active sets "*.rb*", dropout keep sets "*.kb*":
        byte [row * 16 + (col >> 3)], bit col & 7 - csrc/tilecore.h)"""
        off, rows, cols = self._ws_table[name]
        by = self._ws[off:off + rows * 16].view(rows, 16).cpu().numpy()
        return torch.from_numpy(np.unpackbits(by, axis=1, bitorder='little').astype(bool))

    # ------------------------------------------------------------------ feeds
    def _to_dev(self, a, dtype):
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(self.device)

    def _prep(self, video_inputs, video_seq_len, word_ids, char_ids):
        if not isinstance(video_seq_len, torch.Tensor) or not video_seq_len.is_cuda:
            lens_h = np.asarray(video_seq_len)
            if int(lens_h.max()) != int(np.shape(video_inputs)[1]):
                raise ValueError('video_inputs T (%d) must equal max(video_seq_len) (%d) - model.py:31'
                                 % (np.shape(video_inputs)[1], int(lens_h.max())))
        # float32 features (the reference's placeholder) or, as a torch tensor, bfloat16 ones (hual_batch.video_dtype)
        vdt = torch.bfloat16 if isinstance(video_inputs, torch.Tensor) and video_inputs.dtype == torch.bfloat16 else torch.float32
        v = self._to_dev(video_inputs, vdt)
        ln = self._to_dev(video_seq_len, torch.int32)
        w = self._to_dev(word_ids, torch.int32)
        c = self._to_dev(char_ids, torch.int32)
        B, T, V = v.shape
        if V != self.cfg.vdim:
            raise ValueError('video feature dim %d != model.vdim %d' % (V, self.cfg.vdim))
        L, C = w.shape[1], c.shape[2]
        if C < 4:   # pad_char_seq can produce < 4 chars; the reference's VALID width-4 conv would fail too
            raise ValueError('char_ids needs at least 4 chars per word')
        bt = lib.hual_batch(lib.ptr(v).value, lib.ptr(ln).value, lib.ptr(w).value, lib.ptr(c).value, B, T, L, C,
                            1 if vdt == torch.bfloat16 else 0)
        return bt, (v, ln, w, c), (B, T, L, C)

    def _outputs(self, B, T, with_loss):
        o = dict(start_logits=torch.empty(B, T, device=self.device), end_logits=torch.empty(B, T, device=self.device),
                 match_scores=torch.empty(B, T, 4, device=self.device),
                 start_index=torch.empty(B, dtype=torch.int64, device=self.device),
                 end_index=torch.empty(B, dtype=torch.int64, device=self.device))
        lt = torch.zeros(4, device=self.device) if with_loss else None
        st = lib.hual_outputs(lib.ptr(o['start_logits']).value, lib.ptr(o['end_logits']).value,
                              lib.ptr(o['match_scores']).value, lib.ptr(o['start_index']).value,
                              lib.ptr(o['end_index']).value, None if lt is None else lib.ptr(lt).value)
        return o, lt, st

    def _labels(self, y1, y2, match_labels, inner_labels):
        t = (self._to_dev(y1, torch.float32), self._to_dev(y2, torch.float32), self._to_dev(match_labels, torch.int32),
             self._to_dev(inner_labels, torch.float32))
        return lib.hual_labels(*[lib.ptr(x).value for x in t]), t

    debug_taps = False       # True: forward also writes the tensors only parity tests read (hual_run_opts.debug_taps)

    def _opts(self, drop_rate, match_denom=0.0, align_external=0):
        return lib.hual_run_opts(float(drop_rate), lib.ptr(self.rng_state).value, float(match_denom), int(align_external), 0, None,
                                 1 if self.debug_taps else 0, None, None, None)

    # ------------------------------------------------------------------ fetches
    def forward(self, video_inputs, video_seq_len, word_ids, char_ids, drop_rate=0.0, labels=None, _opts=None):
        """One pass producing start_logits, end_logits, match_scores, start_index, end_index
        (and loss terms when labels=(y1, y2, match_labels, inner_labels) is given)."""
        bt, keep, (B, T, L, C) = self._prep(video_inputs, video_seq_len, word_ids, char_ids)
        ws = self._workspace(B, T, L, C)
        o, lt, ost = self._outputs(B, T, labels is not None)
        lab_st, lab_keep = (None, None) if labels is None else self._labels(*labels)
        opts = _opts if _opts is not None else self._opts(drop_rate)
        lib.check(lib.load().hual_seqpan_forward(
            ctypes.byref(self.cfg), lib.ptr(self.params), lib.ptr(self.word_table), ctypes.byref(bt),
            None if lab_st is None else ctypes.byref(lab_st), ctypes.byref(ost), ctypes.byref(opts), lib.ptr(ws),
            ws.numel(), lib.stream_ptr()))
        if lt is not None:
            o.update(loss=lt[0], loc_loss=lt[1], match_loss=lt[2], align_loss=lt[3])
        elif self.cfg.no_gumbel == 0:
            # loss.no_gumbel false: the matching head samples gumbel noise in every evaluation of match_scores (ops.py:6-9 draws
            # fresh tf.random.uniform values per sess.run); the train step advances the Philox offset in its Adam launch, a
            # label-free forward advances it here so that consecutive evaluation batches do not share their noise
            self.rng_state[2] += 1
        self._last = (bt, keep, lab_st, lab_keep, opts)
        return o

    def backward(self):
        """tf.gradients(loss, tvars) for the batch of the last forward(labels=...) -> self.grads (flat)."""
        bt, keep, lab_st, lab_keep, opts = self._last
        assert lab_st is not None, 'backward() needs forward(labels=...)'
        ws = self._ws
        lib.check(lib.load().hual_seqpan_backward(
            ctypes.byref(self.cfg), lib.ptr(self.params), lib.ptr(self.word_table), ctypes.byref(bt),
            ctypes.byref(lab_st), ctypes.byref(opts), lib.ptr(self.grads), lib.ptr(ws), ws.numel(), lib.stream_ptr()))
        return self.grads

    def apply_gradients(self, lr, grad_prescale=1.0):
        """clip_by_global_norm + AdamWeightDecay (ops.py:119-132); lr is the fed scalar of main.py:61."""
        if isinstance(lr, torch.Tensor):
            self.lr.copy_(lr.reshape(1))
            self.lr_value = None
        elif self.lr_value != float(lr):
            self.lr.fill_(float(lr))
            self.lr_value = float(lr)
        lib.check(lib.load().hual_adamw_clip_step(
            lib.ptr(self.params), lib.ptr(self.grads), lib.ptr(self.adam_m), lib.ptr(self.adam_v), lib.ptr(self.decay),
            self.params.numel(), lib.ptr(self.lr), float(self.cfg.clip_norm), float(grad_prescale), lib.ptr(self.sqnorm),
            lib.stream_ptr()))
        self.global_step += 1

    def train_step(self, video_inputs, video_seq_len, word_ids, char_ids, y1, y2, match_labels, inner_labels, lr,
                   drop_rate):
        """sess.run([train_op, loss, start_index, end_index]) of runner_utils.py:147."""
        o = self.forward(video_inputs, video_seq_len, word_ids, char_ids, drop_rate=drop_rate,
                         labels=(y1, y2, match_labels, inner_labels))
        self.backward()
        self.apply_gradients(lr)
        self.rng_state[2] += 1
        return dict(loss=o['loss'], start_index=o['start_index'], end_index=o['end_index'])

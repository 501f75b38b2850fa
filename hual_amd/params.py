"""Flat parameter store for the HIP SeqPAN path.

The layout (names, TF shapes, offsets, weight-decay flags) comes from the library
(`hual_seqpan_param_table`, hual_amd/csrc/params.cpp) and follows the TF variable inventory of
/root/reference/models/model.py.  Initialisers restate the reference's:
TF1 `get_variable` default = glorot_uniform, ones/zeros for layer-norm scale / every bias,
orthogonal for `label_emb` (model.py:86).
"""
import math

import numpy as np

from . import lib


def _glorot_limit(shape):
    # tf init_ops._compute_fans: receptive field = prod(shape[:-2])
    if len(shape) == 1:
        fan_in = fan_out = shape[0]
    elif len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    return math.sqrt(6.0 / (fan_in + fan_out))


def init_kind(name):
    if name.endswith('layer_norm_scale'):
        return 'ones'
    if 'bias' in name.rsplit('/', 1)[-1]:
        return 'zeros'
    if name == 'label_emb':
        return 'orthogonal'
    return 'glorot'


class ParamTable:
    def __init__(self, cfg):
        self.cfg = cfg
        self.entries, self.padded, self.count = lib.param_table(cfg)
        self.by_name = {e['name']: e for e in self.entries}

    def names(self):
        return [e['name'] for e in self.entries]

    def init_flat(self, seed=12345):
        """numpy float32 [padded] with the reference initialisers."""
        g = np.random.default_rng(seed)
        flat = np.zeros(self.padded, dtype=np.float32)
        for e in self.entries:
            shape, kind = e['shape'], init_kind(e['name'])
            if kind == 'ones':
                a = np.ones(shape)
            elif kind == 'zeros':
                a = np.zeros(shape)
            elif kind == 'orthogonal':
                m = g.standard_normal((shape[1], shape[0]))
                q, r = np.linalg.qr(m)
                a = (q * np.sign(np.diag(r))).T[:shape[0]]
            else:
                lim = _glorot_limit(shape)
                a = g.uniform(-lim, lim, size=shape)
            flat[e['offset']:e['offset'] + e['size']] = np.asarray(a, dtype=np.float32).reshape(-1)
        return flat

    def decay_flat(self, rate=0.01):
        """per-element weight-decay rate (ops.py:121-123: everything except LayerNorm|layer_norm|bias)."""
        d = np.zeros(self.padded, dtype=np.float32)
        for e in self.entries:
            if e['decay']:
                d[e['offset']:e['offset'] + e['size']] = rate
        return d

    def pack(self, named):
        """dict name -> array (TF shapes) -> flat numpy float32"""
        flat = np.zeros(self.padded, dtype=np.float32)
        for e in self.entries:
            a = np.asarray(named[e['name']], dtype=np.float32)
            assert list(a.shape) == e['shape'], (e['name'], a.shape, e['shape'])
            flat[e['offset']:e['offset'] + e['size']] = a.reshape(-1)
        return flat

    def unpack(self, flat):
        """flat array-like -> dict name -> numpy array in TF shape"""
        flat = np.asarray(flat)
        return {e['name']: flat[e['offset']:e['offset'] + e['size']].reshape(e['shape']).copy() for e in self.entries}

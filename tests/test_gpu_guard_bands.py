"""GPU: no kernel writes outside the memory the caller handed over.  Every buffer of a train step - parameters, gradient bucket, both Adam slots, the
workspace at EXACTLY the size hual_seqpan_query_workspace reports, the feeds, the labels, the fetch tensors - is carved out of one arena with 64 KB
guard bands of a known byte on both sides; after forward / backward / optimizer at ragged and tile-boundary shapes every guard band must be intact.
(The GPU box has no address sanitizer; the workspace is otherwise over-allocated by a cache line and torch's allocator rounds every tensor up, so
a write a few bytes past an end would go unnoticed by the parity tests.)"""
import numpy as np
import pytest
import torch

import parity_util as pu
from hual_amd import lib

pytestmark = pytest.mark.gpu
GUARD, BYTE = 64 * 1024, 0xA5


class Arena:
    def __init__(self, nbytes, dev):
        self.buf = torch.full((nbytes,), BYTE, dtype=torch.uint8, device=dev)
        assert self.buf.data_ptr() % 256 == 0
        self.off, self.guards = GUARD, [(0, GUARD)]

    def take(self, nbytes, dtype=torch.uint8, shape=None):
        n = (int(nbytes) + 255) // 256 * 256
        v = self.buf[self.off:self.off + int(nbytes)]
        self.guards.append((self.off + int(nbytes), self.off + n + GUARD))      # (the padding up to the next 256 bytes belongs to the guard)
        self.off += n + GUARD
        assert self.off <= self.buf.numel(), 'arena too small'
        v = v.view(dtype)
        return v.view(shape) if shape is not None else v

    def check(self, what):
        for a, b in self.guards:
            bad = (self.buf[a:b] != BYTE).nonzero()
            assert bad.numel() == 0, '%s: %d bytes of the guard band at arena offset %d were overwritten (first at +%d)' % (what, bad.numel(), a, int(bad[0]))


SHAPES = [dict(B=3, T=20, L=6, C=5, vdim=64, max_vlen=32), dict(B=5, T=131, L=33, C=9, vdim=320, max_vlen=160), dict(B=2, T=256, L=20, C=8, vdim=64, max_vlen=256),
          dict(B=7, T=17, L=3, C=4, vdim=64, max_vlen=32), dict(B=4, T=65, L=40, C=22, vdim=64, max_vlen=128), dict(B=1, T=5, L=3, C=4, vdim=64, max_vlen=16)]


@pytest.mark.parametrize('variant', ['f32', 'bf16 feed', 'gumbel'])
@pytest.mark.parametrize('shape', SHAPES, ids=lambda s: 'B%d_T%d_L%d_C%d' % (s['B'], s['T'], s['L'], s['C']))
def test_no_write_outside_the_callers_buffers(shape, variant):
    if variant != 'f32' and shape is not SHAPES[1] and shape is not SHAPES[3]:
        pytest.skip('the feed / loss variants run at two shapes')
    cfg, p, wv, b, labels = pu.make_case(seed=9, **shape)
    if variant == 'gumbel':
        cfg['no_gumbel'] = False          # loss.no_gumbel false: gumbel noise in the matching head (layers.py:163-166)
        cfg['tau'] = 0.3
    m = pu.hip_model(cfg, p, wv)
    B, T, L, C, V = shape['B'], shape['T'], shape['L'], shape['C'], shape['vdim']
    vdt, vb = (torch.bfloat16, 2) if variant == 'bf16 feed' else (torch.float32, 4)
    need = lib.query_workspace(m.cfg, B, T, L, C)                 # exactly what the ABI asks for: no slack
    n = m.params.numel()
    total = 25 * (GUARD + 512) + need + 4 * n * 4 + B * T * V * 4 + B * L * (C + 1) * 4 + 8 * B * T * 4 + 4096
    ar = Arena(total + (1 << 20), m.device)
    f32, i32 = torch.float32, torch.int32
    for name in ('params', 'grads', 'adam_m', 'adam_v'):
        t = ar.take(n * 4, f32)
        t.copy_(getattr(m, name))
        setattr(m, name, t)
    m._ws = ar.take((need + 255) // 256 * 256)                    # (the facade adds 256 bytes of its own to the query: still inside this slice)
    assert m._ws.data_ptr() % 256 == 0
    m._ws_need[(B, T, L, C)] = need
    m._ws_tables[(B, T, L, C)] = lib.ws_table(m.cfg, B, T, L, C)
    video = ar.take(B * T * V * vb, vdt, (B, T, V)); video.copy_(b['video'])      # (bfloat16: hual_batch.video_dtype, exactly half the bytes)
    lens = ar.take(B * 4, i32, (B,)); lens.copy_(b['lens'])
    words = ar.take(B * L * 4, i32, (B, L)); words.copy_(b['word_ids'])
    chars = ar.take(B * L * C * 4, i32, (B, L, C)); chars.copy_(b['char_ids'])
    lab = []
    for t, dt in zip(labels, (f32, f32, i32, f32)):
        v = ar.take(B * T * 4, dt, (B, T)); v.copy_(t.to(dt)); lab.append(v)
    outs = dict(start_logits=ar.take(B * T * 4, f32, (B, T)), end_logits=ar.take(B * T * 4, f32, (B, T)),
                match_scores=ar.take(B * T * 16, f32, (B, T, 4)), start_index=ar.take(B * 8, torch.int64, (B,)),
                end_index=ar.take(B * 8, torch.int64, (B,)))
    loss_terms = ar.take(16, f32, (4,))

    def _outputs(B_, T_, with_loss):      # the facade's fetch tensors, from the arena
        st = lib.hual_outputs(*[lib.ptr(outs[k]).value for k in ('start_logits', 'end_logits', 'match_scores', 'start_index', 'end_index')],
                              lib.ptr(loss_terms).value if with_loss else None)
        return outs, (loss_terms if with_loss else None), st
    m._outputs = _outputs
    ar.check('set-up')
    m.set_rng(3, 1)
    for rep, drop in enumerate((0.0, 0.2)):
        m.forward(video, lens, words, chars, drop_rate=drop)                       # label-free pass (evaluation fetch set)
        torch.cuda.synchronize(); ar.check('forward without labels, dropout %.1f' % drop)
        o = m.forward(video, lens, words, chars, drop_rate=drop, labels=tuple(lab))
        torch.cuda.synchronize(); ar.check('forward with labels, dropout %.1f' % drop)
        m.backward()
        torch.cuda.synchronize(); ar.check('backward, dropout %.1f' % drop)
        m.apply_gradients(1e-4)
        torch.cuda.synchronize(); ar.check('clip + AdamWD')
        assert torch.isfinite(o['loss']).all()
    assert torch.isfinite(m.params).all() and int(outs['start_index'].min()) >= 0


def test_batch_assembly_stays_inside_its_feed_tensors():
    """hual_assemble_batch / hual_assemble_batch_cursor (the loaders' process_batch on the device) into feed tensors of EXACTLY the batch's padded
    shape, guard bands around each: nothing outside is touched, and the contents equal an assembly into ordinary tensors"""
    import al_synth
    from hual_amd import al
    from hual_amd.dataset import DeviceDataset
    recs, vis, data_gt, _ = al_synth.make_trainset(60, 12, 96, 40, seed=4, max_words=17)
    ds = DeviceDataset(recs, vis)
    s0, e0 = al.labels_from_times(data_gt, ds.vlen_h)
    ds.set_labels(s0, e0)
    g = np.random.default_rng(2)
    f32, i32 = torch.float32, torch.int32
    for rep in range(6):
        sel = g.choice(60, size=int(g.integers(1, 9)), replace=False).astype(np.int32)
        B = len(sel)
        T, L, C = ds.batch_shape(sel)
        C = max(C, 4)
        ar = Arena(12 * (GUARD + 512) + B * T * 96 * 4 + B * L * (C + 1) * 4 + 5 * B * T * 4 + 65536, ds.dev)
        out = dict(video=ar.take(B * T * 96 * 4, f32, (B, T, 96)), video_seq_len=ar.take(B * 4, i32, (B,)), word_ids=ar.take(B * L * 4, i32, (B, L)),
                   char_ids=ar.take(B * L * C * 4, i32, (B, L, C)), sel=ar.take(B * 4, i32, (B,)), y1=ar.take(B * T * 4, f32, (B, T)),
                   y2=ar.take(B * T * 4, f32, (B, T)), match_labels=ar.take(B * T * 4, i32, (B, T)), inner_labels=ar.take(B * T * 4, f32, (B, T)))
        got = ds.assemble(sel, out=out, min_chars=4)
        torch.cuda.synchronize()
        ar.check('hual_assemble_batch, batch %s' % (sel.tolist(),))
        ref = ds.assemble(sel, min_chars=4)
        for k in ('video', 'video_seq_len', 'word_ids', 'char_ids', 'y1', 'y2', 'match_labels', 'inner_labels'):
            assert got[k].data_ptr() == out[k].data_ptr() and torch.equal(got[k], ref[k]), k
        # the cursor form: the same batch as ids[2 .. 2 + B) of a device-side list
        ids = torch.zeros(2 + B + 3, dtype=i32, device=ds.dev)
        ids[2:2 + B] = torch.from_numpy(sel).to(ds.dev)
        cursor = torch.tensor([2, 0], dtype=torch.int64, device=ds.dev)
        for k in out:
            if k != 'sel':
                out[k].fill_(7)
        ds.enqueue_assemble_cursor({k: v for k, v in out.items() if k != 'sel'}, ids, cursor)
        torch.cuda.synchronize()
        ar.check('hual_assemble_batch_cursor, batch %s' % (sel.tolist(),))
        for k in ('video', 'video_seq_len', 'word_ids', 'char_ids', 'y1', 'y2', 'match_labels', 'inner_labels'):
            assert torch.equal(out[k], ref[k]), k

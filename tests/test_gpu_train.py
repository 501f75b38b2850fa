"""GPU: whole training steps (forward + backward + clip_by_global_norm + AdamWeightDecay) against the CPU oracle's
train_step (ops.py:119-174), eager and as a replayed hipGraph."""
import collections
import os

import numpy as np
import pytest
import torch

import parity_util as pu
from oracle import seqpan_ref as R

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('use_graph,nsteps', [(False, 3), (True, 30)])
def test_train_steps_match_oracle(use_graph, nsteps):
    """3 (eager) / 30 (replayed hipGraph) consecutive steps on one batch; after every HIP step the oracle takes the same step
    from the same state with the ReLU active sets of that HIP forward (parity_util.relu_pins, audited), so loss, spans and
    parameters are comparable tightly at EVERY step of the trajectory (the free-running comparison is the next test)"""
    from hual_amd.train import Trainer
    # the short run uses a 10x learning rate; the long one the reference's (configs/*/SeqPAN.yaml: 1e-4) - at 1e-3 this tiny batch
    # diverges (loss 10 -> 130 within 15 steps) and ill-conditioned pre-activations then sit within rounding of the ReLU kink
    lr, drop, seed, off = (1e-3 if nsteps <= 3 else 1e-4), 0.2, 99, 5
    cfg, p, wv, b, labels = pu.make_case(B=4, T=24, L=7, C=5, seed=21)
    B, T, L = 4, 24, 7
    m = pu.hip_model(cfg, p, wv)
    m.set_rng(seed, off)
    tr = Trainer(m, world=1, use_graph=use_graph)
    tr.set_batch(b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy(),
                 *[x.numpy() for x in labels])
    rp = collections.OrderedDict((k, v.clone()) for k, v in p.items())
    rm = {k: torch.zeros_like(v) for k, v in p.items()}
    rv = {k: torch.zeros_like(v) for k, v in p.items()}
    batch = (b['video'], b['lens'], b['word_ids'], b['char_ids'])
    for s in range(nsteps):
        prev = {k: v.clone() for k, v in rp.items()}
        tr.step(lr=lr, drop_rate=drop)
        torch.cuda.synchronize()
        pins = pu.relu_pins(m, B, T, L)
        rp, rm, rv, info = R.train_step(rp, rm, rv, cfg, wv, batch, labels, lr, drop, seed=seed, offset=off + s, relu_pin=pins,
                                        want_tap=True)
        pu.audit_pins(info['tap'], pins)          # the pins are the kernels' own active sets: rounding-level disagreements only
        np.testing.assert_allclose(float(tr.last_loss()), float(info['loss']), rtol=1e-3, atol=1e-3)
        assert torch.equal(tr.start_index.cpu(), info['start_index']) and torch.equal(tr.end_index.cpu(), info['end_index'])
        got = m.state_dict()
        # AdamWeightDecay without bias correction moves a weight by ~lr * m / sqrt(v): about 3.16 * lr in step 1 whatever the
        # size of its gradient, so an entry whose gradient is pure rounding noise may land anywhere within that step
        for k, v in rp.items():
            moved = float(np.abs(v.numpy() - prev[k].numpy()).max())
            d = float(np.abs(got[k] - v.numpy()).max())
            assert d < 2e-5 + 0.02 * moved, (s, k, d, moved)
        # continue the oracle from the HIP parameters so that step s+1 compares ONE step, not an accumulated drift
        rp = collections.OrderedDict((k, torch.from_numpy(got[k])) for k in rp)
        rm = {k: torch.from_numpy(a) for k, a in m.table.unpack(m.adam_m.cpu().numpy()).items()}
        rv = {k: torch.from_numpy(a) for k, a in m.table.unpack(m.adam_v.cpu().numpy()).items()}


def _free_run(cfg, p, wv, b, labels, dtype, lr, drop, seed, off, steps):
    rp = collections.OrderedDict((k, v.clone().to(dtype)) for k, v in p.items())
    rm = {k: torch.zeros_like(v) for k, v in rp.items()}
    rv = {k: torch.zeros_like(v) for k, v in rp.items()}
    batch = (b['video'].to(dtype), b['lens'], b['word_ids'], b['char_ids'])
    lab = tuple(x.to(dtype) if x.dtype.is_floating_point else x for x in labels)
    out = []
    for s in range(steps):
        rp, rm, rv, info = R.train_step(rp, rm, rv, cfg, wv.to(dtype), batch, lab, lr, drop, seed=seed, offset=off + s)
        out.append((float(info['loss']), info['start_logits'].double(), info['end_logits'].double(),
                    info['start_index'], info['end_index']))
    return out


def test_twenty_step_trajectory_against_free_running_oracle():
    """14 consecutive steps on one batch (20 in round 5, 30 before: the envelope below is vacuous from step ~12 on - 1e-3 . 2.5^12 = 60 - and the
    sixty CPU-oracle steps of the two reference trajectories were 180 s of a GPU suite that has to finish in 900 s on any box) at the reference's settings (lr 1e-4, dropout 0.2: configs/charades/SeqPAN.yaml).
    The oracle runs ON ITS OWN - its own parameters, Adam slots and ReLU signs, never re-seeded from the HIP state; only the
    dropout stream (seed, step) is shared.

    What two correct implementations can agree on is bounded by the optimizer, not by the kernels: AdamWeightDecay without
    bias correction (ops.py:149-174) moves a weight by ~lr.m/sqrt(v) = O(lr) whatever the size of its gradient, so
    rounding-level gradient differences become O(lr) parameter differences and the trajectories separate exponentially
    (1.3x - 2x per step for this batch, depending on the dropout sample path and on the machine's float32 kernels: the
    float32 PyTorch oracle run in the build container left the float64 oracle's trajectory by 0.64 in the logits at step 17).
    The test runs all three trajectories, prints them side by side, and demands of the HIP path:
      * steps 0-3: loss within 1e-3 relative of the float64 oracle's and the spans EQUAL;
      * steps 4-7: loss within 1e-2 relative (1e-2 absolute below 1);
      * all 14 steps: logit deviation from the float64 trajectory inside the envelope 1e-3 . 1.5^step (the HIP path injects
        more rounding per step than float32 PyTorch - split-bf16 weight-gradient and attention-backward products, 2^-16 per
        product - and the printed table shows both deviations side by side);
      * the float64 loss actually falls (an optimizer that does nothing would pass the rest).
    Step-by-step agreement over 30 steps (same state on both sides at every step) is test_train_steps_match_oracle."""
    from hual_amd.train import Trainer
    lr, drop, seed, off, steps = 1e-4, 0.2, 31, 11, 14      # (20 in round 5: sixty CPU-oracle steps were a quarter of the GPU suite)
    cfg, p, wv, b, labels = pu.make_case(B=4, T=24, L=7, C=5, seed=21)
    m = pu.hip_model(cfg, p, wv)
    m.set_rng(seed, off)
    tr = Trainer(m, world=1, use_graph=True)
    tr.set_batch(b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy(),
                 *[x.numpy() for x in labels])
    o64 = _free_run(cfg, p, wv, b, labels, torch.float64, lr, drop, seed, off, steps)
    # the float32 oracle's own trajectory is printed beside the two (nothing is asserted on it) - only on request: its steps were a
    # quarter of the GPU suite's run time; bench.py prints the same three-way comparison on its batch (cpu_baseline.loss_trajectory)
    o32 = _free_run(cfg, p, wv, b, labels, torch.float32, lr, drop, seed, off, steps) if os.environ.get('HUAL_TEST_F32_TRAJ') == '1' else o64

    def dev(a, ref):
        return max(float((a[1] - ref[1]).abs().max()), float((a[2] - ref[2]).abs().max()))

    dev_h, dev_o = [], []
    for s in range(steps):
        tr.step(lr=lr, drop_rate=drop)
        torch.cuda.synchronize()
        hip = (float(tr.last_loss()), tr.start_logits.cpu().double(), tr.end_logits.cpu().double())
        dev_h.append(dev(hip, o64[s]))
        dev_o.append(dev(o32[s], o64[s]))
        print('step %2d  loss hip %.5f  f64 %.5f  f32 %.5f   logit deviation from f64: hip %.2e  f32 oracle %s' %
              (s, hip[0], o64[s][0], o32[s][0], dev_h[-1], '%.2e' % dev_o[-1] if o32 is not o64 else 'not run'))
        if s < 4:
            assert abs(hip[0] - o64[s][0]) <= 1e-3 * max(abs(o64[s][0]), 1.0), (s, hip[0], o64[s][0])
            assert torch.equal(tr.start_index.cpu(), o64[s][3]) and torch.equal(tr.end_index.cpu(), o64[s][4]), s
        elif s < 8:
            assert abs(hip[0] - o64[s][0]) <= 1e-2 * max(abs(o64[s][0]), 1.0), (s, hip[0], o64[s][0])
        assert dev_h[s] <= 1e-3 * 1.5 ** s, (s, dev_h[s])      # (round 6: 2.5^s until then; measured 3.3e-3 at step 13 against 0.19)
    assert o64[-1][0] < o64[0][0] - 1.0


def test_c1_trajectory_matches_the_clean_fp64_oracle():
    """The free-running comparison at BASELINE configs[0] (B16 T64 vdim1024 L20 C8, lr 1e-4, dropout 0.2 - the shape and settings
    of bench.py's cpu_baseline.loss_trajectory).  All three runs start from the same parameters and ZERO Adam slots, share only the
    dropout stream, and never exchange state.

    What sets the drift: AdamWeightDecay has no bias correction (ops.py:149-174), so the first updates are lr * 3.16 * sign(g) for
    EVERY element - an element whose gradient is small against the rounding noise of its tensor moves a full step in a direction
    the noise decides.  Round 4's kernels (attention and context-query backward on bf16 pairs, 16-bit operands) put every gradient
    tensor 2-4e-5 of its maximum from the float64 oracle's and this trajectory left it by 1.8e-3 at step 9.  Round 5 (fp16 pairs, 22
    bits, in the attention forward + backward and the context-query backward; the [B,B] alignment similarity in double): median
    5.3e-6 of each tensor's maximum against 4.6e-6 for float32 PyTorch (profiles/r5_grad_error_c1.txt), and on this sample path the
    HIP loss stays within 1.2e-6 of float64 through step 7 and within 4e-5 through step 9 (float32 PyTorch: 8e-7).  The test demands:
    steps 0-2 within 1e-6 (the forward's precision), steps 0-6 within 1e-5, ALL TEN steps within 1e-3 (the 5e-3 of round 4 is gone),
    a falling loss, and the spans EQUAL to the float64 oracle's while the loss agrees to 1e-5."""
    from hual_amd.train import Trainer
    lr, drop, seed, off, steps = 1e-4, 0.2, 1, 1, 10
    cfg, p, wv, b, labels = pu.make_case(B=16, T=64, L=20, C=8, seed=12345, max_vlen=64, vdim=1024, num_words=1000)
    m = pu.hip_model(cfg, p, wv)
    m.set_rng(seed, off)
    tr = Trainer(m, world=1, use_graph=True)
    tr.set_batch(b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy(),
                 *[x.numpy() for x in labels])
    o64 = _free_run(cfg, p, wv, b, labels, torch.float64, lr, drop, seed, off, steps)
    o32 = _free_run(cfg, p, wv, b, labels, torch.float32, lr, drop, seed, off, steps) if os.environ.get('HUAL_TEST_F32_TRAJ') == '1' else o64
    worst_h = worst_o = 0.0
    first = last = None
    for s in range(steps):
        tr.step(lr=lr, drop_rate=drop)
        torch.cuda.synchronize()
        hl = float(tr.last_loss())
        first = hl if first is None else first
        last = hl
        den = max(abs(o64[s][0]), 1.0)
        rel_h, rel_o = abs(hl - o64[s][0]) / den, abs(o32[s][0] - o64[s][0]) / den
        worst_h, worst_o = max(worst_h, rel_h), max(worst_o, rel_o)
        print('step %2d  loss hip %.5f  f32 %.5f  f64 %.5f   rel to f64: hip %.2e  f32 oracle %.2e' % (s, hl, o32[s][0], o64[s][0], rel_h, rel_o))
        assert rel_h <= 1e-3, (s, hl, o64[s][0])
        if s < 7:
            assert rel_h <= 1e-5, (s, hl, o64[s][0])
        if s < 3:
            assert rel_h <= 1e-6, (s, hl, o64[s][0])
        if worst_h <= 1e-5:
            assert torch.equal(tr.start_index.cpu(), o64[s][3]) and torch.equal(tr.end_index.cpu(), o64[s][4]), s
    assert last < first - 5.0 and o64[-1][0] < o64[0][0] - 5.0
    print('c1 free-running trajectory over %d steps: worst relative loss difference to float64 - HIP %.2e, float32 oracle %.2e' % (steps, worst_h, worst_o))


def test_weight_decay_mask_and_clip_on_device():
    from hual_amd import lib
    m = pu.hip_model(*pu.make_case()[:3])
    n = m.params.numel()
    g = torch.Generator().manual_seed(0)
    m.grads.copy_(torch.randn(n, generator=g))
    p0 = m.params.clone()
    m.apply_gradients(0.5)
    torch.cuda.synchronize()
    gn = float(m.grads.norm())
    gc = m.grads * (1.0 / max(gn, 1.0))
    nm, nv = 0.1 * gc, 0.001 * gc * gc
    upd = nm / (nv.sqrt() + 1e-6) + m.decay * p0
    exp = p0 - 0.5 * upd
    assert float((m.params - exp).abs().max()) < 5e-5
    assert float((m.adam_m - nm).abs().max()) < 1e-7


def test_backward_zeroes_the_bucket_unless_the_forward_left_a_receipt():
    """hual_run_opts.prezero_token: the forward's first launch zeroes the gradient bucket and leaves a host-side receipt; a second
    backward call on the same forward finds no receipt and zeroes the bucket itself - gradients are not accumulated twice"""
    from hual_amd.train import Trainer
    cfg, p, wv, b, labels = pu.make_case(B=3, T=20, L=6, C=5, seed=5)
    feeds = (b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy(), *[x.numpy() for x in labels])
    m = pu.hip_model(cfg, p, wv)
    m.set_rng(3, 1)
    tr = Trainer(m, world=1, use_graph=False)
    tr.set_batch(*feeds)
    opts = tr._opts(0.1, 0)
    m.grads.fill_(7.0)                               # garbage the forward has to clear
    tr._forward(opts)
    assert tr._prezero_token.value == m.grads.data_ptr()
    tr._backward(opts)
    assert tr._prezero_token.value == 0
    g1 = m.grads.detach().clone()
    tr._backward(opts)                               # no receipt: zeroes, then accumulates the same gradient again
    g2 = m.grads.detach().clone()
    torch.cuda.synchronize()
    assert torch.isfinite(g1).all() and float(g1.abs().max()) < 1e3
    scale = float(g1.abs().max())
    assert float((g1 - g2).abs().max()) <= 1e-5 * scale      # (atomics: the summation order differs between the two launches)


def test_static_step_graph_survives_other_users_of_the_workspace():
    """A Trainer fed by set_batch replays ONE graph with static job tables (hual_run_opts.static_tables).  The model's workspace is
    shared: an evaluation forward at another shape (or a second Trainer) runs in the same allocation in between.  The table of the
    weight-gradient launch is the trainer's own (hual_run_opts.dw_table), so the replay still computes the same gradients."""
    from hual_amd.train import Trainer
    cfg, p, wv, b, labels = pu.make_case(B=4, T=24, L=7, C=5, seed=21)
    feeds = (b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy(), *[x.numpy() for x in labels])
    # undisturbed run: two steps at lr 0 (parameters stay), the gradient of the second replay
    m0 = pu.hip_model(cfg, p, wv)
    t0 = Trainer(m0, world=1, use_graph=True)
    t0.set_batch(*feeds)
    for _ in range(3):
        t0.step(lr=0.0, drop_rate=0.0)
    torch.cuda.synchronize()
    want = m0.grads.clone()
    # disturbed run: a forward of a SMALLER batch (its buffers lie over the table's old place in the workspace) and a second
    # trainer at yet another shape between the replays
    m1 = pu.hip_model(cfg, p, wv)
    m1.ws_poison = None
    t1 = Trainer(m1, world=1, use_graph=True)
    t1.set_batch(*feeds)
    t1.step(lr=0.0, drop_rate=0.0)
    t1.step(lr=0.0, drop_rate=0.0)
    cfg2, _, _, b2, labels2 = pu.make_case(B=2, T=12, L=4, C=4, seed=5)
    m1.forward(b2['video'].numpy(), b2['lens'].numpy(), b2['word_ids'].numpy(), b2['char_ids'].numpy())
    t2 = Trainer(m1, world=1, use_graph=False)
    t2.set_batch(b2['video'].numpy(), b2['lens'].numpy(), b2['word_ids'].numpy(), b2['char_ids'].numpy(), *[x.numpy() for x in labels2])
    t2.step(lr=0.0, drop_rate=0.0)
    torch.cuda.synchronize()
    m1._ws.fill_(0xFF)                                            # and whatever else was left there
    m1._workspace(4, 24, 7, 5)
    t1.step(lr=0.0, drop_rate=0.0)                                # replay of the graph captured before
    torch.cuda.synchronize()
    assert torch.isfinite(m1.grads).all()
    scale = max(1.0, float(want.abs().max()))
    assert float((m1.grads - want).abs().max()) <= 1e-5 * scale   # (float atomics: not bit-reproducible)

#!/usr/bin/env python3
"""Headline benchmark: SeqPAN train clips/sec on MI355X (BASELINE.json metric).

A "step" = one sess.run([train_op, loss, start_index, end_index]) of the reference
(/root/reference/utils/runner_utils.py:147): forward + backward + clip_by_global_norm + AdamWeightDecay on one
synthetic batch that is already resident in HBM, with the reference's training dropout (train.droprate 0.2).
Workload at N=1: BASELINE.json configs[1]  (batch 64, T=128, vdim 1024, L=20, C=8, dim 128, 8 heads, 2 attention
layers).  N>1: one process per GPU (torch.distributed, RCCL), weak scaling (64 clips per GPU), gradients averaged
with one flat all-reduce per step.

Prints ONE JSON line on rank 0 (contract in the task description) with two extra objects:
  roofline     - dominant kernel (by total time), every launch timed by its own dispatch events (hual_prof_*)
  cpu_baseline - the CPU oracle (oracle/seqpan_ref.py, "port") timed on this host on a bounded sample
"""
import argparse
import ctypes
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MATRIX_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2_f32
PEAK_BF16_MATRIX_TFLOPS = 2516.8  # MI355X_MICROARCH.md: bf16 MFMA = 16 x the fp32 matrix rate (~2.5 PF dense)
PEAK_HBM_GBS = 8000.0
PEAK_HBM_ACHIEVABLE_GBS = 6290.0   # MI355X_MICROARCH.md: what a streaming copy kernel reaches on this part (0.79 of the spec figure)
def _traffic_file():
    """newest committed per-kernel HBM-traffic summary (scripts/pmc_traffic.py: separate FETCH_SIZE / WRITE_SIZE passes)"""
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')))
    return os.path.basename(fs[-1]) if fs else 'r1f_pmc_traffic.json'


TRAFFIC_FILE = _traffic_file()
MFLOP_PER_CLIP_C2 = 1126.5               # fwd+bwd, T=128 vdim=1024 L=20 C=8 (72.1 GFLOP per B=64 step)
FEATURE_LOAD_KERNEL = 'feature_ksplit_kernel'


PIPE_NAMES = {0: 'none', 1: 'fp32 matrix (v_mfma_f32_*_f32)', 2: '16-bit matrix (v_mfma_f32_*_f16 / _bf16), split operands'}
CLOCK_MHZ = 2400.0                # MI355X_MICROARCH.md: engine clock, for cycles <-> microseconds of the SQ counter figures
N_SIMD = 1024                     # 256 CUs x 4 SIMDs
DEFAULT_SHAPE_KEY = 'B64 T128 L20 C8 vdim1024 drop0.2 f32'


def mfma_peak(kernel):
    """(peak TFLOP/s of the matrix pipe the kernel runs on, MFMA passes per algorithmic product, pipe id) - from the
    library's own per-kernel table (hual_prof_kernel_pipe, csrc/prof.cpp)"""
    from hual_amd import lib
    pipe, passes = ctypes.c_int(0), ctypes.c_int(0)
    lib.check(lib.load().hual_prof_kernel_pipe(kernel.encode(), ctypes.byref(pipe), ctypes.byref(passes)))
    peak = {0: 0.0, 1: PEAK_F32_MATRIX_TFLOPS, 2: PEAK_BF16_MATRIX_TFLOPS}[pipe.value]
    return peak, passes.value, pipe.value


def _sq_file():
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_sq_counters.txt')))
    return fs[-1] if fs else None


def sq_figures(kernel, avg_launch_us, shape_key=None):
    """committed SQ counter figures of `kernel` (newest profiles/r*_pmc_sq_counters.txt, scripts/profile_round.sh step 4):
    mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES per SIMD / cycles of a launch, wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES.
    The table is keyed by the FULL kernel name (template arguments included: ln_proj_kernel<2> and <3> are different kernels) and
    is only used when the profile was taken at this run's shape (the file's '# shape:' line; files without one are the default
    bench shape)."""
    f = _sq_file()
    if not f:
        return None
    cur, tab, shape = None, {}, DEFAULT_SHAPE_KEY
    for line in open(f):
        if line.startswith('# shape:'):
            shape = line.split(':', 1)[1].strip()
        elif line and not line[0].isspace() and 'dispatches' in line:
            cur = line[:line.rindex('dispatches')].strip()
            tab[cur] = {}
        elif cur and 'mean' in line:
            w = line.split()
            tab[cur][w[0]] = float(w[2])
    if shape_key is not None and shape_key != shape:
        return None
    if kernel not in tab or 'SQ_WAVE_CYCLES' not in tab[kernel]:
        return None
    t = tab[kernel]
    return dict(mfma_busy_frac=round(t['SQ_VALU_MFMA_BUSY_CYCLES'] / N_SIMD / (avg_launch_us * CLOCK_MHZ), 4),
                wait_frac=round(t['SQ_WAIT_ANY'] / t['SQ_WAVE_CYCLES'], 3), sq_source='profiles/' + os.path.basename(f))


def synth_batch(B, T, L, C, vdim, num_words, num_chars, seed):
    g = np.random.default_rng(seed)
    lens = g.integers((T + 1) // 2, T + 1, size=B).astype(np.int32)
    lens[g.integers(0, B)] = T
    video = g.standard_normal((B, T, vdim)).astype(np.float32)
    for b in range(B):
        video[b, lens[b]:] = 0.0
    qlens = g.integers(3, L + 1, size=B)
    qlens[g.integers(0, B)] = L
    word_ids = np.zeros((B, L), dtype=np.int32)
    char_ids = np.zeros((B, L, C), dtype=np.int32)
    for b in range(B):
        word_ids[b, :qlens[b]] = g.integers(2, num_words, size=qlens[b])
        for l in range(qlens[b]):
            cl = g.integers(1, C + 1)
            char_ids[b, l, :cl] = g.integers(1, num_chars, size=cl)
    s = np.array([g.integers(0, lens[b] - 1) for b in range(B)])
    e = np.array([g.integers(s[b] + 1, lens[b]) for b in range(B)])
    from hual_amd import data
    y1, y2, m, i = data.make_labels(s, e, lens, max_len=T)
    return dict(video=video, lens=lens, word_ids=word_ids, char_ids=char_ids, y1=y1, y2=y2, match=m,
                inner=i.astype(np.float32), s_ind=s, e_ind=e)


def cpu_baseline(seconds_budget=20.0):
    """oracle train step (fwd + bwd + AdamWD) at BASELINE configs[0] shape: B16 T64 vdim1024 fp32."""
    from oracle import seqpan_ref as R
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else os.cpu_count()
    cfg = R.default_cfg(max_vlen=64, num_words=1000)
    p0 = R.init_params(cfg, seed=12345)
    p = p0
    wv = R.init_word_vectors(cfg)
    b = synth_batch(16, 64, 20, 8, cfg.vdim, cfg.num_words, cfg.num_chars, 12345)
    batch = (torch.tensor(b['video']), torch.tensor(b['lens']), torch.tensor(b['word_ids']), torch.tensor(b['char_ids']))
    labels = (torch.tensor(b['y1']), torch.tensor(b['y2']), torch.tensor(b['match']), torch.tensor(b['inner']))
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v = {k: torch.zeros_like(t) for k, t in p.items()}
    # thread count: the oracle's matrices are small ([1024,128] x [128,128]), more threads than that work feeds are slower -
    # time one step at each candidate count up to ALL affinity cores and keep the fastest (the scan is reported)
    # (ascending; the scan stops at the first count that is clearly slower than the best so far - oversubscribed counts take many
    #  seconds per step - and every candidate prints a progress line: a silent bench is taken for a hung one)
    # (R.train_step returns NEW parameter / slot dicts and never writes into the ones passed in - tests/test_oracle.py - so the
    #  discarded warm-up and scan steps leave p, m, v at their initial values: round 3's trajectory started from leaked slots)
    scan = {}
    torch.set_num_threads(max(1, min(ncores, 16)))
    R.train_step(p, m, v, cfg, wv, batch, labels, 1e-4, 0.2, seed=1, offset=0)          # warm-up (result discarded)
    for nt in sorted({min(ncores, c) for c in (16, 32, 64, 128, ncores)}):
        torch.set_num_threads(max(1, nt))
        t0 = time.perf_counter()
        R.train_step(p, m, v, cfg, wv, batch, labels, 1e-4, 0.2, seed=1, offset=0)
        scan[nt] = round(time.perf_counter() - t0, 3)
        print('[bench] cpu baseline: %d threads %.3f s/step' % (nt, scan[nt]), file=sys.stderr, flush=True)
        if scan[nt] > 1.3 * min(scan.values()):
            break
    best = min(scan, key=scan.get)
    torch.set_num_threads(best)
    print('[bench] cpu baseline on %d threads (affinity %d, cpu_count %d; s/step by thread count: %s)'
          % (best, ncores, os.cpu_count(), scan), file=sys.stderr, flush=True)
    n, t0 = 0, time.perf_counter()
    losses = []
    while True:
        p, m, v, info = R.train_step(p, m, v, cfg, wv, batch, labels, 1e-4, 0.2, seed=1, offset=n + 1)
        losses.append(float(info['loss']))
        n += 1
        dt = time.perf_counter() - t0
        if n % 5 == 0:
            print('[bench] cpu baseline: %d steps, %.1f s' % (n, dt), file=sys.stderr, flush=True)
        if dt > seconds_budget or n >= 200:
            break
    out = dict(value=round(16 * n / dt, 2), unit='clips/s', cores=best, kind='port',
               sample='%d train steps (fwd+bwd+AdamWD, dropout 0.2) of the PyTorch-CPU oracle at B16 T64 vdim1024 L20 fp32, %.1f s'
                      % (n, dt), affinity_cores=ncores, seconds_per_step_by_threads=scan)
    # the forward alone (the reference's evaluation fetches, runner_utils.py:166) and the CPU's name beside the numbers
    try:
        with torch.no_grad():
            R.forward(p0, cfg, wv, batch[0], batch[1], batch[2], batch[3], drop_rate=0.0)
            nf, tf0 = 0, time.perf_counter()
            while nf < 40 and time.perf_counter() - tf0 < 4.0:
                R.forward(p0, cfg, wv, batch[0], batch[1], batch[2], batch[3], drop_rate=0.0)
                nf += 1
            out['forward_only'] = dict(value=round(16 * nf / (time.perf_counter() - tf0), 2), unit='clips/s', passes=nf)
        model_name = [l.split(':', 1)[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name')]
        out['cpu_model'] = model_name[0] if model_name else None
        out['cpu_count'] = os.cpu_count()
    except Exception as e:      # never cost the bench line
        out['forward_only'] = dict(error=str(e)[:200])
    # the float64 oracle's first ten losses from the same clean start (untimed): the yardstick both float32 paths are read against
    k64 = min(n, 10)
    p64 = {k: t.double() for k, t in p0.items()}
    m64 = {k: torch.zeros_like(t) for k, t in p64.items()}
    v64 = {k: torch.zeros_like(t) for k, t in p64.items()}
    b64 = (batch[0].double(), batch[1], batch[2], batch[3])
    l64 = tuple(x.double() if x.dtype.is_floating_point else x for x in labels)
    losses64 = []
    for s64 in range(k64):
        p64, m64, v64, info = R.train_step(p64, m64, v64, cfg, wv.double(), b64, l64, 1e-4, 0.2, seed=1, offset=s64 + 1)
        losses64.append(float(info['loss']))
    print('[bench] cpu baseline: float64 oracle trajectory done', file=sys.stderr, flush=True)
    # the same n steps from the same initial parameters, ZERO Adam slots, batch and dropout stream on the GPU: losses side by side
    try:
        from hual_amd import lib
        from hual_amd.model import SeqPAN
        from hual_amd.train import Trainer
        hc = lib.make_cfg(vdim=cfg.vdim, dim=cfg.dim, num_heads=cfg.num_heads, word_dim=cfg.word_dim, char_dim=cfg.char_dim,
                          max_vlen=cfg.max_vlen, attn_layer=cfg.attn_layer, num_chars=cfg.num_chars, num_words=cfg.num_words,
                          match_lambda=cfg.match_lambda, clip_norm=cfg.clip_norm)
        tm = SeqPAN(hc, wv.numpy())
        tm.load_state_dict({k: t.detach().numpy() for k, t in p0.items()})
        tm.set_rng(1, 1)
        tt = Trainer(tm, world=1, use_graph=False)
        tt.set_batch(b['video'], b['lens'], b['word_ids'], b['char_ids'], b['y1'], b['y2'], b['match'], b['inner'])
        hl = []
        for _ in range(n):
            tt.step(lr=1e-4, drop_rate=0.2)
            hl.append(float(tt.last_loss()))
        k = min(n, 10)
        rel = lambda xs, ys: round(max(abs(a - c) / max(abs(a), 1.0) for a, c in zip(xs, ys)), 7)
        out['loss_trajectory'] = dict(steps=n, oracle_first=[round(x, 4) for x in losses[:k]], hip_first=[round(x, 4) for x in hl[:k]],
                                      oracle_f64_first=[round(x, 4) for x in losses64[:k]],
                                      oracle_final=round(losses[-1], 4), hip_final=round(hl[-1], 4),
                                      max_rel_diff_first=rel(losses[:k], hl[:k]),
                                      max_rel_diff_first_hip_vs_f64=rel(losses64[:k], hl[:k]),
                                      max_rel_diff_first_f32_oracle_vs_f64=rel(losses64[:k], losses[:k]),
                                      note='free-running from the same parameters and zero Adam slots: own state on each side, '
                                           'shared dropout stream')
    except Exception as e:      # never cost the bench line
        out['loss_trajectory'] = dict(error=str(e)[:200])
    # the metric's second half, "R@1 IoU=0.5": no real features exist here, so it is reported as agreement - the
    # weights the oracle just trained are loaded into the HIP model, both predict spans for the same batch, and R@1 at
    # IoU 0.5 / mIoU against the synthetic spans (runner_utils.py:25-38) is computed for both.  Same spans => same R@1.
    try:
        from hual_amd import lib
        from hual_amd.model import SeqPAN
        hc = lib.make_cfg(vdim=cfg.vdim, dim=cfg.dim, num_heads=cfg.num_heads, word_dim=cfg.word_dim, char_dim=cfg.char_dim,
                          max_vlen=cfg.max_vlen, attn_layer=cfg.attn_layer, num_chars=cfg.num_chars, num_words=cfg.num_words,
                          match_lambda=cfg.match_lambda, clip_norm=cfg.clip_norm)
        hm = SeqPAN(hc, wv.numpy())
        hm.load_state_dict({k: v.detach().numpy() for k, v in p.items()})
        ho = hm.forward(b['video'], b['lens'], b['word_ids'], b['char_ids'], drop_rate=0.0)
        ro = R.forward(p, cfg, wv, batch[0], batch[1], batch[2], batch[3], drop_rate=0.0)
        hs, he = ho['start_index'].cpu().numpy(), ho['end_index'].cpu().numpy()
        rs, re_ = ro['start_index'].numpy(), ro['end_index'].numpy()

        def r1(si, ei):
            ious = []
            for k in range(len(si)):
                lo, hi = min(si[k], b['s_ind'][k]), max(ei[k] + 1, b['e_ind'][k] + 1)
                inter = min(ei[k] + 1, b['e_ind'][k] + 1) - max(si[k], b['s_ind'][k])
                ious.append(max(0.0, inter / (hi - lo)))
            a = np.asarray(ious)
            return round(float(np.mean(a >= 0.5) * 100.0), 2), round(float(a.mean() * 100.0), 2)
        out['r1_iou05'] = dict(hip=r1(hs, he)[0], oracle=r1(rs, re_)[0], miou_hip=r1(hs, he)[1], miou_oracle=r1(rs, re_)[1],
                               spans_equal=bool(np.array_equal(hs, rs) and np.array_equal(he, re_)),
                               note='synthetic clips, weights after the oracle steps above; frame-index IoU')
    except Exception as e:      # the agreement figure must never cost the bench line
        out['r1_iou05'] = dict(error=str(e)[:200])
    return out


def gpu_at_cpu_shape(dev, drop, steps=400):
    """the same train step on the GPU at the shape the CPU baseline is timed on (BASELINE configs[0]: B16 T64 vdim1024 L20 C8)"""
    from hual_amd import lib
    from hual_amd.model import SeqPAN
    from hual_amd.train import Trainer
    cfg = lib.make_cfg(vdim=1024, max_vlen=64, num_words=1000, num_chars=40)
    wv = np.random.default_rng(777).normal(0, 0.4, size=(998, 300)).astype(np.float32)
    model = SeqPAN(cfg, wv, device=dev, seed=12345, rng_seed=12345)
    b = synth_batch(16, 64, 20, 8, 1024, 1000, 40, 12345)
    tr = Trainer(model, world=1, use_graph=True)
    tr.set_batch(b['video'], b['lens'], b['word_ids'], b['char_ids'], b['y1'], b['y2'], b['match'], b['inner'])
    for _ in range(50):
        tr.step(lr=1e-4, drop_rate=drop)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(lr=1e-4, drop_rate=drop)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dict(value=round(16 * steps / dt, 1), unit='clips/s', ms_per_step=round(dt / steps * 1e3, 4), steps=steps,
                shape='B16 T64 vdim1024 L20 C8 (the cpu_baseline shape)')


def _resident_ms(dev, cfg, wv, B, T, L, C, vdim, drop, steps=300, warm=60):
    """ms/step of ONE resident batch of this padded shape replayed as a graph (the headline's kind of number) - the yardstick of an epoch loop"""
    from hual_amd.model import SeqPAN
    from hual_amd.train import Trainer
    model = SeqPAN(cfg, wv, device=dev, seed=12345, rng_seed=12345)
    b = synth_batch(B, T, L, C, vdim, 1000, 40, 12345)
    tr = Trainer(model, world=1, use_graph=True)
    tr.set_batch(b['video'], b['lens'], b['word_ids'], b['char_ids'], b['y1'], b['y2'], b['match'], b['inner'])
    for _ in range(warm):
        tr.step(lr=1e-4, drop_rate=drop)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(lr=1e-4, drop_rate=drop)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def _epoch_set(dev, kind, N, vdim, max_vlen, L):
    """device-resident training set for an epoch-loop leg.  kind 'synthetic': clips of T/2..T frames, queries of 3..L words
    (tests/al_synth.py make_trainset); 'anet' / 'charades': (v_len, words, longest word) drawn from the reference's OWN training
    annotations (tests/golden/lengths_<task>.npz <- /root/reference/data/<task>/train.json, scripts/gen_lengths.py), features random"""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import al_synth
    from hual_amd import al
    from hual_amd.dataset import DeviceDataset
    if kind == 'synthetic':
        recs, vis, data_gt, _ = al_synth.make_trainset(N, 512, vdim, max_vlen, seed=11, num_words=1000, num_chars=40, max_words=L)
        ds = DeviceDataset(recs, vis, device=dev)
    else:
        recs, vlens, data_gt, _ = al_synth.make_trainset_from_lengths(kind, N, vdim, max_vlen, 11, feats=False)
        total = sum(vlens[v] for v in vlens)
        bank = torch.randn(total, vdim, device=dev, generator=torch.Generator(device=dev).manual_seed(11))
        ds = DeviceDataset(recs, vlens, device=dev, feat_bank=bank)
    s0, e0 = al.labels_from_times(data_gt, ds.vlen_h)
    ds.set_labels(s0, e0)
    return ds


def epoch_loop_leg(dev, args, resident_ms, kind='synthetic', bs=None, max_vlen=None, N=None, world=1, epochs=3):
    """The reference's ACTUAL loop (runner_utils.py:139-159, data_loader.py:23-28): a shuffled epoch over a training set, every batch
    padded to its own longest clip / query / word, on the device-fed path (DeviceDataset + Trainer.run_epoch).  The FIRST epoch is
    timed on its own (every padded shape is seen for the first time: launched eagerly, then captured), a second untimed epoch
    follows, then `epochs` timed epochs - assembly launches, shape changes and the span fetch included.  world > 1: the same loop data
    parallel (every rank holds the set, trains on its shard of each global batch of bs x world clips; Trainer.run_epoch).
    resident_ms None: a resident batch of the loop's MEAN padded shape is timed as the yardstick."""
    from hual_amd import lib
    from hual_amd.model import SeqPAN
    from hual_amd.train import Trainer
    from hual_amd import dist as hdist
    bs = bs or args.batch
    max_vlen = max_vlen or args.T
    N = N or args.epoch_samples
    ds = _epoch_set(dev, kind, N, args.vdim, max_vlen, args.L)
    N = len(ds)
    Tm, Lm, Cm = ds.max_shape(4)
    cfg = lib.make_cfg(vdim=args.vdim, max_vlen=max(Tm, Lm, max_vlen), num_words=1000, num_chars=40)
    wv = np.random.default_rng(777).normal(0, 0.4, size=(998, 300)).astype(np.float32)
    model = SeqPAN(cfg, wv, device=dev, seed=12345, rng_seed=12345 + hdist.rank())
    tr = Trainer(model, world=world, use_graph=not args.no_graph, force_dp=(world == 1 and os.environ.get('HUAL_DP_FORCE_COLLECTIVES') == '1'))
    g = np.random.default_rng(0)

    def epoch():
        order = g.permutation(N)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.run_epoch(ds, order, bs, lr=1e-4, drop_rate=args.drop, min_chars=4)
        torch.cuda.synchronize()
        return order, time.perf_counter() - t0
    gb = bs * world
    nsteps_ep = (N + gb - 1) // gb
    _, first_s = epoch()                                                             # every shape new: eager launches + captures
    first_modes = dict(tr.stats)
    epoch()
    warm = dict(tr.stats)
    shapes, Ts, Ls, Cs, dt, step_shapes = set(), [], [], [], 0.0, []
    for _ in range(epochs):
        order, d = epoch()
        dt += d
        for lo in range(0, N, gb):
            ids = order[lo:lo + gb]
            shp = ds.batch_shape(ids)
            full = (min(bs, len(ids) // world if world > 1 else len(ids)),) + (shp[0], shp[1], max(4, shp[2]))
            shapes.add(full)
            step_shapes.append(full)
            Ts.append(shp[0]); Ls.append(shp[1]); Cs.append(max(4, shp[2]))
    nsteps = epochs * nsteps_ep
    timed = {k: (round(tr.stats[k] - warm[k], 4) if isinstance(tr.stats[k], float) else tr.stats[k] - warm[k]) for k in warm}
    ms = dt / nsteps * 1e3
    mT, mL, mC = int(round(np.mean(Ts))), int(round(np.mean(Ls))), int(round(np.mean(Cs)))
    yard = 'the headline batch'
    if resident_ms is None:
        resident_ms = _resident_ms(dev, cfg, wv, bs, mT, mL, mC, args.vdim, args.drop)
        yard = 'ONE resident batch of the mean padded shape B%d T%d L%d C%d, graph replay' % (bs, mT, mL, mC)
    # the yardstick that separates the LOOP from the SHAPES: resident-batch time of the padded shapes of a random sample of the timed
    # steps (a resident batch of the MEAN shape never sees a 60-word query; a third of the ActivityNet steps do)
    weighted = None
    if kind != 'synthetic':
        pick = np.random.default_rng(5).choice(len(step_shapes), size=min(40, len(step_shapes)), replace=False)
        memo = {}
        for i in pick:
            shp = step_shapes[i]
            if shp not in memo and shp[0] > 0:
                memo[shp] = _resident_ms(dev, cfg, wv, shp[0], shp[1], shp[2], shp[3], args.vdim, args.drop, steps=60, warm=20)
        ws = [memo[step_shapes[i]] for i in pick if step_shapes[i] in memo]
        if ws:
            weighted = dict(resident_ms_per_step=round(float(np.mean(ws)), 4), sampled_steps=len(ws), distinct_shapes=len(memo),
                            frac_of_resident_batch_rate=round(float(np.mean(ws)) / ms, 3),
                            note='mean resident-batch time (graph replay of ONE batch) over the padded shapes of a random sample of the timed steps')
    return dict(value=round(N * epochs / dt, 1), unit='clips/s', ms_per_step=round(ms, 4), steps=nsteps, epochs=epochs,
                resident_same_shapes=weighted, samples=N, batch_per_gpu=bs, n_gpus=world, lengths_from=kind, mean_T=round(float(np.mean(Ts)), 1), max_T=int(max(Ts)),
                mean_L=round(float(np.mean(Ls)), 1), max_L=int(max(Ls)), mean_C=round(float(np.mean(Cs)), 1),
                distinct_padded_shapes=len(shapes), graph_cache_entries=len(tr._cache),
                step_launch_modes=timed, first_epoch=dict(ms_per_step=round(first_s / nsteps_ep * 1e3, 4), steps=nsteps_ep,
                                                          step_launch_modes={k: first_modes[k] for k in ('eager', 'captured', 'replayed')}),
                resident_batch_ms_per_step=round(resident_ms, 4), resident_batch=yard,
                frac_of_resident_batch_rate=round(resident_ms / ms, 3),
                workload='shuffled epochs over an HBM-resident training set (%d samples, max_vlen %d, vdim %d): hual_assemble_batch + train '
                         'step per batch, every batch padded to its own longest clip / query / word (data parallel: to the global '
                         "batch's), spans fetched once per epoch" % (N, max_vlen, args.vdim))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=64, help='clips per GPU')
    ap.add_argument('--T', type=int, default=128)
    ap.add_argument('--L', type=int, default=20)
    ap.add_argument('--C', type=int, default=8)
    ap.add_argument('--vdim', type=int, default=1024)
    ap.add_argument('--drop', type=float, default=0.2)
    ap.add_argument('--no-graph', action='store_true', help='launch eagerly instead of replaying a hipGraph')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-epoch-loop', action='store_true', help='skip the epoch-loop leg (varying padded shapes, after the timed region)')
    ap.add_argument('--epoch-samples', type=int, default=4096)
    ap.add_argument('--anet-samples', type=int, default=8192, help='queries drawn from the ActivityNet length fixture for the epoch_loop_anet legs')
    ap.add_argument('--prewarm', type=int, default=200, help='untimed steps before the warm-up steps (clock ramp)')
    ap.add_argument('--video-dtype', choices=['f32', 'bf16'], default='f32',
                    help='element type of the clip features in HBM (hual_batch.video_dtype); arithmetic is the same')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    # rehearsal switches (tests/test_gpu_dp.py): HUAL_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and HUAL_BENCH_BACKEND=gloo moves the
    # collectives through the host, so that the N > 1 code path - the custom all-reduce leg included - runs on a one-GPU box (RCCL refuses
    # two ranks on one device).  The numbers of such a run mean nothing.
    if os.environ.get('HUAL_BENCH_ONE_DEVICE') == '1':
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    # HUAL_DP_FORCE_COLLECTIVES=1: the multi-rank code path (RCCL group, data-parallel step, rccl block, MAX-reduce of the time)
    # on however many ranks there are - one included (tests/test_gpu_dp.py runs it that way on a single GPU)
    dp = world > 1 or os.environ.get('HUAL_DP_FORCE_COLLECTIVES') == '1'
    if dp:
        import torch.distributed as dist
        backend = os.environ.get('HUAL_BENCH_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    from hual_amd import lib
    from hual_amd.model import SeqPAN
    from hual_amd.train import Trainer

    num_words, num_chars = 1000, 40
    cfg = lib.make_cfg(vdim=args.vdim, max_vlen=max(args.T, args.L), num_words=num_words, num_chars=num_chars)
    wv = np.random.default_rng(777).normal(0, 0.4, size=(num_words - 2, 300)).astype(np.float32)
    model = SeqPAN(cfg, wv, device=dev, seed=12345, rng_seed=12345 + rank)
    b = synth_batch(args.batch, args.T, args.L, args.C, args.vdim, num_words, num_chars, 12345 + rank)
    trainer = Trainer(model, world=world, use_graph=not args.no_graph, force_dp=dp)
    vdt = torch.bfloat16 if args.video_dtype == 'bf16' else torch.float32
    trainer.set_batch(b['video'], b['lens'], b['word_ids'], b['char_ids'], b['y1'], b['y2'], b['match'], b['inner'], video_dtype=vdt)

    def barrier():
        if dp:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # clock ramp: the first ~second of work after an idle period runs at reduced clocks on this part (measured:
    # 25-step runs read 5.8 ms/step, 60-step runs 4.3 ms/step for the same binary), so run --prewarm untimed steps
    # (a fixed count: every rank must run the same number of collectives) before the W warm-up steps.
    for _ in range(args.prewarm):
        trainer.step(lr=1e-4, drop_rate=args.drop)
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        trainer.step(lr=1e-4, drop_rate=args.drop)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.step(lr=1e-4, drop_rate=args.drop)
    barrier()
    dt = time.perf_counter() - t0
    if dp:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    loss = float(trainer.last_loss())
    clips = args.batch * world * args.steps

    # Everything below rides in the same line but is not the measurement: the legs report into `legs` as they finish, and a deadline
    # (HUAL_BENCH_LEG_DEADLINE_S, default 900 s after the timed region) prints the line with what has finished and ends every rank - a
    # leg that hangs (a collective one rank never reaches, a peer mapping that never answers) must not cost the timed number.
    legs = dict()
    emitted = [False]
    emit_lock = threading.Lock()

    def emit(extra=None):
        with emit_lock:
            if emitted[0]:
                return
            emitted[0] = True
            if rank == 0:
                out = dict(metric='train clips/sec', value=round(clips / dt, 2), unit='clips/s', n_gpus=world, steps=args.steps,
                           warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 4), higher_is_better=True,
                           scaling='weak', vs_baseline=None, dtype='f32 (fp16x3 / bf16x3 split-operand MFMA, fp32 accumulate)', data='synthetic',
                           config=dict(workload='Charades-STA SeqPAN train step (fwd+bwd+clip+AdamWD, dropout %.1f), batch %d/GPU, '
                                                'T=%d, vdim=%d, L=%d, C=%d, dim=128, 8 heads, 2 attention layers, random init; ONE batch resident in '
                                                'HBM, batch assembly (hual_assemble_batch) outside the timed region'
                                                % (args.drop, args.batch, args.T, args.vdim, args.L, args.C),
                                       global_batch=args.batch * world, T=args.T, vdim=args.vdim, L=args.L,
                                       parallelism='dp%d' % world, launch='eager' if args.no_graph else 'hipGraph',
                                       video_dtype=args.video_dtype,
                                       final_loss=round(loss, 4)),
                           roofline=legs.get('roofline'), cpu_baseline=legs.get('cpu_baseline'))
                for k in ('other_feature_dtype', 'forward_only', 'host_fed', 'epoch_loop', 'epoch_loop_anet', 'rccl'):
                    if legs.get(k) is not None:
                        out[k] = legs[k]
                out.update(extra or {})
                print(json.dumps(out), flush=True)

    deadline_s = float(os.environ.get('HUAL_BENCH_LEG_DEADLINE_S', '900'))

    def cut_short():
        emit(dict(legs_cut_short='the legs after the timed region did not finish within %.0f s; the line holds those that did' % deadline_s))
        sys.stderr.write('[bench] rank %d: leg deadline reached, leaving\n' % rank)
        sys.stderr.flush()
        os._exit(0)
    watchdog = threading.Timer(deadline_s, cut_short)
    watchdog.daemon = True
    watchdog.start()

    rccl = None
    if dp:
        # the step's one collective of size: all-reduce(sum) of the flat fp32 gradient bucket, timed on its own
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            torch.distributed.all_reduce(model.grads)
        barrier()
        ev0.record()
        for _ in range(50):
            torch.distributed.all_reduce(model.grads)
        ev1.record()
        torch.cuda.synchronize()
        rccl = dict(rccl_ranks=world, allreduce_us=round(ev0.elapsed_time(ev1) * 1e3 / 50, 1),
                    allreduce_bytes=int(model.grads.numel() * 4), backend=torch.distributed.get_backend(),
                    step_launch=trainer.dp_launch)
        # the same sum through the one-shot peer-mapped all-reduce (hual_amd/xgmi.py; off by default in the step:
        # HUAL_ALLREDUCE=custom): ONE guarded call first - compared with RCCL's result, status word checked - and only then the timing
        if world > 1:
            # every decision below is COLLECTIVE (an all-reduced flag): a rank that raised alone would leave its peers in a collective
            def agree(ok):
                f = torch.tensor([1.0 if ok else 0.0], device=dev)
                torch.distributed.all_reduce(f, op=torch.distributed.ReduceOp.MIN)
                return float(f.item()) >= 1.0
            ar, msg, err = None, None, None
            try:
                from hual_amd.xgmi import OneShotAllReduce
                buf = torch.randn(model.grads.numel(), device=dev, generator=torch.Generator(device=dev).manual_seed(7 + rank))
                ref = buf.clone()
                torch.distributed.all_reduce(ref)
                ar = OneShotAllReduce(buf)                     # (its own failures are collective: hual_amd/xgmi.py)
            except Exception as e:
                msg = 'setup: ' + str(e)[:250]
            if agree(ar is not None):
                try:
                    ar()
                    torch.cuda.synchronize()
                    ar.check()
                    err = float((buf - ref).abs().max() / ref.abs().max())
                    if not err < 1e-5:
                        msg = 'result differs from RCCL (max rel %.3g on this rank)' % err
                except Exception as e:
                    msg = 'first call: ' + str(e)[:250]
                if agree(msg is None):
                    for _ in range(5):
                        ar()
                    barrier()
                    ev0.record()
                    for _ in range(50):
                        ar()
                    ev1.record()
                    torch.cuda.synchronize()
                    st = int(ar.status.item())
                    rccl['custom_allreduce'] = dict(us=round(ev0.elapsed_time(ev1) * 1e3 / 50, 1), max_rel_diff_vs_rccl=err, status_word=st,
                                                    kind='one-shot reduce-scatter + all-gather over hipIpc peer mappings, 2 flag barriers',
                                                    used_by_the_timed_steps=os.environ.get('HUAL_ALLREDUCE') == 'custom')
                else:
                    rccl['custom_allreduce'] = dict(error=msg or 'failed on another rank')
                try:
                    ar.close()
                except Exception:
                    pass
            else:
                rccl['custom_allreduce'] = dict(error=msg or 'setup failed on another rank')
        model.grads.zero_()
    legs['rccl'] = rccl
    roof = None
    print('[bench] timed region done: %.3f ms/step' % (dt / args.steps * 1e3), file=sys.stderr, flush=True)
    if rank == 0 and not args.no_roofline:
        # roofline leg: same step launched eagerly; every launch carries start/stop events of its own dispatch
        l = lib.load()
        eager = Trainer(model, world=1, use_graph=False)
        eager.set_batch(b['video'], b['lens'], b['word_ids'], b['char_ids'], b['y1'], b['y2'], b['match'], b['inner'], video_dtype=vdt)
        eager.step(lr=1e-4, drop_rate=args.drop)
        torch.cuda.synchronize()
        l.hual_prof_begin()
        psteps = 5
        for _ in range(psteps):
            eager.step(lr=1e-4, drop_rate=args.drop)
        nk = l.hual_prof_end()
        fam = []
        for i in range(nk):
            name = ctypes.create_string_buffer(128)
            launches, usec = ctypes.c_int64(0), ctypes.c_double(0)
            flops, byts = ctypes.c_double(0), ctypes.c_double(0)
            l.hual_prof_get(i, name, 128, ctypes.byref(launches), ctypes.byref(usec), ctypes.byref(flops), ctypes.byref(byts))
            if launches.value > 0:
                fam.append(dict(kernel=name.value.decode(), launches=int(launches.value), us=usec.value, flops=flops.value,
                                bytes=byts.value))
        fam.sort(key=lambda d: -d['us'])
        total_us = sum(d['us'] for d in fam)
        try:
            tr_tab = json.load(open(os.path.join(ROOT, 'profiles', TRAFFIC_FILE)))
        except (OSError, ValueError):
            tr_tab = {}

        shape_key = 'B%d T%d L%d C%d vdim%d drop%.1f %s' % (args.batch, args.T, args.L, args.C, args.vdim, args.drop, args.video_dtype)

        def roof_of(top):
            # which roof bounds a kernel: arithmetic intensity of its launches (algorithmic FLOPs / algorithmic bytes, both
            # summed by the launch wrappers) against the ridge of the pipe it runs on.  A [M,128]x[128,128] fp32-in/out
            # layer has 32 FLOP/B: above the fp32-matrix ridge (157.3 TF / 8 TB/s = 20 FLOP/B) but below the split-bf16 ridge
            # (2516.8 / 3 passes / 8 TB/s = 105 FLOP/B), so with the split-bf16 kernels the dense layers are HBM-bound.
            tflops = top['flops'] / top['us'] / 1e6 if top['flops'] > 0 else 0.0      # algorithmic TFLOP/s
            gbs = top['bytes'] / top['us'] / 1e3                                       # algorithmic GB/s
            peak, passes, pipe = mfma_peak(top['kernel'])
            ai = top['flops'] / top['bytes'] if top['bytes'] > 0 else float('inf')
            ridge = (peak / passes) * 1e3 / PEAK_HBM_GBS if passes else float('inf')
            common = dict(kernel=top['kernel'], avg_launch_us=round(top['us'] / top['launches'], 2),
                          launches_per_step=top['launches'] // psteps, share_of_kernel_time=round(top['us'] / total_us, 3),
                          arithmetic_intensity_flop_per_byte=round(ai, 1), ridge_flop_per_byte=round(ridge, 1))
            if top['flops'] > 0 and passes and ai >= ridge:
                roof = dict(bound='mfma', achieved=round(tflops, 2), peak=peak, unit='TFLOP/s', frac=round(tflops / peak, 4),
                            traffic=None, **common)
            else:
                roof = dict(bound='hbm', achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit='GB/s', frac=round(gbs / PEAK_HBM_GBS, 4),
                            peak_achievable_gbs=PEAK_HBM_ACHIEVABLE_GBS, frac_of_achievable=round(gbs / PEAK_HBM_ACHIEVABLE_GBS, 4),
                            traffic=None, **common)
            if top['flops'] > 0 and passes:
                roof.update(algorithmic_tflops=round(tflops, 2), mfma_pipe=PIPE_NAMES[pipe], mfma_pipe_peak_tflops=peak,
                            mfma_passes=passes, mfma_issue_frac=round(passes * tflops / peak, 4),
                            frac_of_fp32_matrix_peak=round(tflops / PEAK_F32_MATRIX_TFLOPS, 4))
            sq = sq_figures(top['kernel'], top['us'] / top['launches'], shape_key)
            if sq:
                roof.update(sq)
            # HBM traffic of that kernel: rocprofv3 PMC passes cannot run inside this process, so the per-launch figure is
            # read from the committed summary of the separate FETCH_SIZE / WRITE_SIZE passes (scripts/pmc_traffic.py,
            # gfx950 correction applied there); null when the file does not cover the kernel.
            if top['kernel'] in tr_tab:
                roof['traffic'] = tr_tab[top['kernel']]['hbm_bytes_per_launch']
                roof['traffic_unit'] = 'bytes/launch'
                roof['traffic_source'] = 'profiles/' + TRAFFIC_FILE
            roof['algorithmic_bytes_per_launch'] = round(top['bytes'] / top['launches'])
            return roof

        # the dominant kernel = the kernel function with the largest time per step (all its launches); when another kernel
        # has the longest single launch, its roofline rides along as `largest_launch`
        roof = roof_of(fam[0])
        big = max(fam, key=lambda d: d['us'] / d['launches'])
        if big is not fam[0]:
            roof['largest_launch'] = roof_of(big)
        # the feature-load phase (video_conv1d + query_conv1d: streams the [B,T,vdim] clip features once, writes the four
        # K-quarter partial slabs) against HBM
        for d in fam:
            if d['kernel'] == FEATURE_LOAD_KERNEL and d['bytes'] > 0:
                gbs = d['bytes'] / d['us'] / 1e3
                roof['feature_load'] = dict(bound='hbm', kernel=d['kernel'], achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit='GB/s',
                                            frac=round(gbs / PEAK_HBM_GBS, 4), avg_launch_us=round(d['us'] / d['launches'], 2),
                                            algorithmic_bytes_per_launch=round(d['bytes'] / d['launches']),
                                            tflops=round(d['flops'] / d['us'] / 1e6, 2))
                if d['kernel'] in tr_tab:
                    roof['feature_load']['traffic'] = tr_tab[d['kernel']]['hbm_bytes_per_launch']
        # the attention contractions (QK^T, PV and their gradients; north_star: "MFMA utilisation on the attention GEMMs"): algorithmic
        # TFLOP/s of those launches against the fp32-matrix peak (the figure of the round-1 kernels, which ran on that pipe)
        # and against the bf16 pipe the kernels now use with three passes per product (csrc/attn.hip)
        def attn_row(d):
            peak, passes, pipe = mfma_peak(d['kernel'])
            tf = d['flops'] / d['us'] / 1e6
            row = dict(kernel=d['kernel'], achieved=round(tf, 2), peak=peak, unit='TFLOP/s (algorithmic, one pass counted)',
                       frac=round(tf / peak, 4), mfma_pipe=PIPE_NAMES[pipe], mfma_passes=passes,
                       mfma_issue_frac=round(passes * tf / peak, 4), frac_of_fp32_matrix_peak=round(tf / PEAK_F32_MATRIX_TFLOPS, 4),
                       us_per_step=round(d['us'] / psteps, 1))
            sq = sq_figures(d['kernel'], d['us'] / d['launches'], shape_key)
            if sq:
                row.update(sq)
            return row
        roof['attention'] = [attn_row(d) for d in fam if d['kernel'].startswith('attn_') and d['flops'] > 0]
        # north_star asks for >= 60 % of the attention-GEMM MFMA peak: at the reference's head size 16 the kernels issue MFMAs at
        # the `mfma_issue_frac` above of the pipe they run on (VALU bound: exp, dropout, operand splits) - not met, stated
        roof['north_star_attention_60pct_met'] = False
        # whole step: algorithmic FLOPs (BASELINE.md section 2 for the c2 shape, else the launch wrappers' own sums) over the timed
        # ms/step; HBM bytes per step measured by the PMC passes (sum over kernels of bytes/launch x launches/step) against
        # the compulsory bytes (clip features + query ids + labels read once, outputs written, parameters read once,
        # gradients written + read, AdamWD reading p, m, v and writing p, m, v)
        n_par = model.params.numel() * 4
        compulsory = (args.batch * args.T * args.vdim * 4 + args.batch * args.L * (1 + args.C) * 4 + args.batch * args.T * 4 * 4
                      + args.batch * args.T * 6 * 4 + 9 * n_par)
        is_c2 = (args.T, args.vdim, args.L, args.C) == (128, 1024, 20, 8)
        gflop = MFLOP_PER_CLIP_C2 * args.batch / 1e3 if is_c2 else sum(d['flops'] for d in fam) / psteps / 1e9
        ms = dt / args.steps * 1e3
        alg_bytes = sum(d['bytes'] for d in fam) / psteps      # what the launch wrappers count: operands in, results out, per launch
        step = dict(algorithmic_gflop_per_step=round(gflop, 2), tflops=round(gflop / ms, 2),
                    algorithmic_bytes_per_step=int(alg_bytes), algorithmic_tb_s=round(alg_bytes / (ms * 1e-3) / 1e12, 3),
                    frac_of_fp32_matrix_peak=round(gflop / ms / PEAK_F32_MATRIX_TFLOPS, 4),
                    frac_of_bf16_matrix_peak=round(gflop / ms / PEAK_BF16_MATRIX_TFLOPS, 4),
                    compulsory_hbm_bytes_per_step=int(compulsory), launches_per_step=sum(d['launches'] for d in fam) // psteps,
                    kernel_time_us_per_step=round(total_us / psteps, 1))
        if tr_tab:
            meas, covered = 0.0, 0
            for d in fam:
                if d['kernel'] in tr_tab:
                    meas += tr_tab[d['kernel']]['hbm_bytes_per_launch'] * (d['launches'] // psteps)
                    covered += 1
            step.update(measured_hbm_bytes_per_step=int(meas), measured_tb_s=round(meas / (ms * 1e-3) / 1e12, 3),
                        measured_frac_of_hbm_peak=round(meas / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                        measured_over_compulsory=round(meas / compulsory, 1),
                        traffic_source='profiles/' + TRAFFIC_FILE, kernels_covered='%d of %d' % (covered, len(fam)))
        roof['step'] = step
        roof['families'] = [dict(kernel=d['kernel'], launches_per_step=d['launches'] // psteps,
                                 us_per_step=round(d['us'] / psteps, 1),
                                 tflops=round(d['flops'] / d['us'] / 1e6, 2) if d['flops'] > 0 else None,
                                 gbs=round(d['bytes'] / d['us'] / 1e3, 1) if d['bytes'] > 0 else None) for d in fam]
    legs['roofline'] = roof
    print('[bench] roofline leg done', file=sys.stderr, flush=True)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()
        try:
            cpu['gpu_at_cpu_shape'] = gpu_at_cpu_shape(dev, args.drop)
            cpu['gpu_over_cpu_same_shape'] = round(cpu['gpu_at_cpu_shape']['value'] / cpu['value'], 1)
        except Exception as e:      # never cost the bench line
            cpu['gpu_at_cpu_shape'] = dict(error=str(e)[:200])

    legs['cpu_baseline'] = cpu
    other_feed = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the same step with the clip features held in HBM in the other element type (BASELINE configs[1] says bf16 features;
        # the headline number is measured on the reference's float32 placeholder)
        try:
            odt = torch.float32 if vdt == torch.bfloat16 else torch.bfloat16
            t2 = Trainer(model, world=1, use_graph=not args.no_graph)
            t2.set_batch(b['video'], b['lens'], b['word_ids'], b['char_ids'], b['y1'], b['y2'], b['match'], b['inner'], video_dtype=odt)
            for _ in range(100):
                t2.step(lr=1e-4, drop_rate=args.drop)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(500):
                t2.step(lr=1e-4, drop_rate=args.drop)
            torch.cuda.synchronize()
            d2 = time.perf_counter() - t0
            other_feed = dict(video_dtype='f32' if odt == torch.float32 else 'bf16', ms_per_step=round(d2 / 500 * 1e3, 4),
                              value=round(args.batch * 500 / d2, 1), unit='clips/s', steps=500)
        except Exception as e:      # never cost the bench line
            other_feed = dict(error=str(e)[:200])

    legs['other_feature_dtype'] = other_feed
    fwd_only = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the evaluation fetch set on its own (runner_utils.py:166: start / end index; all five fetches come out of the one pass), eager launches
        try:
            dv = [torch.as_tensor(b[k]).to(dev) for k in ('video', 'lens', 'word_ids', 'char_ids')]
            for _ in range(20):
                model.forward(*dv, drop_rate=0.0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(300):
                model.forward(*dv, drop_rate=0.0)
            torch.cuda.synchronize()
            d3 = time.perf_counter() - t0
            fwd_only = dict(value=round(args.batch * 300 / d3, 1), unit='clips/s', ms_per_pass=round(d3 / 300 * 1e3, 4), passes=300,
                            launch='eager', note='hual_seqpan_forward without labels: logits, match scores and spans of one batch per pass')
            if cpu is not None and isinstance(cpu.get('forward_only'), dict) and 'value' in cpu['forward_only']:
                fwd_only['cpu_forward_only_c1_clips_s'] = cpu['forward_only']['value']
        except Exception as e:      # never cost the bench line
            fwd_only = dict(error=str(e)[:200])

    legs['forward_only'] = fwd_only
    host_fed = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # The PCIe-inclusive rate (never `value`): the same step fed from HOST batches the way the reference feeds it
        # (runner_utils.py:141-147: a numpy batch per step through feed_dict), as hual_amd/feeder.py pipelines it - batch k + 1 uploads
        # from pinned memory on a copy stream while step k runs.  `pinned`: the producer wrote the batch into the pinned slot itself
        # (no host copy in the loop: what the device side can sustain); `pageable`: numpy arrays in pageable memory, staged by one
        # memcpy per step on the calling thread (what a drop-in feed_dict caller gets).
        try:
            from hual_amd.feeder import HostFeeder
            host_fed = dict(note='train step fed from host batches through pinned staging + one async upload per step (hual_amd/feeder.py); '
                                 'PCIe-inclusive, not the bench value')
            hb = dict(video=np.ascontiguousarray(b['video'], dtype=np.float32), video_seq_len=b['lens'], word_ids=b['word_ids'],
                      char_ids=b['char_ids'], y1=b['y1'], y2=b['y2'], match_labels=b['match'], inner_labels=b['inner'])
            for fdt, name in ((torch.float32, 'f32'), (torch.bfloat16, 'bf16')):
                ht = Trainer(model, world=1, use_graph=not args.no_graph)
                fd = HostFeeder(ht, capacity=(args.batch, args.T, args.L, args.C), vdim=args.vdim, video_dtype=fdt)
                for _ in range(6):
                    fd.feed(hb, 1e-4, args.drop)                     # fills both pinned slots, captures both slots' graphs
                fd.collect()
                nst = 300
                t0 = time.perf_counter()
                for _ in range(nst):
                    fd.stage_views(args.batch, args.T, args.L, args.C)      # (the slot already holds the batch)
                    fd.submit(1e-4, args.drop)
                fd.collect()
                d4 = time.perf_counter() - t0
                nbytes = fd.stats['bytes_uploaded'] // fd.stats['batches']
                # the upload on its own
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
                for _ in range(20):
                    fd.dev[0][:nbytes].copy_(fd.host[0][:nbytes], non_blocking=True)
                ev1.record()
                torch.cuda.synchronize()
                up_ms = ev0.elapsed_time(ev1) / 20
                row = dict(ms_per_step=round(d4 / nst * 1e3, 4), value=round(args.batch * nst / d4, 1), unit='clips/s', steps=nst,
                           upload_bytes_per_step=int(nbytes), upload_alone_ms=round(up_ms, 4),
                           upload_gb_s=round(nbytes / up_ms / 1e6, 1), over_resident=round(d4 / nst / (dt / args.steps), 3))
                if fdt == torch.float32:
                    npg = 60
                    t0 = time.perf_counter()
                    for _ in range(npg):
                        fd.feed(hb, 1e-4, args.drop)
                    fd.collect()
                    d5 = time.perf_counter() - t0
                    row['pageable'] = dict(ms_per_step=round(d5 / npg * 1e3, 4), value=round(args.batch * npg / d5, 1), steps=npg,
                                           note='numpy batch in pageable memory, staged into the pinned slot by 4 copy threads per step')
                host_fed['pinned_' + name] = row
                del fd, ht
        except Exception as e:      # never cost the bench line
            host_fed = dict(error=str(e)[:300])
        model.grads.zero_()

    legs['host_fed'] = host_fed
    epoch_loop, epoch_anet = None, None
    if not args.no_epoch_loop and (world == 1 or dp):
        # every rank runs these legs when the job is data parallel (collectives inside); rank 0 reports
        def leg(name, **kw):
            try:
                r = epoch_loop_leg(dev, args, world=world, **kw)
                if rank == 0:
                    print('[bench] %s leg done: %.3f ms/step' % (name, r['ms_per_step']), file=sys.stderr, flush=True)
                return r
            except Exception as e:      # never cost the bench line (every rank fails or passes together: the legs are collective)
                return dict(error=str(e)[:300])
        epoch_loop = leg('epoch-loop', resident_ms=dt / args.steps * 1e3)
        legs['epoch_loop'] = epoch_loop
        # the reference's own shape distributions (BASELINE configs[3], [4]: ActivityNet annotations): the YAML's batch 16 at
        # max_vlen 100, and configs[3]'s 32 clips per GPU at T <= 256
        epoch_anet = [leg('epoch-loop anet b16', resident_ms=None, kind='anet', bs=16, max_vlen=100, N=args.anet_samples, epochs=3),
                      leg('epoch-loop anet b32 T256', resident_ms=None, kind='anet', bs=32, max_vlen=256, N=args.anet_samples // 2, epochs=3)]

    legs['epoch_loop_anet'] = epoch_anet
    emit()
    watchdog.cancel()
    if dp:
        torch.distributed.barrier()      # rank 0 ran the roofline leg alone: leave together
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()

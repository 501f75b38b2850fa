"""GPU: the main.py-shaped driver (train / test / infer_trainset, checkpoints by TF variable name) on a small synthetic,
learnable task: the query's first word says where the moment is, the clip features carry a position signal."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _videos(nvid, vdim, seed):
    g = np.random.default_rng(seed)
    vis = {}
    for v in range(nvid):
        T = int(g.integers(20, 33))
        f = 0.1 * g.standard_normal((T, vdim)).astype(np.float32)
        f[:, 0] = np.linspace(-1, 1, T)                    # position signal
        vis['v%d' % v] = f
    return vis


def _task(n, vis, seed):
    g = np.random.default_rng(seed)
    nvid = len(vis)
    dur = {k: float(v.shape[0]) for k, v in vis.items()}
    recs = []
    for i in range(n):
        vid = 'v%d' % int(g.integers(0, nvid))
        T = vis[vid].shape[0]
        part = int(g.integers(0, 3))                       # early / middle / late third, named by the first word
        s = part * T // 3 + 1
        e = min(T - 1, s + T // 3 - 2)
        words = ['w%d' % (2 + part), 'w%d' % int(g.integers(5, 30)), 'w%d' % int(g.integers(5, 30))]
        recs.append(dict(vid=vid, duration=dur[vid], v_len=T, words=words, w_ids=[int(w[1:]) for w in words],
                         c_ids=[[1 + part, 2, 3, 4]] * 3, s_ind=s, e_ind=e))
    return recs


def _test_mious(lines):
    """mIoU of every per-epoch 'TEST:' line the runner logged"""
    return [float(l.split('\t')[4]) for s in lines for l in str(s).splitlines() if l.startswith('TEST:')]


def test_train_test_infer_and_checkpoint(tmp_path):
    from hual_amd.runner import Runner
    vdim = 64
    vis = _videos(24, vdim, 0)
    train = _task(192, vis, 1)
    test = _task(64, vis, 2)
    cfg = dict(task='synth', train=dict(batch_size=32, droprate=0.1, lr=2e-3, epochs=8, clip_norm=1.0),
               model=dict(vdim=vdim, dim=128, num_heads=8, word_dim=300, char_dim=50, max_vlen=32, attn_layer=2),
               loss=dict(match_lambda=1.0, tau=0.3, no_gumbel=True), num_chars=10)
    wv = np.random.default_rng(0).normal(0, 0.4, size=(40, 300)).astype(np.float32)
    lines = []

    class L:
        def info(self, s):
            lines.append(str(s))
    r = Runner(cfg, wv, train, test, vis, ckpt_dir=str(tmp_path / 'ckpt'), logger=L())
    before = r.test_epoch()
    best = r.train()
    after = r.test_epoch()
    assert any(l.startswith('TRAIN:\t') for l in lines) and any(l.startswith('TEST:\t') for l in lines)
    assert os.path.exists(tmp_path / 'ckpt' / 'best_SeqPAN.npz')
    # mIoU improves on the learnable task.  The run is not bit-reproducible (float atomics) and short, and the 64-sample test metric jumps
    # from epoch to epoch: the BEST epoch gains +5.9 .. +33 points (median +16) over 600 runs of scripts/exp/runner_repeat.py, the LAST
    # epoch alone lands below +3 in ~1 % of them - so the bar is on the best epoch.  (Until the weight-gradient launch took a running
    # scale for its A operand, one run in ~20 ended in NaN parameters - activation products beyond the fixed fp16 operand range.)
    assert max(_test_mious(lines)) > before[3] + 3.0, (before, _test_mious(lines))
    assert np.isfinite(after[3])
    assert r.clips_per_s > 0
    # checkpoint round trip: perturb, reload, same predictions as the best epoch's weights give
    t_best = r.test()
    p = r.model.params.clone()
    r.model.params.add_(0.05 * torch.randn_like(p))
    r.load(str(tmp_path / 'ckpt' / 'best_SeqPAN.npz'))
    assert r.test_epoch() == t_best
    recs, m = r.infer_trainset(path=str(tmp_path / 'results' / 're0.pkl'), mc_dropout=0.5)
    assert len(recs) == len(train) and os.path.exists(tmp_path / 'results' / 're0.pkl')
    assert recs[0]['prop_logits'][0].shape == recs[0]['prop_logits1'][0].shape


def test_host_fed_runner_follows_the_device_fed_one(tmp_path):
    """Runner(feed='host'): the reference's own data path (process_batch on the host, one upload per step: hual_amd/feeder.py) instead of
    the HBM-resident set.  Same seeds, same epoch orders, no dropout: the first epoch's metrics are those of the device-fed runner (the
    batches are the same arrays - test_gpu_al.py pins the device assembly to the loader bit for bit), and the task is learnt."""
    from hual_amd.runner import Runner
    vdim = 64
    vis = _videos(24, vdim, 0)
    train = _task(192, vis, 1)
    test = _task(64, vis, 2)
    cfg = dict(task='synth', train=dict(batch_size=32, droprate=0.0, lr=2e-3, epochs=8, clip_norm=1.0),
               model=dict(vdim=vdim, dim=128, num_heads=8, word_dim=300, char_dim=50, max_vlen=32, attn_layer=2),
               loss=dict(match_lambda=1.0, tau=0.3, no_gumbel=True), num_chars=10)
    wv = np.random.default_rng(0).normal(0, 0.4, size=(40, 300)).astype(np.float32)

    lines = []

    class L:
        def info(self, s):
            lines.append(str(s))
    rd = Runner(cfg, wv, train, test, vis, ckpt_dir=str(tmp_path / 'd'), logger=L())
    rh = Runner(cfg, wv, train, test, vis, ckpt_dir=str(tmp_path / 'h'), logger=L(), feed='host')
    assert rh.train_set is None
    before = rh.test_epoch()                                       # untrained
    md, mh = rd.train_epoch(1e-4), rh.train_epoch(1e-4)            # six steps each, at a step size where 1e-7 of gradient noise stays noise
    assert np.allclose(md, mh, atol=2.0), (md, mh)                 # (percent)
    d = (rd.model.params - rh.model.params).abs()
    # (two device-fed runs of this task agree on 93 % of the parameters by the same yardstick after these six steps, and on 4 % after twelve:
    # scripts/exp/runner_feed_cmp.py - the bit-level equivalence of the two feeds is test_gpu_feeder.py's, at lr 0)
    assert float((d <= 0.1 * 1e-4).float().mean()) >= 0.85, float((d <= 0.1 * 1e-4).float().mean())
    rh.train(epochs=8)
    after = rh.test_epoch()
    assert max(_test_mious(lines)) > before[3] + 3.0, (before, _test_mious(lines))      # (best epoch: +6.4 .. +28 over 300 host-fed runs)
    assert np.isfinite(after[3])
    assert rh.clips_per_s > 0 and rh._feeder.stats['batches'] == 9 * 6
    recs, m = rh.infer_trainset(mc_dropout=0.5)                    # builds the device-resident set on first use
    assert len(recs) == len(train)

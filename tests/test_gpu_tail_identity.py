"""The layer norms + projections that ride as the TAIL of the launch in front (conv_block_fwd_lnproj_kernel, da_post_lnproj_kernel:
csrc/lnproj_body.h on the rows the producer left in LDS) against the same code launched on its own (HUAL_CB_NO_TAIL=1): the SAME BITS in
every output, loss term and tap of the forward pass - the tail is a forward-only change, and the backward's weight gradients go through
float atomics whose order differs from run to run whatever the path, so gradient tensors are compared to rounding only.  The switch is
read once per process, so each side runs in a process of its own."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize('shape', [(8, 64, 12, 6), (5, 130, 20, 8)])
def test_tail_and_separate_launches_give_the_same_bits(tmp_path, shape):
    outs = []
    for no_tail in ('0', '1'):
        env = dict(os.environ)
        env['HUAL_CB_NO_TAIL'] = no_tail
        out = str(tmp_path / ('tail%s.npz' % no_tail))
        subprocess.run([sys.executable, os.path.join(HERE, 'tail_identity_worker.py'), out] + [str(x) for x in shape], env=env, check=True,
                       timeout=600)
        outs.append(np.load(out))
    a, b = outs
    assert sorted(a.files) == sorted(b.files) and len(a.files) > 150
    fwd = [k for k in a.files if not k.startswith('grad.')]
    assert len(fwd) >= 20
    diff = [k for k in fwd if not np.array_equal(a[k], b[k], equal_nan=True)]
    assert not diff, 'forward tensors that differ between the tail and the separate launches: %s' % diff[:10]
    gmax = max(float(np.abs(a[k]).max()) for k in a.files if k.startswith('grad.'))
    for k in a.files:
        if k.startswith('grad.'):      # (atomic order: rounding of sums that may cancel, so relative to the run's largest gradient as well)
            scale = max(float(np.abs(a[k]).max()), 1e-3 * gmax)
            assert float(np.abs(a[k] - b[k]).max()) <= 1e-5 * scale, k

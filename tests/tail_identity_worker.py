"""Worker of tests/test_gpu_tail_identity.py: one forward + backward of the HIP model at a small shape with fixed seeds; every output,
loss term, a set of taps and all gradient tensors go to an .npz (argv[1]).  Run once with and once without HUAL_CB_NO_TAIL=1."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import parity_util as pu


def main():
    out = sys.argv[1]
    B, T, L, C = (int(x) for x in sys.argv[2:6])
    cfg, p, wv, b, labels = pu.make_case(B=B, T=T, L=L, C=C, seed=99, max_vlen=max(T, L, 8), vdim=512)
    m = pu.hip_model(cfg, p, wv, 'cuda:0')
    m.set_rng(5, 7)
    h = m.forward(b['video'].numpy(), b['lens'].numpy(), b['word_ids'].numpy(), b['char_ids'].numpy(), drop_rate=0.2,
                  labels=tuple(x.numpy() for x in labels))
    torch.cuda.synchronize()
    m.backward()
    torch.cuda.synchronize()
    d = {}
    for k in ('start_logits', 'end_logits', 'match_scores', 'loss', 'loc_loss', 'match_loss', 'align_loss', 'start_index', 'end_index'):
        d['out.' + k] = h[k].detach().cpu().numpy()
    for name in ('da0.ln1', 'da0.lnt', 'da0.qkv', 'da0.ktvt', 'da1.ln1', 'da1.qkv', 'da1.ktvt', 'fe0.a', 'fe0.qkv', 'fe1.qkv', 'fe1.out'):
        d['tap.' + name] = m.tap(name).detach().cpu().numpy()
    for k, g in m.grads_dict().items():
        d['grad.' + k] = np.asarray(g)
    np.savez(out, **d)


if __name__ == '__main__':
    main()

"""Generates the committed golden fixtures with the CPU oracle (float64 master).

The reference cannot be imported here (it needs TensorFlow 1.x, which is not installed and not
installable offline) and ships no fixtures of its own, so these vectors pin the oracle against
itself over time (regression), not against the reference: PARITY UNPINNED.

  python tests/golden/make_golden.py        # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ctr_oracle as O  # noqa: E402
from tests.helpers import no_dropout, tiny_batch, tiny_cfg  # noqa: E402

CASES = [('DeepFM', {}), ('DCN', {}), ('NFM', {}), ('ESMM', {}), ('PNN', {}), ('PNN', {'inner_product': False}),
         ('AFM', {}), ('AFM', {'attention': 'pairs', 'canonical_softmax': True})]


def make(model, mode, extra):
    cfg = no_dropout(tiny_cfg(model, mode, **extra))
    P = O.init_params(model, cfg, seed=11, std=0.12 if model == 'AFM' else 0.3)
    Xc, Xf, y = tiny_batch(cfg, B=8, seed=1)
    if model == 'AFM':                      # keep 0 < y_out < 1 (AFM feeds log_loss an unbounded value)
        Xf = np.abs(Xf) * 0.3
        P = {k: (v if extra and 'attention' in k else np.abs(v)) for k, v in P.items()}
    m = O.OracleModel(model, cfg, P, dtype=torch.float64)
    out = dict(X_cate=Xc, X_cont=Xf, y=y)
    for k, v in P.items():
        out['param/' + k] = np.asarray(v)
    fw = m.forward(Xc, Xf)
    out['y_out'] = fw.y_out.numpy()
    out['Z'] = fw.Z.numpy()
    loss, g, _ = m.loss_and_grads(Xc, Xf, y)
    out['loss'] = np.float64(loss.item())
    assert np.isfinite(out['loss']), (model, mode)
    for k, v in g.items():
        out['grad/' + k] = v.numpy()
    for step in (1, 2, 3):
        m.run(Xc, Xf, y, True)
        if step in (1, 3):
            for k, v in m.P.items():
                out['after%d/%s' % (step, k)] = v.numpy()
    return out


def name(model, mode, extra):
    tag = '_outer' if extra.get('inner_product') is False else '_pairs' if extra.get('attention') else ''
    return '%s%s_%s.npz' % (model, tag, mode)


if __name__ == '__main__':
    only = sys.argv[1:]                     # optional file-name filter: rewrite just those fixtures
    for model, extra in CASES:
        for mode in ('R', 'S'):
            if only and name(model, mode, extra) not in only:
                continue
            np.savez_compressed(os.path.join(HERE, name(model, mode, extra)), **make(model, mode, extra))
            print('wrote', name(model, mode, extra))

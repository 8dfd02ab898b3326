"""HIP path vs the committed golden vectors (tests/golden/*.npz): forward, loss, gradients and
the variables after one and three optimizer steps.  fp32 vs float64 goldens: 1e-5 relative on
forward/gradients (the stated bar), 1e-4 on weights after three Adam/Adagrad steps."""
import glob
import importlib
import os

import numpy as np
import pytest

from tests.helpers import assert_close, no_dropout, rel_err, tiny_cfg

pytestmark = pytest.mark.gpu
FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'golden', '*.npz')))


@pytest.mark.parametrize('path', FILES, ids=[os.path.basename(p)[:-4] for p in FILES])
def test_hip_path_reproduces_golden(sess, path):
    import recommendation_amd as ra
    ra.enable_zoo_imports()
    parts = os.path.basename(path)[:-4].split('_')
    name, mode = parts[0], parts[-1]
    g = np.load(path)
    cfg = no_dropout(tiny_cfg(name, mode))
    if 'outer' in parts:
        cfg.inner_product = False
    if 'pairs' in parts:
        cfg.attention = 'pairs'
    cfg.adam_mode = 'tf1'
    P = {k[6:]: g[k] for k in g.files if k.startswith('param/')}
    m = getattr(importlib.import_module('%s.%s' % (name, name)), name)(cfg, cfg.cate_num, cfg.cont_num, cfg.cate_list)
    m.build()
    m.initialize(sess, params=P)
    Xc, Xf, y = g['X_cate'], g['X_cont'], g['y']
    assert rel_err(m.predict_proba(sess, Xc, Xf), g['y_out']) < 1e-5
    loss = m.run(sess, Xc, Xf, y, True)
    assert abs(float(loss) - float(g['loss'])) < 1e-5 * max(1.0, abs(float(g['loss'])))
    gmax = max(float(np.max(np.abs(g['grad/' + k]))) for k in m.get_grads())
    for k, v in m.get_grads().items():
        # softmax gradients sum to zero: attention_b's gradient cancels exactly in R-mode's repeated rows
        if 'pairs' in parts and 'attention' in k:
            # gradients through the softmax over the pairs are sums of terms that cancel (sum_k da_k = 0; exactly,
            # for attention_b in R-mode's repeated rows): their bar is 1e-5 of the model's largest gradient
            assert np.max(np.abs(v - g['grad/' + k])) < 1e-5 * gmax, k
        else:
            assert_close(v, g['grad/' + k], 1e-5, k)
    for k, v in m.get_params().items():
        assert rel_err(v, g['after1/' + k]) < 1e-5, k
    m.run(sess, Xc, Xf, y, True)
    m.run(sess, Xc, Xf, y, True)
    for k, v in m.get_params().items():
        assert rel_err(v, g['after3/' + k]) < 1e-4, k        # free-running steps: see helpers.assert_weights_track

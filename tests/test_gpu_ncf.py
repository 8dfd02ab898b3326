"""NCF (GMF + MLP over looked-up user / item rows, dense Adam) on the HIP path vs the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import ctr_oracle as O
from tests.helpers import assert_close, assert_weights_track, rel_err

pytestmark = pytest.mark.gpu


def _case(dropout=0.0):
    cfg = O.default_cfg('NCF', embed_size=8, dropout=dropout)
    cfg.user_size, cfg.item_size = 11, 7
    rng = np.random.default_rng(4)
    P = O.ncf_init_params(cfg, seed=3, std=0.4)
    P = {k: (v if k.endswith('kernel') else rng.standard_normal(v.shape) * 0.2) for k, v in P.items()}
    B = 24
    X = np.stack([rng.integers(0, 11, B), rng.integers(0, 7, B)], 1)
    X[0, 0] = 50                                    # out-of-range user id: all-zero one-hot row
    y = rng.integers(0, 2, B)
    return cfg, P, X, y


def _build(sess, cfg, P):
    import recommendation_amd as ra
    ra.enable_zoo_imports()
    from NCF import NCF
    m = NCF.NCF(cfg, cfg.user_size, cfg.item_size)
    m.build()
    m.initialize(sess, params=P)
    return m


def test_ncf_forward_grads_and_adam_steps(sess):
    cfg, P, X, y = _case()
    orc = O.OracleNCF(cfg, P)
    m = _build(sess, cfg, P)
    assert rel_err(m.predict_proba(sess, X), orc.forward(X).y_out.numpy()) < 1e-5
    loss_ref, g_ref, _ = orc.loss_and_grads(X, y)
    loss = m.run(sess, X, y, True)
    assert abs(float(loss) - float(loss_ref)) < 1e-5
    for k, v in m.get_grads().items():
        assert_close(v, g_ref[k].numpy(), 1e-5, k)
    orc.run(X, y, True)
    for _ in range(2):
        assert abs(float(orc.run(X, y, True)) - float(m.run(sess, X, y, True))) < 1e-5
    assert_weights_track(m.get_params(), orc.P)
    a, p = m.run(sess, X, y, False)
    a_o, p_o = orc.run(X, y, False)
    assert np.array_equal(a, a_o) and np.array_equal(p, p_o)


def test_ncf_dropout_masks_and_checkpoint(sess, tmp_path):
    cfg, P, X, y = _case(dropout=0.25)
    m = _build(sess, cfg, P)
    B, E = len(y), cfg.embed_size
    seed = sess.seed * 1000003 + 0
    masks = {'mlp_%d' % i: torch.as_tensor(O.keep_mask(seed, 10 + i, B * w, 0.75).reshape(B, w))
             for i, w in ((1, 2 * E), (2, E), (3, E // 2))}
    loss_ref, g_ref, _ = O.OracleNCF(cfg, P).loss_and_grads(X, y, masks)
    assert abs(float(m.run(sess, X, y, True)) - float(loss_ref)) < 1e-5
    for k, v in m.get_grads().items():
        assert_close(v, g_ref[k].numpy(), 1e-5, k)
    path = str(tmp_path / 'ncf.ckpt')
    m.saver.save(sess, path)
    m2 = _build(sess, cfg, P)
    m2.saver.restore(sess, path)
    assert m.run(sess, X, y, True) == m2.run(sess, X, y, True)
    for (k, a), b in zip(m.get_params().items(), m2.get_params().values()):
        assert np.array_equal(a, b), k

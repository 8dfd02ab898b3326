"""Wide & Deep (FTRL-with-L1 wide part, Adagrad deep part) on the HIP path vs the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import ctr_oracle as O
from tests.helpers import assert_close, rel_err

pytestmark = pytest.mark.gpu


def _case(dropout=0.0, std=0.3):
    cfg = O.default_cfg('Wide_Deep', embed_size=4, hidden_layers=[8, 4], dropout=dropout)
    cfg.cross_list, cfg.cate_list, cfg.cont_num = [7, 4], [3, 5, 6], 2
    cfg.strict_ids = False                         # the out-of-range id below reads a zero row instead of raising
    P = O.wide_deep_init_params(cfg, seed=2, std=std)
    rng = np.random.default_rng(1)
    B = 16
    Xr = np.stack([rng.integers(0, v, B) for v in cfg.cross_list], 1)
    Xc = np.stack([rng.integers(0, v, B) for v in cfg.cate_list], 1)
    Xc[0, 1] = 9                                   # out of range: zero row, no update
    Xf = rng.standard_normal((B, 2))
    y = rng.integers(0, 2, B)
    return cfg, P, Xr, Xc, Xf, y


def _build(sess, cfg, P):
    import recommendation_amd as ra
    ra.enable_zoo_imports()
    from Wide_Deep import Wide_Deep
    m = Wide_Deep.Wide_Deep(cfg, len(cfg.cross_list), len(cfg.cate_list), cfg.cont_num, cfg.cross_list, cfg.cate_list)
    m.build()
    m.initialize(sess, params=P)
    return m


def test_wide_deep_forward_grads_and_steps(sess):
    cfg, P, Xr, Xc, Xf, y = _case()
    orc = O.OracleWideDeep(cfg, P)
    m = _build(sess, cfg, P)
    assert rel_err(m.predict_proba(sess, Xr, Xc, Xf), orc.forward(Xr, Xc, Xf).y_out.numpy()) < 1e-5
    loss_ref, g_ref, _ = orc.loss_and_grads(Xr, Xc, Xf, y)
    loss = m.run(sess, Xr, Xc, Xf, y, True)
    assert abs(float(loss) - float(loss_ref)) < 1e-5
    for k, v in m.get_grads().items():
        assert_close(v, g_ref[k].numpy(), 1e-5, k)
    orc.run(Xr, Xc, Xf, y, True)
    for _ in range(3):
        l_o, l_g = orc.run(Xr, Xc, Xf, y, True), m.run(sess, Xr, Xc, Xf, y, True)
        assert abs(float(l_o) - float(l_g)) < 1e-5
    assert m.global_step == orc.global_step == 8    # two minimize() calls per step
    for k, v in m.get_params().items():
        ref = orc.P[k].numpy()
        assert np.abs(v - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), k
    # FTRL with l1 = 0.1 zeroes (almost all of) the touched wide rows; untouched rows keep their initial values
    w0 = m.get_params()['wide_net/wide_cross_w_0']
    touched = np.zeros(7, bool)
    touched[Xr[:, 0]] = True
    assert np.mean(w0[touched] == 0.0) > 0.8
    assert np.array_equal(w0[~touched], np.asarray(P['wide_net/wide_cross_w_0'], np.float32)[~touched])
    a, p = m.run(sess, Xr, Xc, Xf, y, False)
    a_o, p_o = orc.run(Xr, Xc, Xf, y, False)
    assert np.array_equal(a, a_o) and np.array_equal(p, p_o)


def test_wide_deep_ftrl_moves_weights_with_large_gradients(sess):
    """With |z| above l1 the FTRL closed form gives non-zero weights: large labels-vs-logit gradients."""
    cfg, P, Xr, Xc, Xf, y = _case(std=0.5)
    cfg.learning_rate = 0.5
    P = {k: v * (20.0 if 'wide_output_w' in k else 1.0) for k, v in P.items()}
    orc = O.OracleWideDeep(cfg, P)
    m = _build(sess, cfg, P)
    for _ in range(4):
        l_o, l_g = orc.run(Xr, Xc, Xf, y, True), m.run(sess, Xr, Xc, Xf, y, True)
        assert abs(float(l_o) - float(l_g)) < 2e-5 * max(1.0, abs(float(l_o)))
    moved = 0
    for k, v in m.get_params().items():
        ref = orc.P[k].numpy()
        assert np.abs(v - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max()), k
        moved += int(k.startswith('wide_net/wide_c') and k[-3] != 'b' and np.abs(ref).max() > 0 and
                     not np.array_equal(ref, np.asarray(P[k])))
    assert moved > 0


def test_wide_deep_dropout_and_checkpoint(sess, tmp_path):
    cfg, P, Xr, Xc, Xf, y = _case(dropout=0.3)
    m = _build(sess, cfg, P)
    B = len(y)
    seed = sess.seed * 1000003 + 0
    masks = {'deep_%d' % i: torch.as_tensor(O.keep_mask(seed, 10 + i, B * L, 0.7).reshape(B, L))
             for i, L in enumerate(cfg.hidden_layers)}
    orc = O.OracleWideDeep(cfg, P)
    loss_ref, _, _ = orc.loss_and_grads(Xr, Xc, Xf, y, masks)
    assert abs(float(m.run(sess, Xr, Xc, Xf, y, True)) - float(loss_ref)) < 1e-5
    path = str(tmp_path / 'wd.ckpt')
    m.saver.save(sess, path)
    m2 = _build(sess, cfg, P)
    m2.saver.restore(sess, path)
    assert m2.global_step == m.global_step
    l1, l2 = m.run(sess, Xr, Xc, Xf, y, True), m2.run(sess, Xr, Xc, Xf, y, True)
    assert l1 == l2
    for (k, a), b in zip(m.get_params().items(), m2.get_params().values()):
        assert np.array_equal(a, b), k


def test_wide_deep_raises_on_an_id_outside_its_table(sess):
    """tf.nn.embedding_lookup (Wide_Deep/Wide_Deep.py:51,57,74) rejects an id outside the table; so does run()."""
    cfg, P, Xr, Xc, Xf, y = _case()
    cfg.strict_ids = True
    m = _build(sess, cfg, P)
    with pytest.raises(IndexError):
        m.run(sess, Xr, Xc, Xf, y, True)
    Xc[0, 1] = 2
    assert np.isfinite(m.run(sess, Xr, Xc, Xf, y, True))      # the counter was reset; a clean batch passes

"""Model classes on the HIP path vs the CPU oracle: forward, loss, gradients and optimizer steps,
in the reference-exact (R) and the standard-lookup (S) semantics.
Tolerance: fp32 forward/gradients within 1e-5 relative (max-norm), the bar BASELINE.json states."""
import numpy as np
import pytest
import torch

from oracle import ctr_oracle as O
from tests.helpers import assert_close, assert_weights_track, no_dropout, rel_err, tiny_batch, tiny_cfg

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _build(sess, model_name, cfg, P):
    import recommendation_amd as ra
    ra.enable_zoo_imports()
    import importlib
    mod = importlib.import_module('%s.%s' % (model_name, model_name))
    m = getattr(mod, model_name)(cfg, cfg.cate_num, cfg.cont_num, cfg.cate_list)
    m.build()
    m.initialize(sess, params=P)
    return m


MODELS = ['DeepFM', 'DCN', 'NFM', 'ESMM', 'PNN', 'PNN_outer', 'PNN_persample', 'AFM']


@pytest.mark.parametrize('mode', ['R', 'S'])
@pytest.mark.parametrize('name', MODELS)
def test_forward_grads_steps(sess, name, mode):
    variant, name = name, name.split('_')[0]
    cfg = no_dropout(tiny_cfg(name, mode))
    if variant == 'PNN_outer':
        cfg.inner_product = False
    if variant == 'PNN_persample':
        cfg.inner_mode = 'per_sample'            # option beyond the reference (per-sample norm)
    cfg.adam_mode = 'tf1'
    # AFM feeds log_loss an unbounded y_out (AFM/AFM.py:85-88); small weights keep 0 < y_out < 1
    P = O.init_params(name, cfg, seed=3, std=0.3)
    Xc, Xf, y = tiny_batch(cfg, B=8)
    if name == 'AFM':
        Xf = np.abs(Xf) * 0.5
        P = {k: np.abs(v) for k, v in P.items()}
    orc = O.OracleModel(name, cfg, P, dtype=torch.float64)
    m = _build(sess, name, cfg, P)
    # forward
    p_gpu = m.predict_proba(sess, Xc, Xf)
    p_ref = orc.forward(Xc, Xf).y_out.numpy()
    assert rel_err(p_gpu, p_ref) < TOL
    # loss + grads (dense vars and tables) at identical weights
    loss_ref, g_ref, _ = orc.loss_and_grads(Xc, Xf, y)
    loss_gpu = m.run(sess, Xc, Xf, y, True)
    assert abs(float(loss_gpu) - float(loss_ref)) < TOL * max(1.0, abs(float(loss_ref)))
    g_gpu = m.get_grads()
    for k, v in g_gpu.items():
        assert_close(v, g_ref[k].numpy(), TOL, k)
    # three optimizer steps: weights track the oracle
    orc.run(Xc, Xf, y, True)
    for _ in range(2):
        l_o = orc.run(Xc, Xf, y, True)
        l_g = m.run(sess, Xc, Xf, y, True)
        assert abs(float(l_o) - float(l_g)) < 1e-5
    P_gpu = m.get_params()
    assert_weights_track(P_gpu, orc.P)
    # eval path: (actual, round(y_out))
    a, pr = m.run(sess, Xc, Xf, y, False)
    a_o, pr_o = orc.run(Xc, Xf, y, False)
    assert np.array_equal(a, a_o) and np.array_equal(pr, pr_o)


def test_deepfm_smode_lazy_adam_matches_dense_on_touched_rows(sess):
    cfg = no_dropout(tiny_cfg('DeepFM', 'S'))
    P = O.init_params('DeepFM', cfg, seed=4, std=0.3)
    Xc, Xf, y = tiny_batch(cfg, B=8, oor=False)
    cfg.adam_mode = 'lazy'
    m1 = _build(sess, 'DeepFM', cfg, P)
    cfg2 = no_dropout(tiny_cfg('DeepFM', 'S'))
    cfg2.adam_mode = 'tf1'
    m2 = _build(sess, 'DeepFM', cfg2, P)
    # same batch every step => every touched row is touched at every step => lazy == dense
    for _ in range(3):
        m1.run(sess, Xc, Xf, y, True)
        m2.run(sess, Xc, Xf, y, True)
    # the row-wise and the whole-table update kernels are compiled separately (different fma contraction):
    # equal to the last bit or two, not bitwise
    for (k, a), b in zip(m1.get_params().items(), m2.get_params().values()):
        assert np.allclose(a, b, rtol=1e-6, atol=1e-8), k


def test_deepfm_dropout_masks_match_hash(sess):
    """Training-mode dropout uses the shared counter hash: feed the oracle the same masks."""
    cfg = tiny_cfg('DeepFM', 'S')
    cfg.adam_mode = 'tf1'
    P = O.init_params('DeepFM', cfg, seed=5, std=0.3)
    Xc, Xf, y = tiny_batch(cfg, B=16, seed=2)
    m = _build(sess, 'DeepFM', cfg, P)
    B, N, E = 16, cfg.cate_num, cfg.embed_size
    seed = sess.seed * 1000003 + 0
    masks = {'fm_first': O.keep_mask(seed, 1, B * N, 1 - cfg.dropout_fm).reshape(B, N),
             'fm_second': O.keep_mask(seed, 2, B * E, 1 - cfg.dropout_fm).reshape(B, E)}
    for i, L in enumerate(cfg.deep_layers):
        masks['deep_%d' % i] = O.keep_mask(seed, 10 + i, B * L, 1 - cfg.dropout_deep).reshape(B, L)
    masks = {k: torch.as_tensor(v) for k, v in masks.items()}
    orc = O.OracleModel('DeepFM', cfg, P, dtype=torch.float64)
    loss_ref, g_ref, _ = orc.loss_and_grads(Xc, Xf, y, masks)
    loss_gpu = m.run(sess, Xc, Xf, y, True)
    assert abs(float(loss_gpu) - float(loss_ref)) < 1e-5
    for k, v in m.get_grads().items():
        assert_close(v, g_ref[k].numpy(), TOL, k)


def test_afm_pair_and_output_dropout_match_hash(sess):
    """AFM in training mode: per-pair dropout on the (constant 1) attention weights and dropout on
    the pooled vector, both through the shared counter hash (AFM/AFM.py:80,84)."""
    cfg = tiny_cfg('AFM', 'S')
    cfg.dropout = 0.4
    P = {k: np.abs(v) for k, v in O.init_params('AFM', cfg, seed=6, std=0.1).items()}
    Xc, Xf, y = tiny_batch(cfg, B=16, seed=3, oor=False)
    Xf = np.abs(Xf) * 0.3
    m = _build(sess, 'AFM', cfg, P)
    B, N, E = 16, cfg.cate_num + cfg.cont_num, cfg.embed_size
    npairs = N * (N - 1) // 2
    seed = sess.seed * 1000003 + 0
    masks = {'att': torch.as_tensor(O.keep_mask(seed, 4, B * npairs, 0.6).reshape(B, npairs, 1)),
             'out': torch.as_tensor(O.keep_mask(seed, 5, B * E, 0.6).reshape(B, E))}
    orc = O.OracleModel('AFM', cfg, P, dtype=torch.float64)
    loss_ref, g_ref, out = orc.loss_and_grads(Xc, Xf, y, masks)
    assert torch.isfinite(loss_ref)
    loss_gpu = m.run(sess, Xc, Xf, y, True)
    assert abs(float(loss_gpu) - float(loss_ref)) < 1e-5 * max(1.0, abs(float(loss_ref)))
    for k, v in m.get_grads().items():
        assert_close(v, g_ref[k].numpy(), TOL, k)


@pytest.mark.parametrize('dropout', [0.0, 0.4])
def test_afm_attention_over_pairs_trains_the_attention_variables(sess, dropout):
    """attention='pairs' (softmax over the pair axis, not a reference flag): forward, gradients of
    every variable including attention_w/b/h, and Adagrad steps against the oracle's
    canonical_softmax form."""
    cfg = tiny_cfg('AFM', 'S')
    cfg.dropout = dropout
    cfg.attention = 'pairs'
    cfg.canonical_softmax = True
    P = {k: (v if 'attention' in k else np.abs(v)) for k, v in O.init_params('AFM', cfg, seed=8, std=0.3).items()}
    Xc, Xf, y = tiny_batch(cfg, B=16, seed=4, oor=False)
    Xf = np.abs(Xf) * 0.5
    m = _build(sess, 'AFM', cfg, P)
    B, N, E = 16, cfg.cate_num + cfg.cont_num, cfg.embed_size
    npairs = N * (N - 1) // 2
    masks = None
    if dropout:
        seed = sess.seed * 1000003 + 0
        masks = {'att': torch.as_tensor(O.keep_mask(seed, 4, B * npairs, 1 - dropout).reshape(B, npairs, 1)),
                 'out': torch.as_tensor(O.keep_mask(seed, 5, B * E, 1 - dropout).reshape(B, E))}
    orc = O.OracleModel('AFM', cfg, P, dtype=torch.float64)
    loss_ref, g_ref, _ = orc.loss_and_grads(Xc, Xf, y, masks)
    assert torch.isfinite(loss_ref)
    p_gpu = m.predict_proba(sess, Xc, Xf)
    assert rel_err(p_gpu, orc.forward(Xc, Xf).y_out.numpy()) < TOL
    loss_gpu = m.run(sess, Xc, Xf, y, True)
    assert abs(float(loss_gpu) - float(loss_ref)) < 1e-5 * max(1.0, abs(float(loss_ref)))
    g = m.get_grads()
    for k, v in g.items():
        assert_close(v, g_ref[k].numpy(), TOL, k)
    assert np.abs(g['attention_based_pooling/attention_w']).max() > 0
    if not dropout:
        orc.run(Xc, Xf, y, True)
        for _ in range(2):
            assert abs(float(orc.run(Xc, Xf, y, True)) - float(m.run(sess, Xc, Xf, y, True))) < 1e-5
        assert_weights_track(m.get_params(), orc.P)


def test_side_stream_weight_gradients_change_nothing(sess):
    """The towers' weight gradients on the side stream give the same bits as everything on one stream."""
    outs = []
    for enabled in (True, False):
        cfg = tiny_cfg('DeepFM', 'S')
        cfg.adam_mode = 'tf1'
        P = O.init_params('DeepFM', cfg, seed=9, std=0.3)
        Xc, Xf, y = tiny_batch(cfg, B=64, seed=3)
        m = _build(sess, 'DeepFM', cfg, P)
        sess.side_enabled = enabled
        try:
            for _ in range(3):
                m.run(sess, Xc, Xf, y, True)
        finally:
            sess.side_enabled = True
        outs.append(m.get_params())
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k


def test_graph_replayed_inference_matches_eager(sess):
    """predict_step captured into a HIP graph and replayed gives the eager result, for new inputs and after a
    training step changed the weights in place."""
    cfg = tiny_cfg('DeepFM', 'S')
    cfg.adam_mode = 'tf1'
    P = O.init_params('DeepFM', cfg, seed=6, std=0.3)
    m = _build(sess, 'DeepFM', cfg, P)
    for seed in (1, 2):
        Xc, Xf, y = tiny_batch(cfg, B=32, seed=seed)
        idx, xc, yy = m._upload(Xc, Xf, y)
        eager = m.predict_step(idx, xc).clone()
        graphed = m.predict_step_graphed(idx, xc).clone()
        assert torch.equal(eager, graphed)
        m.train_step(idx, xc, yy)
        assert torch.equal(m.predict_step(idx, xc).clone(), m.predict_step_graphed(idx, xc))


@pytest.mark.parametrize('name,mode', [('DeepFM', 'S'), ('DCN', 'S'), ('PNN', 'S'), ('DeepFM', 'R')])
def test_graph_replayed_training_matches_eager(sess, name, mode):
    """Five training steps replayed from a captured HIP graph (dropout on, Adam bias correction and decay changing
    every step through device words) leave the same bits as five eager steps."""
    outs = []
    for graphed in (False, True):
        cfg = tiny_cfg(name, mode)
        cfg.adam_mode = 'tf1'
        cfg.decay_steps = 2
        P = O.init_params(name, cfg, seed=8, std=0.3)
        m = _build(sess, name, cfg, P)
        losses = []
        for step in range(5):
            Xc, Xf, y = tiny_batch(cfg, B=48, seed=step)
            idx, xc, yy = m._upload(Xc, Xf, y)
            fn = m.train_step_graphed if graphed else m.train_step
            losses.append(float(fn(idx, xc, yy).item()))
        outs.append((losses, m.get_params(), m.global_step))
    assert outs[0][0] == outs[1][0]
    assert outs[0][2] == outs[1][2] == 5
    for k in outs[0][1]:
        assert np.array_equal(outs[0][1][k], outs[1][1][k]), k


def test_graph_replay_with_the_field_sort_and_the_every_row_adam(sess):
    """A batch of 2048 takes the field-by-field pair sort whose passes exchange their digit counts inside one launch,
    and adam_mode 'tf1' without a gradient buffer adds the marking and the pass over the rows without a gradient:
    four steps replayed from a captured graph leave the bits of four eager steps (weights, slots, row map)."""
    outs = []
    for graphed in (False, True):
        cfg = _criteo_like_cfg('DeepFM', 'DeepFM')
        cfg.adam_mode, cfg.table_grads = 'tf1', False
        P = O.init_params('DeepFM', cfg, seed=4, std=0.05)
        m = _build(sess, 'DeepFM', cfg, P)
        losses = []
        for step in range(4):
            Xc, Xf, y = tiny_batch(cfg, B=2048, seed=step)
            idx, xc, yy = m._upload(Xc, Xf, y)
            fn = m.train_step_graphed if graphed else m.train_step
            losses.append(float(fn(idx, xc, yy).item()))
        outs.append((losses, m.get_params(), m.tables.slot0.cpu().numpy(), m.tables.slot1.cpu().numpy(),
                     m.tables.touched.cpu().numpy()))
    assert outs[0][0] == outs[1][0]
    for k in outs[0][1]:
        assert np.array_equal(outs[0][1][k], outs[1][1][k]), k
    for i in (2, 3, 4):
        assert np.array_equal(outs[0][i], outs[1][i])
    assert set(np.unique(outs[0][4])) <= {0, 2}


def test_tf1_adam_without_the_dense_gradient_buffer(sess):
    """adam_mode 'tf1' moves every table row each step, as tf.train.AdamOptimizer does (DeepFM/DeepFM.py:32,100).  With
    table_grads=False it does so without the dense gradient buffer - the row-wise pass on the rows with a gradient,
    the decay pass on all others: the same weights and slots as the buffer form to within the last bits (the two
    row updates are compiled separately), rows that never saw a gradient identical, and still the oracle's AdamTF1."""
    outs = []
    for keep in (True, False):
        cfg = _criteo_like_cfg('DeepFM', 'DeepFM')
        cfg.dropout_fm = cfg.dropout_deep = 0.0
        cfg.adam_mode, cfg.table_grads = 'tf1', keep
        P = O.init_params('DeepFM', cfg, seed=5, std=0.05)
        m = _build(sess, 'DeepFM', cfg, P)
        assert (m.tables.grad is None) == (not keep)
        for step in range(3):
            Xc, Xf, y = tiny_batch(cfg, B=200, seed=step)
            m.train_step(*m._upload(Xc, Xf, y))
        if not keep:
            assert int((m.tables.touched & 1).sum().item()) == 0    # the step's marks are cleared after every step
        outs.append((m.get_params(), m.tables.slot0.cpu().numpy(), m.tables.slot1.cpu().numpy()))
    for k in outs[0][0]:
        assert np.allclose(outs[0][0][k], outs[1][0][k], rtol=1e-6, atol=1e-9), k
    assert rel_err(outs[1][1], outs[0][1]) < 1e-6 and rel_err(outs[1][2], outs[0][2]) < 1e-6
    orc = O.OracleModel('DeepFM', cfg, P, dtype=torch.float64)
    for step in range(3):
        Xc, Xf, y = tiny_batch(cfg, B=200, seed=step)
        orc.run(Xc, Xf, y, True)
    assert_weights_track(outs[1][0], orc.P)


@pytest.mark.parametrize('variant', ['tiny', 'criteo_like'])
def test_deepfm_dz_in_the_gemm_epilogue_leaves_the_same_bits(sess, variant):
    """DeepFM training steps with dZ's FM terms added in the epilogue of the dx0 product (the default on one process)
    and with the separate ctr_embed_bwd_dz pass (what sharded tables use): same losses, same weights, bit for bit,
    dropout on (gradient of DeepFM/DeepFM.py:55-66)."""
    outs = []
    for fuse in (True, False):
        if variant == 'tiny':
            cfg = tiny_cfg('DeepFM', 'S')
            B = 48
        else:
            cfg = _criteo_like_cfg('DeepFM', 'DeepFM')
            cfg.dropout_fm, cfg.dropout_deep = 0.5, 0.2      # the reference's rates (DeepFM/main.py:55-56)
            B = 300
        cfg.adam_mode = 'tf1'
        P = O.init_params('DeepFM', cfg, seed=21, std=0.3 if variant == 'tiny' else 0.05)
        m = _build(sess, 'DeepFM', cfg, P)
        m.fuse_dz = fuse
        m.fm_z_in_update = False             # the term with the looked-up row in the epilogue too: the same arithmetic
        losses = []
        for step in range(4):
            Xc, Xf, y = tiny_batch(cfg, B=B, seed=step)
            losses.append(float(m.train_step(*m._upload(Xc, Xf, y)).item()))
        outs.append((losses, m.get_params()))
    assert outs[0][0] == outs[1][0]
    for k in outs[0][1]:
        assert np.array_equal(outs[0][1][k], outs[1][1][k]), k


@pytest.mark.parametrize('adam_mode', ['tf1', 'lazy'])
def test_deepfm_fm_row_term_in_the_table_update(sess, adam_mode):
    """fm_z_in_update (an option; off by default): the dx0 product adds g_first + g_second * s and leaves -g_second * Z to the row
    update, which subtracts (sum of g_second over a row's lookups) * row once per table row (DeepFM/DeepFM.py:59-63:
    every lookup of a row sees the same Z row).  Against the form with the term in the product's epilogue: same losses
    (the forward pass is untouched) and weights within 1e-6 of each other after four steps with dropout on - the two
    differ only in where one multiplication is rounded; the table gradients against the float64 oracle are held by
    test_forward_grads_steps.  Run twice: the same bits."""
    outs = []
    for late in (True, True, False):
        cfg = _criteo_like_cfg('DeepFM', 'DeepFM')
        cfg.dropout_fm, cfg.dropout_deep = 0.5, 0.2
        cfg.adam_mode = adam_mode
        P = O.init_params('DeepFM', cfg, seed=21, std=0.05)
        m = _build(sess, 'DeepFM', cfg, P)
        m.fm_z_in_update = late
        losses = []
        for step in range(4):
            Xc, Xf, y = tiny_batch(cfg, B=300, seed=step)
            losses.append(float(m.train_step(*m._upload(Xc, Xf, y)).item()))
        outs.append((losses, m.get_params()))
    assert outs[0][0] == outs[1][0] and all(np.array_equal(outs[0][1][k], outs[1][1][k]) for k in outs[0][1])
    assert np.allclose(outs[0][0], outs[2][0], rtol=1e-6, atol=0)
    moved = 0.0
    for k in outs[0][1]:
        a, b = outs[0][1][k], outs[2][1][k]
        assert float(np.max(np.abs(a - b))) < 1e-6 * max(1.0, float(np.max(np.abs(b)))), k
        moved = max(moved, float(np.max(np.abs(b - P[k]))))
    assert moved > 1e-3                               # the steps did move the weights


@pytest.mark.gpu
def test_graph_replay_survives_workspace_growth(sess):
    """Train graph captured at a small batch, then a much larger evaluation batch makes the scratch workspace
    grow; later replays must still match eager steps (the outgrown scratch buffer stays allocated)."""
    outs = []
    for graphed in (False, True):
        cfg = tiny_cfg('DeepFM', 'S')
        P = O.init_params('DeepFM', cfg, seed=3, std=0.3)
        m = _build(sess, 'DeepFM', cfg, P)
        fn = m.train_step_graphed if graphed else m.train_step
        losses = []
        for step in range(6):
            Xc, Xf, y = tiny_batch(cfg, B=32, seed=step)
            losses.append(float(fn(*m._upload(Xc, Xf, y)).item()))
            if step == 2:
                sess.ws.get(2 * sess.ws.buf.numel())                     # what a large batch would do
                junk = torch.full((sess.ws.retired[-1].numel() // 4,), float('nan'), device=sess.device)
                Xc2, Xf2, y2 = tiny_batch(cfg, B=4096, seed=99)
                m.predict_step(*m._upload(Xc2, Xf2, y2)[:2])
                del junk
        outs.append((losses, m.get_params()))
    assert outs[0][0] == outs[1][0]
    for k in outs[0][1]:
        assert np.array_equal(outs[0][1][k], outs[1][1][k]), k


def _criteo_like_cfg(name, variant):
    """The reference's default architecture (256-128-64-32 tower, DeepFM/main.py:48-59 and siblings) on a
    Criteo-like field layout: 26 categorical + 13 continuous fields, embedding width 16, S semantics - the shapes at
    which the models take the chained tower kernels, the fused output unit and the matrix-core interaction kernels."""
    cfg = no_dropout(O.default_cfg(name, embed_size=16, mode='S'))
    rng = np.random.default_rng(11)
    cfg.cate_list = [int(v) for v in rng.integers(3, 200, 26)]
    cfg.cate_num, cfg.cont_num = 26, 13
    cfg.adam_mode = 'tf1'
    if variant == 'PNN_outer':
        cfg.inner_product = False
    return cfg


@pytest.mark.parametrize('variant', ['DeepFM', 'DCN', 'NFM', 'ESMM', 'PNN', 'PNN_outer', 'AFM'])
def test_reference_default_architecture_matches_oracle(sess, variant):
    """Forward, loss, every gradient and three optimizer steps at the reference's default layer sizes, B = 300."""
    name = variant.split('_')[0]
    cfg = _criteo_like_cfg(name, variant)
    P = O.init_params(name, cfg, seed=5, std=0.05)
    Xc, Xf, y = tiny_batch(cfg, B=300, seed=2)
    if name == 'AFM':          # AFM feeds log_loss an unbounded y_out (AFM/AFM.py:85-88): keep 0 < y_out < 1
        Xf = np.abs(Xf) * 0.1
        P = {k: np.abs(v) * 0.2 for k, v in P.items()}
    orc = O.OracleModel(name, cfg, P, dtype=torch.float64)
    m = _build(sess, name, cfg, P)
    p_ref = orc.forward(Xc, Xf).y_out.numpy()
    assert rel_err(m.predict_proba(sess, Xc, Xf), p_ref) < TOL
    loss_ref, g_ref, _ = orc.loss_and_grads(Xc, Xf, y)
    loss_gpu = m.run(sess, Xc, Xf, y, True)
    assert abs(float(loss_gpu) - float(loss_ref)) < TOL * max(1.0, abs(float(loss_ref)))
    for k, v in m.get_grads().items():
        assert_close(v, g_ref[k].numpy(), TOL, k)
    orc.run(Xc, Xf, y, True)
    for _ in range(2):
        l_o = orc.run(Xc, Xf, y, True)
        l_g = m.run(sess, Xc, Xf, y, True)
        assert abs(float(l_o) - float(l_g)) < 1e-5
    assert_weights_track(m.get_params(), orc.P)


def test_out_of_range_ids_are_counted_and_optionally_rejected(sess):
    """Standard lookup: an id outside [0, V_f) reads a zero row and is counted; with strict_ids run() raises
    IndexError after the step (the reference's one-hot block is silent, its embedding_lookup call sites raise)."""
    cfg = no_dropout(tiny_cfg('DeepFM', 'S'))
    P = O.init_params('DeepFM', cfg, seed=3, std=0.3)
    Xc, Xf, y = tiny_batch(cfg, B=8)                 # two out-of-range ids
    m = _build(sess, 'DeepFM', cfg, P)
    m.run(sess, Xc, Xf, y, True)
    assert m.out_of_range_lookups() == 2 and m.out_of_range_lookups() == 0
    cfg.strict_ids = True
    m = _build(sess, 'DeepFM', cfg, P)
    with pytest.raises(IndexError):
        m.run(sess, Xc, Xf, y, True)
    Xc2, Xf2, y2 = tiny_batch(cfg, B=8, oor=False)
    assert np.isfinite(m.run(sess, Xc2, Xf2, y2, True))
    with pytest.raises(ValueError):                  # the one-hot block reads row 1 of every table
        cfg_r = tiny_cfg('DeepFM', 'R')
        cfg_r.cate_list = [3, 1, 4]
        _build(sess, 'DeepFM', cfg_r, None)


def test_captured_sizes_and_buffers_are_bounded(sess):
    """A caller whose batch size keeps changing: at most GRAPH_SIZES_PER_MODEL sizes are captured (the rest run
    eagerly, same bits), and buffers of sizes without a graph are dropped least-recently-used first."""
    from recommendation_amd import model_base as mb
    outs = []
    sizes = [8, 16, 24, 32, 40, 48, 56, 64, 72, 80, 88, 96, 104, 112, 8, 16]
    for graphed in (False, True):
        cfg = tiny_cfg('DeepFM', 'S')
        cfg.adam_mode = 'tf1'
        P = O.init_params('DeepFM', cfg, seed=8, std=0.3)
        m = _build(sess, 'DeepFM', cfg, P)
        losses = []
        for rep in range(2):
            for B in sizes:
                Xc, Xf, y = tiny_batch(cfg, B=B, seed=B + rep)
                idx, xc, yy = m._upload(Xc, Xf, y)
                losses.append(float((m.train_step_graphed if graphed else m.train_step)(idx, xc, yy).item()))
        outs.append((losses, m.get_params()))
        if graphed:
            assert len(m._train_graphs) == mb.GRAPH_SIZES_PER_MODEL
            assert len(m._bufs) <= mb.GRAPH_SIZES_PER_MODEL + mb.BUFFER_SIZES_PER_MODEL
            assert len(m._stage) <= 3 * mb.BUFFER_SIZES_PER_MODEL
    assert outs[0][0] == outs[1][0]
    for k in outs[0][1]:
        assert np.array_equal(outs[0][1][k], outs[1][1][k]), k

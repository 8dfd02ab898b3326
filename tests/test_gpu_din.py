"""DIN on the HIP path vs the CPU oracle (DIN/DIN.py): train (loss, dense gradients, SGD steps on
dense variables and on the item/category/bias tables), eval (mf_auc, p_and_n), test (logits over all items),
padding-invariance."""
import numpy as np
import pytest
import torch

from oracle import ctr_oracle as O
from tests.helpers import assert_close, rel_err

pytestmark = pytest.mark.gpu


def _setup(sess, H, seed=0, item_count=40, cate_count=6, B=9, T=11, fused=True):
    import recommendation_amd as ra
    ra.enable_zoo_imports()
    from DIN import DIN
    rng = np.random.default_rng(seed)
    cfg = O.default_cfg('DIN')
    cfg.hidden_units = H
    cfg.fused_attention = fused
    cate_list = rng.integers(0, cate_count, item_count)
    P = O.din_init_params(cfg, 5, item_count, cate_count, seed=seed + 1)
    # non-trivial values for the variables that initialise to constants
    for k in P:
        if k.endswith('bias') or k.endswith('beta') or 'alphadice' in k or k.endswith('item_b'):
            P[k] = rng.standard_normal(P[k].shape) * 0.2
        if k.endswith('gamma'):
            P[k] = 1.0 + rng.standard_normal(P[k].shape) * 0.2
    m = DIN.DIN(cfg, 5, item_count, cate_count, cate_list)
    m.initialize(sess, params=P)
    orc = O.OracleDIN(cfg, cate_list, P, dtype=torch.float64)
    sl = rng.integers(1, T + 1, B)
    sl[0] = T
    hist = np.zeros((B, T), dtype=np.int64)
    for b in range(B):
        hist[b, :sl[b]] = rng.integers(0, item_count, sl[b])
    i = rng.integers(0, item_count, B)
    j = rng.integers(0, item_count, B)
    y = rng.integers(0, 2, B)
    u = np.zeros(B, dtype=np.int64)
    return m, orc, (u, i, y, hist, sl), (u, i, j, hist, sl)


@pytest.mark.parametrize('H,B,T,fused', [(8, 9, 11, True), (32, 9, 11, True), (32, 300, 200, True), (64, 40, 200, True),
                                         (128, 64, 200, True), (128, 20, 50, False)])
def test_din_train_and_eval_match_oracle(sess, H, B, T, fused):
    """The T = 200 cases have the history length of the Amazon-Books benchmark shape: the per-sample block kernels
    walk several tiles of positions.  H = 8 and the last case go layer by layer through the materialised
    [B*T, 4H] (no fused scoring network for that width / fused_attention=False)."""
    m, orc, uij, uij_eval = _setup(sess, H, B=B, T=T, item_count=40 if B < 100 else 5000, cate_count=6 if B < 100 else 80,
                                   fused=fused)
    assert m.fused_attention == (fused and H >= 16)
    auc_g, pn_g = m.eval(sess, uij_eval)
    auc_o, pn_o = orc.eval(uij_eval)
    assert abs(float(auc_g) - float(auc_o)) < 1e-6
    assert rel_err(pn_g, pn_o) < 1e-5
    loss_o, g_o, _ = orc.loss_and_grads(uij)
    loss_g = m.train(sess, uij, 0.1)
    assert abs(float(loss_g) - float(loss_o)) < 1e-5
    gmax = max(float(g_o[k].abs().max()) for k in g_o)
    for k, v in m.get_grads().items():
        if k in g_o:      # the f3_att bias gradient is exactly 0 (softmax is shift invariant): absolute floor
            ref = g_o[k].numpy()
            if np.max(np.abs(ref)) < 1e-12:
                assert np.max(np.abs(v)) < 1e-5 * gmax, k
            else:
                assert_close(v, ref, 1e-5, k)
    orc.train(uij, 0.1)
    for _ in range(2):
        lo = orc.train(uij, 0.1)
        lg = m.train(sess, uij, 0.1)
        assert abs(float(lo) - float(lg)) < 1e-5
    for k, v in m.get_params().items():
        assert rel_err(v, orc.P[k].numpy()) < 1e-5, k          # plain SGD: no amplification of the gradient's rounding


@pytest.mark.parametrize('H,tile_rows', [(8, 1 << 17), (32, 64)])
def test_din_logits_over_all_items_match_oracle(sess, H, tile_rows):
    """test() (DIN/DIN.py:73-89,133-138): every user row against every item, with the `_all` Dice alphas; the second
    case forms the [B, items, 2H] input in several item tiles with a ragged last one (40 items, 7 per tile)."""
    m, orc, uij, _ = _setup(sess, H, seed=5)
    u, i, _, hist, sl = uij
    m.ITEM_TILE_ROWS = tile_rows
    got = m.test(sess, u, hist, sl, i=i)
    want = orc.test(i, hist, sl)
    assert got.shape == want.shape == (len(i), 40) and got.dtype == np.float32
    assert rel_err(got, want) < 1e-5
    # the branch has alphas of its own: the target item's column differs from the training branch's score
    _, pn = m.eval(sess, (u, i, i, hist, sl))
    assert np.max(np.abs(got[np.arange(len(i)), i] - pn[:, 0])) > 1e-4
    # fed as the reference feeds it (no target items) the call stops, as the reference's session run does
    with pytest.raises(ValueError, match="placeholder 'i'"):
        m.test(sess, u, hist, sl)


def test_din_padding_positions_do_not_matter(sess):
    m, orc, uij, _ = _setup(sess, 8, seed=3)
    u, i, y, hist, sl = uij
    hist2 = hist.copy()
    for b in range(hist.shape[0]):
        hist2[b, sl[b]:] = 7                      # garbage ids behind the valid prefix
    p0 = m.get_params()
    l1 = m.train(sess, (u, i, y, hist, sl), 0.0)
    l2 = m.train(sess, (u, i, y, hist2, sl), 0.0)
    assert abs(float(l1) - float(l2)) < 1e-6


def test_din_graph_replay_matches_eager_steps(sess):
    """train_step_graphed (one captured graph per (B, T), learning rate through a device word) against train_step on
    a second model: the same bits after four steps on changing batches and two learning rates."""
    B, T, H = 64, 50, 32
    m1, _, uij, _ = _setup(sess, H, seed=11, B=B, T=T, item_count=500, cate_count=20)
    m2, _, _, _ = _setup(sess, H, seed=11, B=B, T=T, item_count=500, cate_count=20)
    rng = np.random.default_rng(5)
    dev = sess.device
    for step, lr in enumerate((0.1, 0.1, 0.05, 0.05)):
        sl = rng.integers(1, T + 1, B)
        hist = rng.integers(0, 500, (B, T))
        hist[np.arange(T)[None, :] >= sl[:, None]] = 0
        args = (torch.as_tensor(rng.integers(0, 500, B).astype(np.int32)).to(dev),
                torch.as_tensor(rng.integers(0, 2, B).astype(np.float32)).to(dev),
                torch.as_tensor(hist.astype(np.int32)).to(dev), torch.as_tensor(sl.astype(np.int32)).to(dev))
        l1 = m1.train_step(*args, lr)
        l2 = m2.train_step_graphed(*args, lr)
        assert float(l1.item()) == float(l2.item()), step
    assert len(m2._train_graphs) == 1
    p1, p2 = m1.get_params(), m2.get_params()
    for k in p1:
        assert np.array_equal(p1[k], p2[k]), k


@pytest.mark.gpu
def test_din_captured_shape_survives_buffer_cache_pressure(sess):
    """A captured (B, T) step keeps its buffers while nine other shapes pass through eval() and train_step() - more
    than the per-shape buffer cache holds - and its next replay still leaves the bits of an eager model that saw
    the same steps.  initialize() on a captured model releases its graphs and buffers."""
    B, T, H = 32, 20, 32
    m1, _, _, _ = _setup(sess, H, seed=3, B=B, T=T, item_count=300, cate_count=12)
    m2, _, _, _ = _setup(sess, H, seed=3, B=B, T=T, item_count=300, cate_count=12)
    rng = np.random.default_rng(9)
    dev = sess.device

    def batch(b, t):
        sl = rng.integers(1, t + 1, b)
        hist = rng.integers(0, 300, (b, t))
        hist[np.arange(t)[None, :] >= sl[:, None]] = 0
        return (torch.as_tensor(rng.integers(0, 300, b).astype(np.int32)).to(dev),
                torch.as_tensor(rng.integers(0, 2, b).astype(np.float32)).to(dev),
                torch.as_tensor(hist.astype(np.int32)).to(dev), torch.as_tensor(sl.astype(np.int32)).to(dev))

    for _ in range(2):                                   # eager step + capture, then one replay
        args = batch(B, T)
        m1.train_step(*args, 0.1)
        m2.train_step_graphed(*args, 0.1)
    assert list(m2._train_graphs) == [(B, T)] and (B, T) in m2._bufs.pinned
    for k in range(9):                                   # nine other shapes, alternately evaluated and trained
        b, t = 8 + k, 5 + 2 * k
        i, y, hist, sl = batch(b, t)
        if k % 2:
            for m in (m1, m2):
                m.train_step(i, y, hist, sl, 0.05)
        else:
            j = torch.as_tensor(rng.integers(0, 300, b).astype(np.int32)).to(dev)
            for m in (m1, m2):
                m.eval(sess, (None, i, j, hist, sl))
    assert (B, T) in m2._bufs and len(m2._bufs) <= m2._bufs.maxsize + 1
    for _ in range(2):                                   # replays against the buffers captured above
        args = batch(B, T)
        l1 = m1.train_step(*args, 0.1)
        l2 = m2.train_step_graphed(*args, 0.1)
        assert float(l1.item()) == float(l2.item())
    p1, p2 = m1.get_params(), m2.get_params()
    for k in p1:
        assert np.array_equal(p1[k], p2[k]), k
    m2.initialize(sess, params=p1)
    assert not m2._train_graphs and not m2._bufs and not m2._bufs.pinned and not sess._retired
    args = batch(B, T)
    assert float(m1.train_step(*args, 0.1).item()) == float(m2.train_step_graphed(*args, 0.1).item())


@pytest.mark.parametrize('rows,cols', [(1, 5), (300, 64), (2048, 80), (5000, 40)])
def test_one_launch_backward_of_batch_norm_and_dice(sess, rows, cols):
    """ctr_bn_infer_bwd and ctr_dice_bwd_sum (the per-column layers after the attention, DIN/DIN.py:45,53-57,149-166)
    against float64 formulas; inputs are column slices of wider tensors, as the model passes them."""
    from recommendation_amd import ops
    rng = np.random.default_rng(rows + cols)
    dev = sess.device
    eps = 1e-3

    def t(a):
        wide = torch.zeros(a.shape[0], a.shape[1] + 3, dtype=torch.float32, device=dev)
        wide[:, 1:1 + a.shape[1]] = torch.as_tensor(a.astype(np.float32))
        return wide[:, 1:1 + a.shape[1]]

    x, dy = rng.standard_normal((rows, cols)), rng.standard_normal((rows, cols))
    gamma, var, alpha = 1 + 0.2 * rng.standard_normal(cols), rng.uniform(0.5, 2.0, cols), 0.3 * rng.standard_normal(cols)
    v = lambda a: torch.as_tensor(a.astype(np.float32)).to(dev)
    dx, dgamma, dbeta = t(np.zeros((rows, cols))), v(np.zeros(cols)), v(np.zeros(cols))
    ops.bn_infer_bwd(t(dy), t(x), dx, gamma=v(gamma), var=v(var), eps=eps, dgamma=dgamma, dbeta=dbeta)
    x32, dy32 = x.astype(np.float32).astype(np.float64), dy.astype(np.float32).astype(np.float64)
    g32, v32, a32 = (a.astype(np.float32).astype(np.float64) for a in (gamma, var, alpha))
    assert_close(dx.cpu().numpy(), g32 * dy32 / np.sqrt(v32 + eps), 1e-5, 'bn dx')
    assert_close(dgamma.cpu().numpy(), (dy32 * x32).sum(0), 1e-5, 'dgamma')
    assert_close(dbeta.cpu().numpy(), dy32.sum(0), 1e-5, 'dbeta')
    dx2, dalpha = t(np.zeros((rows, cols))), v(np.zeros(cols))
    ops.dice_bwd_sum(t(x), v(alpha), t(dy), dx2, dalpha, eps)
    c0 = 1.0 / np.sqrt(1.0 + eps)
    p = 1.0 / (1.0 + np.exp(-x32 * c0))
    want_dx = dy32 * (a32 * (1 - p) + p + x32 * (1 - a32) * p * (1 - p) * c0)
    assert_close(dx2.cpu().numpy(), want_dx, 1e-5, 'dice dx')
    assert_close(dalpha.cpu().numpy(), (dy32 * (1 - p) * x32).sum(0), 1e-5, 'dalpha')
    # the same bits on a second call: the column sums have a fixed order
    dalpha2 = v(np.zeros(cols))
    ops.dice_bwd_sum(t(x), v(alpha), t(dy), dx2, dalpha2, eps)
    assert torch.equal(dalpha, dalpha2)


def test_din_batch_norm_and_dice_backward_paths_agree(sess):
    """Above SMALL_ROWS the model takes the elementwise pass + column sums; both paths give the oracle's gradients."""
    m, orc, uij, _ = _setup(sess, 32, seed=21, B=40, T=30)
    m.SMALL_ROWS = 0
    _, g_o, _ = orc.loss_and_grads(uij)
    m.train(sess, uij, 0.1)
    for k, val in m.get_grads().items():
        if k in g_o and float(g_o[k].abs().max()) > 1e-12:
            assert_close(val, g_o[k].numpy(), 1e-5, k)

"""The oracle checked against itself three ways (it is the only pin there is: the reference has
no tests or fixtures and cannot run here): literal graph emulation vs closed forms, autograd vs
hand-derived formulas, and float64 finite differences; plus the committed golden vectors."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import ctr_oracle as O
from tests.helpers import no_dropout, rel_err, tiny_batch, tiny_cfg

MODELS = ['DeepFM', 'DCN', 'NFM', 'PNN', 'AFM', 'ESMM']


@pytest.mark.parametrize('model', MODELS)
def test_literal_graph_equals_closed_form(model):
    """R-mode: one_hot -> lookup -> multiply -> concat, the [B,D,D] cross product and the Python
    pair loops, evaluated literally, give the closed forms the kernels implement."""
    cfg = no_dropout(tiny_cfg(model, 'R'))
    P = O.init_params(model, cfg, seed=2, std=0.3)
    Xc, Xf, y = tiny_batch(cfg)
    a = O.OracleModel(model, cfg, P, literal=True)
    b = O.OracleModel(model, cfg, P, literal=False)
    la, ga, oa = a.loss_and_grads(Xc, Xf, y)
    lb, gb, ob = b.loss_and_grads(Xc, Xf, y)
    assert torch.allclose(oa.y_out, ob.y_out, atol=1e-12, equal_nan=True)
    for k in ga:
        assert torch.allclose(ga[k], gb[k], atol=1e-12, equal_nan=True), k


def test_rmode_only_row_one_is_read_and_trained():
    cfg = no_dropout(tiny_cfg('DeepFM', 'R'))
    P = O.init_params('DeepFM', cfg, seed=2, std=0.3)
    Xc, Xf, y = tiny_batch(cfg)
    m = O.OracleModel('DeepFM', cfg, P)
    _, g, out = m.loss_and_grads(Xc, Xf, y)
    for f, V in enumerate(cfg.cate_list):
        gt = g['embedding/embed_cate_w_%d' % f]
        assert torch.count_nonzero(gt[0]) == 0 and torch.count_nonzero(gt[2:]) == 0
        assert torch.count_nonzero(gt[1]) > 0
    # out-of-range ids give all-zero rows (tf.one_hot), in-range ids copy W_f[1] bit for bit
    Z = out.Z.detach().numpy()
    off = np.concatenate([[0], np.cumsum(cfg.cate_list)])
    assert np.all(Z[0, off[0]:off[1]] == 0) and np.all(Z[1, off[1]:off[2]] == 0)
    assert np.array_equal(Z[2, off[2] + 0], np.asarray(P['embedding/embed_cate_w_2'])[1])


def test_fm_and_cross_hand_gradients_match_autograd():
    rng = np.random.default_rng(0)
    B, N, E = 5, 4, 3
    Z = torch.tensor(rng.standard_normal((B, N, E)), requires_grad=True)
    g1 = torch.tensor(rng.standard_normal((B, N)))
    g2 = torch.tensor(rng.standard_normal((B, E)))
    s = Z.sum(1)
    first, second = Z.sum(2), 0.5 * (s * s - (Z * Z).sum(1))
    ((first * g1).sum() + (second * g2).sum()).backward()
    hand = g2[:, None, :] * (s.detach()[:, None, :] - Z.detach()) + g1[:, :, None]
    assert torch.allclose(Z.grad, hand, atol=1e-12)
    # cross layer: dx_l = g + t w^T, dx0 += g s_l, dw = sum_b t x_l, db = sum_b g, t = g . x0
    D, L = 6, 3
    x0 = torch.tensor(rng.standard_normal((B, D)), requires_grad=True)
    ws = [torch.tensor(rng.standard_normal((D, 1)) * 0.3, requires_grad=True) for _ in range(L)]
    bs = [torch.tensor(rng.standard_normal((D, 1)) * 0.3, requires_grad=True) for _ in range(L)]
    gL = torch.tensor(rng.standard_normal((B, D)))
    (O.dcn_cross(x0, ws, bs) * gL).sum().backward()
    xs = [x0.detach()]
    for w, b in zip(ws, bs):
        xs.append(x0.detach() * (xs[-1] @ w.detach()) + b.detach().view(1, -1) + xs[-1])
    g, dx0 = gL.clone(), torch.zeros(B, D, dtype=torch.float64)
    for l in range(L - 1, -1, -1):
        t = (g * x0.detach()).sum(1, keepdim=True)
        assert torch.allclose(ws[l].grad.view(-1), (t * xs[l]).sum(0), atol=1e-10)
        assert torch.allclose(bs[l].grad.view(-1), g.sum(0), atol=1e-10)
        dx0 += g * (xs[l] @ ws[l].detach())
        g = g + t * ws[l].detach().view(1, -1)
    assert torch.allclose(x0.grad, dx0 + g, atol=1e-10)


@pytest.mark.parametrize('model', ['DeepFM', 'DCN', 'NFM', 'PNN', 'ESMM'])
def test_finite_differences(model):
    cfg = no_dropout(tiny_cfg(model, 'S'))
    P = O.init_params(model, cfg, seed=4, std=0.3)
    Xc, Xf, y = tiny_batch(cfg, oor=False)
    m = O.OracleModel(model, cfg, P)
    _, g, _ = m.loss_and_grads(Xc, Xf, y)
    rng = np.random.default_rng(1)
    for k in list(P)[:2] + list(P)[-3:]:
        v = np.asarray(P[k])
        ix = tuple(rng.integers(0, s) for s in v.shape)
        if k.startswith('embedding/embed_cate') and ix[0] not in set(Xc[:, int(k.rsplit('_', 1)[1])]):
            continue
        h = 1e-6
        vals = []
        for sgn in (+1, -1):
            P2 = {kk: np.array(vv, dtype=np.float64) for kk, vv in P.items()}
            P2[k][ix] += sgn * h
            vals.append(O.OracleModel(model, cfg, P2).loss_and_grads(Xc, Xf, y)[0].item())
        fd = (vals[0] - vals[1]) / (2 * h)
        assert abs(fd - g[k][ix].item()) < 1e-6 * max(1.0, abs(fd)), (k, ix)


def test_tf1_optimizers_and_schedule():
    p = {'w': torch.tensor([1.0, -2.0], dtype=torch.float64)}
    g = {'w': torch.tensor([0.5, 0.25], dtype=torch.float64)}
    a = O.AdamTF1(p)
    a.apply(p, g, 0.001)
    # first Adam step: m = .1 g, v = .001 g^2, lr_t = lr*sqrt(1-.999)/(1-.9); eps outside the sqrt
    lr_t = 0.001 * np.sqrt(1 - 0.999) / (1 - 0.9)
    exp = np.array([1.0, -2.0]) - lr_t * (0.1 * np.array([0.5, 0.25])) / (np.sqrt(0.001) * np.array([0.5, 0.25]) + 1e-8)
    assert np.allclose(p['w'].numpy(), exp, atol=1e-15)
    p = {'w': torch.tensor([1.0], dtype=torch.float64)}
    ad = O.AdagradTF1(p)
    ad.apply(p, {'w': torch.tensor([2.0], dtype=torch.float64)}, 0.1)
    assert np.allclose(p['w'].numpy(), 1.0 - 0.1 * 2.0 / np.sqrt(0.1 + 4.0))
    assert O.exponential_decay(0.001, 9999, 10000, 0.99) == 0.001
    assert np.isclose(O.exponential_decay(0.001, 20000, 10000, 0.99), 0.001 * 0.99 ** 2)


def test_afm_attention_is_identically_one_and_gets_no_gradient():
    cfg = no_dropout(tiny_cfg('AFM', 'S'))
    P = {k: np.abs(v) for k, v in O.init_params('AFM', cfg, seed=3, std=0.3).items()}
    Xc, Xf, y = tiny_batch(cfg, oor=False)
    m = O.OracleModel('AFM', cfg, P)
    _, g, out = m.loss_and_grads(Xc, np.abs(Xf) * 0.3, y)
    assert torch.all(out.a.detach() == 1.0)
    for k in g:
        if 'attention' in k:
            assert torch.count_nonzero(g[k]) == 0
    Zd = out.Z.detach()
    s = Zd.sum(1)
    assert torch.allclose(out.y_out.detach(), (0.5 * (s * s - (Zd ** 2).sum(1))).sum(1), atol=1e-12)


def test_keep_mask_matches_the_library_hash():
    from recommendation_amd import _lib
    lib = _lib.load()
    m = O.keep_mask(12345, 7, 200, 0.7)
    thr = O.keep_threshold(0.7)
    ref = np.array([lib.ctr_keep_hash24(12345, 7, i) < thr for i in range(200)], dtype=np.float32)
    assert np.array_equal(m, ref)
    # indices beyond 32 bits and large seeds take the same path on both sides
    big = np.array([(1 << 32) + 5, (1 << 40) + 12345, (1 << 63) + 1], dtype=np.uint64)
    assert [int(v) for v in O.hash24(2 ** 60 + 3, 99, big)] == [lib.ctr_keep_hash24(2 ** 60 + 3, 99, int(i)) for i in big]
    # keep rate, and independence of neighbouring elements / streams (no visible lattice)
    a = O.keep_mask(1, 2, 200000, 0.8)
    b = O.keep_mask(1, 3, 200000, 0.8)
    assert abs(a.mean() - 0.8) < 5e-3 and abs(b.mean() - 0.8) < 5e-3
    assert abs((a[1:] * a[:-1]).mean() - 0.64) < 5e-3
    assert abs((a * b).mean() - 0.64) < 5e-3


def test_din_attention_masks_padding_and_auc():
    rng = np.random.default_rng(0)
    cfg = O.default_cfg('DIN')
    cfg.hidden_units = 8
    P = O.din_init_params(cfg, 10, 30, 5, seed=1)
    cate_list = rng.integers(0, 5, 30)
    m = O.OracleDIN(cfg, cate_list, P)
    B, T = 6, 7
    hist = rng.integers(0, 30, (B, T))
    sl = np.array([7, 3, 1, 5, 2, 7])
    i = rng.integers(0, 30, B)
    yv = rng.integers(0, 2, B)
    l1, g, o1 = m.loss_and_grads((None, i, yv, hist, sl))
    hist2 = hist.copy()
    for b in range(B):
        hist2[b, sl[b]:] = 0                      # whatever sits in the padded positions is ignored
    l2, _, o2 = m.loss_and_grads((None, i, yv, hist2, sl))
    assert torch.allclose(o1.logit.detach(), o2.logit.detach(), atol=1e-12)
    auc, pn = m.eval((None, i, rng.integers(0, 30, B), hist, sl))
    assert 0.0 <= auc <= 1.0 and pn.shape == (B, 2)


def test_din_logits_over_all_items_reduce_to_the_scored_branch():
    """logits_all (DIN/DIN.py:73-89) runs the FCN of the scored branch over every item with Dice alphas of its own:
    with those alphas set equal to the `_i` ones, the target item's column is score_i (DIN/DIN.py:93)."""
    rng = np.random.default_rng(2)
    cfg = O.default_cfg('DIN')
    cfg.hidden_units = 8
    P = O.din_init_params(cfg, 10, 30, 5, seed=1)
    for k in list(P):
        if 'alphadice' in k and k.endswith('_i'):
            P[k] = rng.standard_normal(P[k].shape) * 0.3
            P[k[:-2] + '_all'] = P[k].copy()
    P['embedding/item_b'] = rng.standard_normal(30) * 0.2
    cate_list = rng.integers(0, 5, 30)
    m = O.OracleDIN(cfg, cate_list, P)
    B, T = 6, 7
    hist, sl, i = rng.integers(0, 30, (B, T)), np.array([7, 3, 1, 5, 2, 7]), rng.integers(0, 30, B)
    la = m.test(i, hist, sl)
    assert la.shape == (B, 30)
    _, pn = m.eval((None, i, i, hist, sl))
    assert np.allclose(la[np.arange(B), i], pn[:, 0], rtol=1e-6, atol=1e-7)
    P['fcn/alphadice_1_all'] = P['fcn/alphadice_1_all'] + 0.5
    assert not np.allclose(O.OracleDIN(cfg, cate_list, P).test(i, hist, sl), la)


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'golden', '*.npz'))))
def test_oracle_reproduces_golden(path):
    from tests.golden.make_golden import make
    base = os.path.basename(path)[:-4]
    parts = base.split('_')
    model, mode = parts[0], parts[-1]
    extra = {'inner_product': False} if 'outer' in parts else {}
    if 'pairs' in parts:
        extra = {'attention': 'pairs', 'canonical_softmax': True}
    new = make(model, mode, extra)
    old = np.load(path)
    assert set(old.files) == set(new)
    for k in old.files:
        assert np.allclose(old[k], new[k], rtol=1e-10, atol=1e-12, equal_nan=True), k
    # the float32 shadow stays within the 1e-5 bar of the float64 master
    if model != 'AFM':
        cfg = no_dropout(tiny_cfg(model, mode, **extra))
        P = {k[6:]: old[k] for k in old.files if k.startswith('param/')}
        m32 = O.OracleModel(model, cfg, P, dtype=torch.float32)
        assert rel_err(m32.forward(old['X_cate'], old['X_cont']).y_out.numpy(), old['y_out']) < 1e-5


def test_ftrl_matches_the_closed_form_and_touches_only_looked_up_rows():
    w = {'t': torch.tensor([[0.3, -0.2], [0.1, 0.4], [0.05, 0.0]], dtype=torch.float64)}
    opt = O.FtrlTF1(w, l1=0.1)
    g = {'t': torch.tensor([[2.0, -3.0], [0.0, 0.0], [1.0, 1.0]], dtype=torch.float64)}
    rows = {'t': torch.tensor([True, True, False])}
    w0 = w['t'].clone()
    opt.apply(w, g, 0.5, rows=rows)
    n0, lr = 0.1, 0.5
    for r in (0, 1):
        for c in (0, 1):
            gi, wi = float(g['t'][r, c]), float(w0[r, c])
            nn = n0 + gi * gi
            z = gi - (nn ** 0.5 - n0 ** 0.5) / lr * wi
            ref = 0.0 if abs(z) <= 0.1 else -(z - np.sign(z) * 0.1) / (nn ** 0.5 / lr)
            assert abs(float(w['t'][r, c]) - ref) < 1e-12
    assert torch.equal(w['t'][2], w0[2]) and float(opt.n['t'][2, 0]) == 0.1     # untouched row
    assert float(w['t'][1, 0]) == 0.0                                           # touched, zero gradient: |z| <= l1


def test_wide_deep_oracle_gradients_match_finite_differences():
    cfg = O.default_cfg('Wide_Deep', embed_size=2, hidden_layers=[3], dropout=0.0)
    cfg.cross_list, cfg.cate_list, cfg.cont_num = [4], [3, 2], 1
    P = O.wide_deep_init_params(cfg, seed=0, std=0.5)
    m = O.OracleWideDeep(cfg, P)
    rng = np.random.default_rng(0)
    B = 5
    Xr, Xc = rng.integers(0, 4, (B, 1)), np.stack([rng.integers(0, 3, B), rng.integers(0, 2, B)], 1)
    Xf, y = rng.standard_normal((B, 1)), rng.integers(0, 2, B)
    loss, g, _ = m.loss_and_grads(Xr, Xc, Xf, y)
    assert torch.count_nonzero(g['wide_net/wide_cross_b_0']) == 0               # never enters the graph
    for k in ('wide_net/wide_output_w', 'deep_net/deep_hidden_w_0', 'wide_net/wide_cate_w_1'):
        idx = tuple(0 for _ in m.P[k].shape)
        eps = 1e-6
        old = m.P[k][idx].item()
        m.P[k][idx] = old + eps
        lp = m.loss_and_grads(Xr, Xc, Xf, y)[0]
        m.P[k][idx] = old - eps
        lm = m.loss_and_grads(Xr, Xc, Xf, y)[0]
        m.P[k][idx] = old
        assert abs((lp - lm).item() / (2 * eps) - g[k][idx].item()) < 1e-7, k


def test_lazy_adam_freezes_unseen_rows_and_equals_dense_adam_on_rows_seen_every_step():
    """LazyAdamTF1: a row looked up at every step follows AdamTF1 exactly; a row never looked up keeps weight,
    m and v; a row seen at steps 1 and 3 only skips step 2 altogether (TF-1.x Adam would decay m, v and move it)."""
    import math
    from collections import OrderedDict
    torch.manual_seed(0)
    W = OrderedDict(t=torch.randn(4, 3, dtype=torch.float64), d=torch.randn(5, dtype=torch.float64))
    P1 = OrderedDict((k, v.clone()) for k, v in W.items())
    P2 = OrderedDict((k, v.clone()) for k, v in W.items())
    lazy, dense = O.LazyAdamTF1(P1), O.AdamTF1(P2)
    seen = [torch.tensor([True, False, True, False]), torch.tensor([True, False, False, False]),
            torch.tensor([True, False, True, False])]
    gs = []
    for s in seen:
        g = OrderedDict(t=torch.randn(4, 3, dtype=torch.float64) * s.view(-1, 1), d=torch.randn(5, dtype=torch.float64))
        gs.append(g['t'][2].clone())
        lazy.apply(P1, g, 0.01, rows={'t': s})
        dense.apply(P2, g, 0.01)
    assert lazy.t == 3
    assert torch.equal(P1['d'], P2['d'])                       # dense variable: identical
    assert torch.equal(P1['t'][0], P2['t'][0])                 # seen every step: identical
    assert torch.equal(P1['t'][1], W['t'][1]) and torch.equal(P1['t'][3], W['t'][3])
    assert torch.all(lazy.m['t'][1] == 0) and torch.all(lazy.v['t'][1] == 0)
    assert not torch.allclose(P1['t'][2], P2['t'][2])
    # row 2 by hand: two updates, bias correction from the GLOBAL update counts 1 and 3
    m = torch.zeros(3, dtype=torch.float64)
    v = torch.zeros(3, dtype=torch.float64)
    w = W['t'][2].clone()
    for t in (1, 3):
        g = gs[t - 1]
        m = 0.9 * m + 0.1 * g
        v = 0.999 * v + 0.001 * g * g
        w = w - 0.01 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) * m / (torch.sqrt(v) + 1e-8)
    assert torch.allclose(P1['t'][2], w, rtol=1e-14, atol=0)


def test_bag_pooling_reduces_to_the_single_id_lookup_and_sums_members():
    """The multi-hot oracle (sum pooling, not a reference input): one-member bags equal embed_block_smode; a bag
    with repeats, an empty bag and an out-of-range member are checked by hand; gradients flow to every member."""
    rng = np.random.default_rng(0)
    tabs = [torch.as_tensor(rng.standard_normal((v, 4))) for v in (3, 5)]
    X = np.array([[0, 4], [2, 1], [1, 3]])
    one = O.BagIds(X.reshape(-1), np.arange(7), 2)
    assert torch.equal(O.embed_block_smode(tabs, one), O.embed_block_smode(tabs, torch.as_tensor(X)))
    # B = 2, F = 2: (0,0) = [0, 2, 2], (0,1) = [], (1,0) = [1, 7 (out of range)], (1,1) = [4]
    bags = O.BagIds([0, 2, 2, 1, 7, 4], [0, 3, 3, 5, 6], 2)
    W = [t.clone().requires_grad_(True) for t in tabs]
    Z = O.embed_block_smode(W, bags)
    assert torch.allclose(Z[0, 0], tabs[0][0] + 2 * tabs[0][2]) and torch.all(Z[0, 1] == 0)
    assert torch.equal(Z[1, 0], tabs[0][1]) and torch.equal(Z[1, 1], tabs[1][4])
    Z.sum().backward()
    assert torch.equal(W[0].grad[:, 0], torch.tensor([1.0, 1.0, 2.0], dtype=torch.float64))
    assert torch.equal(W[1].grad[:, 0], torch.tensor([0.0, 0.0, 0.0, 0.0, 1.0], dtype=torch.float64))

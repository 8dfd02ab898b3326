"""Multi-hot fields (gather and pool): ctr_embed_bag_fwd / ctr_emb_bag_sort through the C ABI, and the model path
that feeds them.  The reference's models look up one id per field (DeepFM/DeepFM.py:43-53); a bag with one member
must therefore reproduce the single-id kernels bit for bit, and a general bag is checked against a numpy bag sum
in the kernel's (member) order - bit-exact - and against the float64 oracle (oracle.embed_bags_smode) for
gradients and optimizer steps, at 1e-5 like every other fp32 check."""
import numpy as np
import pytest
import torch

from oracle import ctr_oracle as O
from tests.helpers import assert_close, no_dropout, sync_oracle_from_gpu
from tests.test_gpu_models import _build

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TOL = 1e-5


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def _tables(rng, cards, E):
    offs = np.zeros(len(cards) + 1, dtype=np.int64)
    offs[1:] = np.cumsum(cards)
    return rng.standard_normal((int(offs[-1]), E)).astype(np.float32), offs


def _random_bags(rng, B, cards, max_members=9, p_empty=0.15):
    """Ragged CSR: empty bags, single members, repeats inside a bag, a few long bags and a few out-of-range ids."""
    F = len(cards)
    counts = rng.integers(1, max_members + 1, B * F)
    counts[rng.random(B * F) < p_empty] = 0
    counts[rng.integers(0, B * F, 3)] = 37                       # longer than any unrolled step
    off = np.zeros(B * F + 1, dtype=np.int64)
    off[1:] = np.cumsum(counts)
    ids = np.zeros(int(off[-1]), dtype=np.int32)
    for i in range(B * F):
        ids[off[i]:off[i + 1]] = rng.integers(0, cards[i % F], counts[i])
    bad = rng.integers(0, ids.size, 5)
    ids[bad[0]], ids[bad[1]], ids[bad[2]] = -1, 2 ** 31 - 1, cards[0] + 100000
    return ids, off


def _bag_sum_fp32(W, offs, cards, ids, off, B, F, E):
    """Member-order fp32 sum, the kernel's order; returns Z[B,F,E] and the out-of-range member count."""
    Z = np.zeros((B, F, E), dtype=np.float32)
    oor = 0
    for i in range(B * F):
        f = i % F
        acc = np.zeros(E, dtype=np.float32)
        for x in ids[off[i]:off[i + 1]]:
            if 0 <= x < cards[f]:
                acc = acc + W[offs[f] + x]
            else:
                oor += 1
        Z[i // F, f] = acc
    return Z, oor


@pytest.mark.parametrize('E', [4, 16, 64, 128])
def test_bag_fwd_bit_exact_pool_and_fm(sess, E):
    from recommendation_amd import ops
    rng = np.random.default_rng(100 + E)
    cards = [3, 50, 7, 1000, 2]
    F, C, B = len(cards), 3, 61
    W, offs = _tables(rng, cards, E)
    ids, off = _random_bags(rng, B, cards)
    xc = rng.standard_normal((B, C)).astype(np.float32)
    Wc = rng.standard_normal((C, E)).astype(np.float32)
    N = F + C
    ld = (N * E + C + 3) // 4 * 4
    z = torch.full((B, ld), 7.0, device=DEV)
    first, s, second = torch.zeros(B, N, device=DEV), torch.zeros(B, E, device=DEV), torch.zeros(B, E, device=DEV)
    oor = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.embed_bag_fwd(dev(W), dev(offs), dev(ids), dev(off, torch.int32), dev(xc), dev(Wc), B=B, F=F, E=E, z_out=z,
                      cont_tail=True, first_out=first, s_out=s, second_out=second, oor_count=oor)
    Zb, n_oor = _bag_sum_fp32(W, offs, cards, ids, off, B, F, E)
    Z = np.concatenate([Zb, Wc[None] * xc[:, :, None]], 1)
    zz = z.cpu().numpy()
    assert np.array_equal(zz[:, :N * E].reshape(B, N, E), Z)
    assert np.array_equal(zz[:, N * E:N * E + C], xc)
    assert int(oor.item()) == n_oor >= 1
    Z64 = Z.astype(np.float64)
    assert_close(first.cpu().numpy(), Z64.sum(2), TOL, 'first')
    assert_close(s.cpu().numpy(), Z64.sum(1), TOL, 's')
    assert_close(second.cpu().numpy(), 0.5 * (Z64.sum(1) ** 2 - (Z64 ** 2).sum(1)), TOL, 'second')


def test_one_member_bags_equal_the_single_id_lookup_bit_for_bit(sess):
    from recommendation_amd import ops
    rng = np.random.default_rng(5)
    cards, E, C, B = [11, 400, 5, 90, 3000, 17], 16, 4, 300
    F = len(cards)
    W, offs = _tables(rng, cards, E)
    idx = np.stack([rng.integers(0, v, B) for v in cards], 1).astype(np.int32)
    idx[3, 1] = -7
    xc = rng.standard_normal((B, C)).astype(np.float32)
    Wc = rng.standard_normal((C, E)).astype(np.float32)
    N = F + C
    d1 = ops.make_dropout(0.5, 99, 1)
    d2 = ops.make_dropout(0.2, 99, 2)
    outs = []
    for bags in (False, True):
        z = torch.zeros(B, N * E, device=DEV)
        first, s, second = torch.zeros(B, N, device=DEV), torch.zeros(B, E, device=DEV), torch.zeros(B, E, device=DEV)
        if bags:
            ops.embed_bag_fwd(dev(W), dev(offs), dev(idx.reshape(-1)), dev(np.arange(B * F + 1), torch.int32), dev(xc),
                              dev(Wc), B=B, F=F, E=E, z_out=z, first_out=first, s_out=s, second_out=second,
                              drop_first=d1, drop_second=d2)
        else:
            ops.embed_fwd(dev(W), dev(offs), dev(idx), dev(xc), dev(Wc), E=E, z_out=z, first_out=first, s_out=s,
                          second_out=second, drop_first=d1, drop_second=d2)
        outs.append([t.cpu().numpy() for t in (z, first, s, second)])
    for a, b, name in zip(outs[0], outs[1], ('Z', 'first', 's', 'second')):
        assert np.array_equal(a, b), name


def test_bag_sort_and_segment_sum_give_the_pooling_gradient(sess):
    """Pairs: one per member, stable by table row.  The summed gradient of row r is the sum of dZ over every
    membership of r (a row that sits twice in a bag gets the bag's dZ twice)."""
    from recommendation_amd import ops
    rng = np.random.default_rng(8)
    cards, E, B = [4, 60, 9, 500], 8, 97
    F = N = len(cards)
    W, offs = _tables(rng, cards, E)
    total = int(offs[-1])
    ids, off = _random_bags(rng, B, cards)
    nnz = ids.size
    keys = torch.zeros(nnz, dtype=torch.int32, device=DEV)
    src = torch.zeros(nnz, dtype=torch.int32, device=DEV)
    ops.emb_bag_sort(dev(ids), dev(off, torch.int32), dev(offs), B=B, F=F, N=N, total_rows=total, keys_out=keys,
                     src_out=src, ws=sess.ws)
    bag = np.repeat(np.arange(B * F), np.diff(off))
    f = bag % F
    ok = (ids >= 0) & (ids < np.asarray(cards)[f])
    k_ref = np.where(ok, offs[f] + np.where(ok, ids, 0), total).astype(np.uint32)
    s_ref = ((bag // F) * N + f).astype(np.uint32)
    order = np.argsort(k_ref, kind='stable')
    assert np.array_equal(keys.cpu().numpy().view(np.uint32), k_ref[order])
    assert np.array_equal(src.cpu().numpy().view(np.uint32), s_ref[order])
    dz = rng.standard_normal((B, N * E)).astype(np.float32)
    g = torch.zeros(total, E, device=DEV)
    ops.emb_sparse_apply(keys, src, nnz, N=N, E=E, total_rows=total, dz=dev(dz), opt=ops.make_opt(0), slot0=g,
                         ws=sess.ws)
    g_ref = np.zeros((total, E))
    np.add.at(g_ref, k_ref[ok], dz.reshape(B * N, E).astype(np.float64)[s_ref[ok]])
    assert_close(g.cpu().numpy(), g_ref, TOL, 'summed table gradient')
    g2 = torch.zeros(total, E, device=DEV)
    ops.emb_sparse_apply(keys, src, nnz, N=N, E=E, total_rows=total, dz=dev(dz), opt=ops.make_opt(0), slot0=g2,
                         ws=sess.ws)
    assert torch.equal(g, g2)                                      # fixed summation order


def test_bag_entry_points_reject_malformed_arguments(sess):
    from recommendation_amd import ops
    W, offs = _tables(np.random.default_rng(0), [5, 6], 8)
    ids = dev(np.zeros(4, dtype=np.int32))
    z = torch.zeros(2, 16, device=DEV)
    with pytest.raises(ValueError):
        ops.embed_bag_fwd(dev(W), dev(offs), ids, dev(np.arange(4), torch.int32), B=2, F=2, E=8, z_out=z)  # B*F+1 = 5
    with pytest.raises(TypeError):
        ops.embed_bag_fwd(dev(W), dev(offs), ids.to(torch.int64), dev(np.arange(5), torch.int32), B=2, F=2, E=8, z_out=z)
    from recommendation_amd.engine import Bags
    with pytest.raises(ValueError):
        Bags.from_host([1, 2, 3], [0, 2, 1, 3, 3], 2, DEV)           # not monotone
    with pytest.raises(ValueError):
        Bags.from_host([1, 2, 3], [0, 1, 2, 2, 2], 2, DEV)           # does not end at len(ids)


def _model_case(name, B, seed):
    cfg = no_dropout(O.default_cfg(name, embed_size=16, mode='S'))
    rng = np.random.default_rng(seed)
    cfg.cate_list = [int(v) for v in rng.integers(3, 200, 7)]
    cfg.cate_num, cfg.cont_num = 7, 4
    return cfg, rng


def _bag_batch(rng, cfg, B):
    ids, off = _random_bags(rng, B, cfg.cate_list, max_members=5)
    # a quarter positives: with balanced labels sum_b (y_out - y) nearly cancels and the bias gradients lose digits
    return (ids, off), rng.standard_normal((B, cfg.cont_num)), (rng.random(B) < 0.25).astype(np.int64)


@pytest.mark.parametrize('name', ['DeepFM', 'DCN', 'NFM', 'PNN'])
def test_models_train_on_bags_like_the_oracle(sess, name):
    """Forward, loss, every gradient (dense Adam path exposes the table gradient) against the float64 oracle with
    the same bag input; out-of-range members are counted, so the strict id check is off here."""
    cfg, rng = _model_case(name, 64, 21)
    cfg.adam_mode, cfg.strict_ids = 'tf1', False
    P = O.init_params(name, cfg, seed=6, std=0.05)
    Xb, Xf, y = _bag_batch(rng, cfg, 64)
    orc = O.OracleModel(name, cfg, P, dtype=torch.float64)
    m = _build(sess, name, cfg, P)
    assert_close(m.predict_proba(sess, Xb, Xf), orc.forward(Xb, Xf).y_out.numpy(), TOL, 'y_out')
    loss_ref, g_ref, _ = orc.loss_and_grads(Xb, Xf, y)
    loss_gpu = m.run(sess, Xb, Xf, y, True)
    assert abs(float(loss_gpu) - float(loss_ref)) < TOL * max(1.0, abs(float(loss_ref)))
    for k, v in m.get_grads().items():
        assert_close(v, g_ref[k].numpy(), TOL, k)
    assert m.out_of_range_lookups() > 0


def test_lazy_adam_steps_on_changing_bags_match_the_oracle(sess):
    cfg, rng = _model_case('DeepFM', 128, 33)
    cfg.adam_mode, cfg.strict_ids = 'lazy', False
    P = O.init_params('DeepFM', cfg, seed=9, std=0.05)
    orc = O.OracleModel('DeepFM', cfg, P, dtype=torch.float64)
    m = _build(sess, 'DeepFM', cfg, P)
    for step in range(4):
        B = 128 if step != 2 else 50                               # a different batch size and member count
        Xb, Xf, y = _bag_batch(rng, cfg, B)
        sync_oracle_from_gpu(orc, m)
        l_o = orc.run(Xb, Xf, y, True)
        l_g = float(m.run(sess, Xb, Xf, y, True))
        assert abs(float(l_o) - l_g) < TOL * max(1.0, abs(float(l_o)))
        for k, v in m.get_params().items():
            d = float(np.max(np.abs(v - orc.P[k].numpy())))
            assert d < 2e-6, '%s: |gpu - oracle| = %.3g after step %d' % (k, d, step)
    assert m.global_step == 4


def test_strict_ids_raise_on_an_out_of_range_bag_member(sess):
    cfg, rng = _model_case('DeepFM', 16, 44)
    cfg.strict_ids = True
    P = O.init_params('DeepFM', cfg, seed=1, std=0.05)
    m = _build(sess, 'DeepFM', cfg, P)
    Xb, Xf, y = _bag_batch(rng, cfg, 16)
    with pytest.raises(IndexError):
        m.run(sess, Xb, Xf, y, True)

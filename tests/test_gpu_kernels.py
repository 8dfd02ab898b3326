"""Kernel-level parity through the C ABI: every libctrhip operator against numpy/torch-CPU
restatements on seeded inputs.  Index handling and row copies must be bit-exact; fp32
reductions within 1e-5 relative (max-norm) of a float64 reference."""
import numpy as np
import pytest
import torch

from oracle import ctr_oracle as O
from tests.helpers import assert_close, assert_sorted_pairs, rel_err

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


@pytest.fixture(scope='module')
def ops_ws(sess):
    from recommendation_amd import ops
    return ops, sess.ws


@pytest.mark.parametrize('n,bits', [(1, 3), (63, 8), (64, 9), (4097, 17), (200000, 26), (70001, 32), (300000, 2),
                                    (5000, 2), (8192, 26), (8193, 26), (777, 32),
                                    (50000, 19), (100000, 20), (4097, 10), (30000, 30), (6000, 28),   # 10-bit digits
                                    (4300000, 27), (65537, 18), (4194304, 9)])   # several thousand tiles: the scan walks its tile ranges in a loop
def test_sort_pairs_stable_exact(ops_ws, n, bits):
    ops, ws = ops_ws
    rng = np.random.default_rng(n + bits)
    keys = rng.integers(0, 2 ** bits, n, dtype=np.uint64).astype(np.uint32)
    if bits == 32:
        keys[::50] = 0xFFFFFFFF      # the largest key, several times
    vals = np.arange(n, dtype=np.uint32)
    ko, vo = ops.sort_pairs(dev(keys.view(np.int32)), dev(vals.view(np.int32)), bits, ws)
    order = np.argsort(keys, kind='stable')
    assert np.array_equal(ko.cpu().numpy().view(np.uint32), keys[order])
    assert np.array_equal(vo.cpu().numpy().view(np.uint32), vals[order])


def _tables(rng, cards, E):
    offs = np.zeros(len(cards) + 1, dtype=np.int64)
    offs[1:] = np.cumsum(cards)
    W = rng.standard_normal((int(offs[-1]), E)).astype(np.float32)
    return W, offs


@pytest.mark.parametrize('E', [4, 8, 16, 64, 128])
def test_embed_fwd_smode_bit_exact_rows_and_fm(ops_ws, E):
    ops, ws = ops_ws
    rng = np.random.default_rng(E)
    cards = [3, 50, 7, 1000, 2, 33]
    F, C, B = len(cards), 5, 257
    W, offs = _tables(rng, cards, E)
    idx = np.stack([rng.integers(0, v, B) for v in cards], 1).astype(np.int32)
    idx[0, 0] = -1
    idx[1, 3] = 1000          # == V: out of range
    idx[2, 5] = 2 ** 31 - 1
    xc = rng.standard_normal((B, C)).astype(np.float32)
    Wc = rng.standard_normal((C, E)).astype(np.float32)
    N = F + C
    ld = (N * E + C + 3) // 4 * 4
    z = torch.full((B, ld), 7.0, device=DEV)
    first = torch.zeros(B, N, device=DEV)
    s = torch.zeros(B, E, device=DEV)
    second = torch.zeros(B, E + 3, device=DEV)
    oor = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.embed_fwd(dev(W), dev(offs), dev(idx), dev(xc), dev(Wc), E=E, z_out=z, cont_tail=True, first_out=first,
                  s_out=s, second_out=second[:, 3:], oor_count=oor)
    # reference
    Z = np.zeros((B, N, E), dtype=np.float32)
    for f in range(F):
        ok = (idx[:, f] >= 0) & (idx[:, f] < cards[f])
        Z[ok, f] = W[offs[f] + idx[ok, f]]
    Z[:, F:] = Wc[None] * xc[:, :, None]
    zz = z.cpu().numpy()
    assert np.array_equal(zz[:, :N * E].reshape(B, N, E), Z)          # copies / one multiply: bit-exact
    assert np.array_equal(zz[:, N * E:N * E + C], xc)
    assert int(oor.item()) == 3
    Z64 = Z.astype(np.float64)
    assert rel_err(first.cpu().numpy(), Z64.sum(2)) < 1e-5
    assert rel_err(s.cpu().numpy(), Z64.sum(1)) < 1e-5
    assert rel_err(second[:, 3:].cpu().numpy(), 0.5 * (Z64.sum(1) ** 2 - (Z64 ** 2).sum(1))) < 1e-5


def test_embed_onehot_rmode_bit_exact(ops_ws):
    ops, ws = ops_ws
    rng = np.random.default_rng(1)
    cards, E, B, C = [3, 5, 4, 9], 8, 33, 2
    W, offs = _tables(rng, cards, E)
    idx = np.stack([rng.integers(0, v, B) for v in cards], 1).astype(np.int32)
    idx[0, 0], idx[1, 1] = 3, -2
    xc = rng.standard_normal((B, C)).astype(np.float32)
    SV = int(offs[-1])
    ld = (SV * E + C + 3) // 4 * 4
    z = torch.full((B, ld), 3.0, device=DEV)
    pos = torch.zeros(B, len(cards), dtype=torch.int32, device=DEV)
    ops.embed_onehot_fwd(dev(W), dev(offs), dev(idx), dev(xc), E=E, z_out=z, cont_tail=True, pos_out=pos)
    tabs = [torch.as_tensor(W[offs[f]:offs[f + 1]]) for f in range(len(cards))]
    Zref = O.embed_block_rmode_literal(tabs, torch.as_tensor(idx.astype(np.int64))).numpy()
    zz = z.cpu().numpy()
    assert np.array_equal(zz[:, :SV * E].reshape(B, SV, E), Zref)
    assert np.array_equal(zz[:, SV * E:SV * E + C], xc)
    pref = np.where((idx >= 0) & (idx < np.array(cards)[None]), offs[:-1][None] + idx, -1)
    assert np.array_equal(pos.cpu().numpy(), pref)


@pytest.mark.parametrize('rmode', [False, True])
@pytest.mark.parametrize('B', [1, 37, 128, 1023, 1024, 1638, 1639, 4096, 4097, 9001])
def test_emb_sort_pairs_small_and_large_batches(ops_ws, B, rmode):
    """(key, src) pairs of the table backward in stable key order, from one pair to several radix tiles, in both
    table semantics; out-of-range ids get the sentinel key.  From 1024 samples on the sort runs field by field (one
    tile per field, then two and three with a ragged last one)."""
    ops, ws = ops_ws
    rng = np.random.default_rng(B)
    cards = [3, 50, 7, 1000, 20]
    F = len(cards)
    offs = np.zeros(F + 1, dtype=np.int64)
    offs[1:] = np.cumsum(cards)
    total = int(offs[-1])
    N = total if rmode else F
    idx = np.stack([rng.integers(0, v, B) for v in cards], 1).astype(np.int32)
    idx[0, 1] = -1
    idx[B // 2, 3] = cards[3]
    ok = (idx >= 0) & (idx < np.array(cards)[None])
    b = np.arange(B, dtype=np.int64)[:, None]
    f = np.arange(F, dtype=np.int64)[None]
    if rmode:
        k = np.where(ok, offs[:-1][None] + 1, total)
        v = np.where(ok, b * N + offs[:-1][None] + idx, 0)
    else:
        k = np.where(ok, offs[:-1][None] + idx, total)
        v = np.broadcast_to(b * N + f, (B, F))
    k, v = k.reshape(-1).astype(np.uint32), np.ascontiguousarray(v).reshape(-1).astype(np.uint32)
    keys = torch.zeros(B * F, dtype=torch.int32, device=DEV)
    src = torch.zeros(B * F, dtype=torch.int32, device=DEV)
    ops.emb_sort(dev(idx), dev(offs), N=N, total_rows=total, rmode=rmode, keys_out=keys, src_out=src, ws=ws)
    assert_sorted_pairs(keys.cpu().numpy().view(np.uint32), src.cpu().numpy().view(np.uint32), k, v, total)


@pytest.mark.parametrize('skew', [False, True])
@pytest.mark.parametrize('B,cards', [(5000, [3, 511, 512, 513, 262143, 262144, 262145, 1, 40, 5000000]),
                                     (20000, [2, 300, 70000, 9000000, 17, 140000]),
                                     (2000, list(range(1, 40))), (131072, [70000, 5]), (131073, [600, 3])])
def test_emb_sort_by_fields_pass_counts_and_tile_boundaries(ops_ws, B, cards, skew):
    """The field-segmented sort: tables whose ids need one, two and three passes side by side (sizes around 2^9 and
    2^18, where the pass count changes; a one-row table; a ten-million-row table), several tiles per field with a
    ragged last one, more fields than one builder block takes, uniform and heavy-tailed ids (long runs of one id
    across tiles), out-of-range ids in several fields; the largest batch that sorts by fields (32 tiles a field) and
    the first that takes the global sort.  Run twice: the same bits (the counters are integer atomics)."""
    ops, ws = ops_ws
    rng = np.random.default_rng(B + len(cards) + skew)
    F = len(cards)
    offs = np.zeros(F + 1, dtype=np.int64)
    offs[1:] = np.cumsum(cards)
    total = int(offs[-1])
    if skew:
        idx = np.stack([np.minimum(rng.pareto(0.7, B).astype(np.int64), v - 1) for v in cards], 1)
    else:
        idx = np.stack([rng.integers(0, v, B) for v in cards], 1)
    idx = idx.astype(np.int32)
    for f in (0, F // 2, F - 1):
        idx[rng.integers(0, B, 3), f] = [-1, cards[f], 2 ** 31 - 1]
    ok = (idx >= 0) & (idx < np.array(cards)[None])
    k = np.where(ok, offs[:-1][None] + idx, total).reshape(-1)
    v = (np.arange(B, dtype=np.int64)[:, None] * F + np.arange(F, dtype=np.int64)[None]).reshape(-1)
    outs = []
    for _ in range(2):
        keys = torch.full((B * F,), -1, dtype=torch.int32, device=DEV)
        src = torch.full((B * F,), -1, dtype=torch.int32, device=DEV)
        ops.emb_sort(dev(idx), dev(offs), N=F, total_rows=total, rmode=False, keys_out=keys, src_out=src, ws=ws)
        outs.append((keys.cpu().numpy().view(np.uint32), src.cpu().numpy().view(np.uint32)))
    assert_sorted_pairs(outs[0][0], outs[0][1], k, v, total)
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize('B,F', [(9, 600), (5, 1100), (3000, 1), (700, 13)])
def test_emb_sort_field_counts_around_the_builder_block(ops_ws, B, F):
    """The pair builder walks (sample, field) by fixed steps of its block size and keeps the field offsets in LDS up to
    1023 fields: more fields than threads of a block, more than the LDS table holds, and a single field."""
    ops, ws = ops_ws
    rng = np.random.default_rng(B * F)
    cards = rng.integers(2, 40, F)
    offs = np.zeros(F + 1, dtype=np.int64)
    offs[1:] = np.cumsum(cards)
    total = int(offs[-1])
    idx = np.stack([rng.integers(0, v, B) for v in cards], 1).astype(np.int32)
    idx[B // 2, F // 2] = -3
    ok = idx >= 0
    k = np.where(ok, offs[:-1][None] + idx, total).reshape(-1).astype(np.uint32)
    v = (np.arange(B, dtype=np.int64)[:, None] * F + np.arange(F, dtype=np.int64)[None]).reshape(-1).astype(np.uint32)
    order = np.argsort(k, kind='stable')
    keys = torch.zeros(B * F, dtype=torch.int32, device=DEV)
    src = torch.zeros(B * F, dtype=torch.int32, device=DEV)
    ops.emb_sort(dev(idx), dev(offs), N=F, total_rows=total, rmode=False, keys_out=keys, src_out=src, ws=ws)
    assert np.array_equal(keys.cpu().numpy().view(np.uint32), k[order])
    assert np.array_equal(src.cpu().numpy().view(np.uint32), v[order])


@pytest.mark.parametrize('skew', [False, True])
@pytest.mark.parametrize('E', [4, 16, 64])
def test_sort_segment_reduce_matches_scatter_add(ops_ws, E, skew):
    """emb_sort + sparse_apply(GRAD_ONLY) == np.add.at, including hot rows with long runs."""
    ops, ws = ops_ws
    rng = np.random.default_rng(E + skew)
    cards = [2, 3, 4000, 17, 100000]
    F, B = len(cards), 6000
    offs = np.zeros(F + 1, dtype=np.int64)
    offs[1:] = np.cumsum(cards)
    total = int(offs[-1])
    if skew:
        idx = np.stack([np.minimum((rng.pareto(0.6, B)).astype(np.int64), v - 1) for v in cards], 1)
    else:
        idx = np.stack([rng.integers(0, v, B) for v in cards], 1)
    idx = idx.astype(np.int32)
    idx[5, 2] = -1
    idx[6, 4] = cards[4]
    ld = F * E + 4
    dz = rng.standard_normal((B, ld)).astype(np.float32)
    keys = torch.zeros(B * F, dtype=torch.int32, device=DEV)
    src = torch.zeros(B * F, dtype=torch.int32, device=DEV)
    d_idx, d_offs, d_dz = dev(idx), dev(offs), dev(dz)
    outs = []
    for _ in range(2):
        grad = torch.zeros(total, E, device=DEV)
        ops.emb_sort(d_idx, d_offs, N=F, total_rows=total, rmode=False, keys_out=keys, src_out=src, ws=ws)
        ops.emb_sparse_apply(keys, src, B * F, N=F, E=E, total_rows=total, dz=d_dz,
                             opt=ops.make_opt(0), slot0=grad, ws=ws)
        outs.append(grad.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])                       # bitwise reproducible
    ref = np.zeros((total, E), dtype=np.float64)
    g = dz[:, :F * E].reshape(B, F, E).astype(np.float64)
    for f in range(F):
        ok = (idx[:, f] >= 0) & (idx[:, f] < cards[f])
        np.add.at(ref, offs[f] + idx[ok, f], g[ok, f])
    assert rel_err(outs[0], ref) < 1e-5
    # sorted keys really are sorted and complete
    k = keys.cpu().numpy().view(np.uint32)
    assert np.all(np.diff(k.astype(np.int64)) >= 0)
    assert int((k == total).sum()) == 2


@pytest.mark.parametrize('E', [8, 64])
@pytest.mark.parametrize('kind', ['adam', 'adagrad', 'sgd'])
def test_sparse_optimizers_match_tf1_formulas(ops_ws, kind, E):
    """E = 64: sixteen lanes a row, sixteen pairs a wave (the 50-row table's runs cross many of those blocks)."""
    ops, ws = ops_ws
    rng = np.random.default_rng(3)
    cards, B = [50, 7], 300
    F = 2
    offs = np.array([0, 50, 57], dtype=np.int64)
    total = 57
    idx = np.stack([rng.integers(0, v, B) for v in cards], 1).astype(np.int32)
    dz = rng.standard_normal((B, F * E)).astype(np.float32)
    W = rng.standard_normal((total, E)).astype(np.float32)
    m0 = np.abs(rng.standard_normal((total, E))).astype(np.float32) * 0.1
    v0 = np.abs(rng.standard_normal((total, E))).astype(np.float32) * 0.1 + 0.1
    g = np.zeros((total, E), dtype=np.float64)
    for f in range(F):
        np.add.at(g, offs[f] + idx[:, f], dz[:, f * E:(f + 1) * E].astype(np.float64))
    touched = np.zeros(total, dtype=bool)
    for f in range(F):
        touched[offs[f] + idx[:, f]] = True
    keys = torch.zeros(B * F, dtype=torch.int32, device=DEV)
    src = torch.zeros(B * F, dtype=torch.int32, device=DEV)
    ops.emb_sort(dev(idx), dev(offs), N=F, total_rows=total, rmode=False, keys_out=keys, src_out=src, ws=ws)
    dW, dm, dv = dev(W), dev(m0), dev(v0)
    lr, b1, b2, eps = 0.01, 0.9, 0.999, 1e-8
    code = dict(adam=1, adagrad=2, sgd=3)[kind]
    ops.emb_sparse_apply(keys, src, B * F, N=F, E=E, total_rows=total, dz=dev(dz),
                         opt=ops.make_opt(code, lr=lr, beta1=b1, beta2=b2, eps=eps), tables=dW,
                         slot0=dm if kind != 'sgd' else None, slot1=dv if kind == 'adam' else None, ws=ws)
    W64, m64, v64 = W.astype(np.float64), m0.astype(np.float64), v0.astype(np.float64)
    if kind == 'adam':
        m1 = b1 * m64 + (1 - b1) * g
        v1 = b2 * v64 + (1 - b2) * g * g
        Wn = W64 - lr * m1 / (np.sqrt(v1) + eps)
    elif kind == 'adagrad':
        m1 = m64 + g * g
        Wn = W64 - lr * g / np.sqrt(m1)
    else:
        Wn = W64 - lr * g
    out = dW.cpu().numpy()
    assert rel_err(out[touched], Wn[touched]) < 1e-5
    assert np.array_equal(out[~touched], W[~touched])             # untouched rows are not written


@pytest.mark.parametrize('E', [8, 16, 64])
def test_adam_pass_over_the_rows_without_a_gradient(ops_ws, E):
    """ctr_adam_untouched_rows: tf.train.AdamOptimizer's step on every row the sorted keys do not name - the bits of
    ctr_dense_update with a zero gradient.  Most rows have no optimizer state (m = v = +0): those are skipped by the
    kernel (their byte in the row map is 0: not even read) and must come out unchanged, as the formula leaves them;
    rows with state (a -0.0 in m included) decay and move; the rows the keys name are left alone, their bit 0 is
    cleared again and bit 1 records that they now have state."""
    ops, ws = ops_ws
    rng = np.random.default_rng(E)
    total = 70001
    W = rng.standard_normal((total, E)).astype(np.float32)
    m0 = np.zeros((total, E), dtype=np.float32)
    v0 = np.zeros((total, E), dtype=np.float32)
    state = rng.random(total) < 0.2
    state[60000:] = False                                   # a long stretch without state: whole waves skip
    m0[state] = (rng.standard_normal((int(state.sum()), E)) * 0.1).astype(np.float32)
    v0[state] = (np.abs(rng.standard_normal((int(state.sum()), E))) * 0.01).astype(np.float32)
    m0[5, 0], v0[5, :] = -0.0, 0.0                          # negative zero alone is state too
    m0[5, 1:] = 0.0
    m0[7, :], v0[7, :] = 0.0, 0.0
    v0[7, 3] = 1e-30                                        # one tiny v in an otherwise empty row
    named = np.sort(rng.choice(total, 3000, replace=False)).astype(np.int32)
    keys = dev(np.concatenate([np.repeat(named, 2), np.full(5, total, dtype=np.int32)]))   # runs, and ignored pairs
    opt = ops.make_opt(1, lr=0.01, beta1=0.9, beta2=0.999, eps=1e-8)
    dW, dm, dv = dev(W), dev(m0), dev(v0)
    has_state = state.copy()
    has_state[[5, 7]] = True
    has_state[100:200] = True                               # marked, yet without state: read and left alone
    touched = dev((has_state * 2).astype(np.uint8))
    ops.adam_untouched_rows(keys, keys.numel(), total_rows=total, E=E, opt=opt, tables=dW, slot0=dm, slot1=dv,
                            touched=touched)
    rW, rm, rv = dev(W), dev(m0), dev(v0)
    ops.dense_update(rW, torch.zeros_like(rW), opt, rm, rv)
    for got, ref, orig in ((dW, rW, W), (dm, rm, m0), (dv, rv, v0)):
        got, ref = got.cpu().numpy(), ref.cpu().numpy()
        ref[named] = orig[named]
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    want = has_state * 2
    want[named] = 2                                         # bit 0 cleared again, bit 1 left on the named rows
    assert np.array_equal(touched.cpu().numpy(), want.astype(np.uint8))


@pytest.mark.parametrize('E', [4, 16, 64])
def test_sparse_adam_with_m_and_v_side_by_side_leaves_the_same_bits(ops_ws, E):
    """ctr_emb_sparse_apply_ld: Adam's slots as the two column blocks of one [rows, 2E] buffer (row pitch 2E) against
    two separate [rows, E] arrays - every word of W, m and v identical after two steps, hot rows (runs across
    several 64-pair blocks: the tail pass) included."""
    ops, ws = ops_ws
    rng = np.random.default_rng(E)
    cards = [3, 900, 50000]
    F, B = len(cards), 5000
    offs = np.zeros(F + 1, dtype=np.int64)
    offs[1:] = np.cumsum(cards)
    total = int(offs[-1])
    idx = np.stack([np.minimum(rng.pareto(0.8, B).astype(np.int64), v - 1) for v in cards], 1).astype(np.int32)
    keys = torch.zeros(B * F, dtype=torch.int32, device=DEV)
    src = torch.zeros(B * F, dtype=torch.int32, device=DEV)
    ops.emb_sort(dev(idx), dev(offs), N=F, total_rows=total, rmode=False, keys_out=keys, src_out=src, ws=ws)
    W0 = rng.standard_normal((total, E)).astype(np.float32)
    Wa, Wb = dev(W0), dev(W0)
    ma, va = torch.zeros(total, E, device=DEV), torch.zeros(total, E, device=DEV)
    mv = torch.zeros(total, 2 * E, device=DEV)
    mb, vb = mv[:, :E], mv[:, E:]
    for step in range(2):
        dz = dev(rng.standard_normal((B, F * E)).astype(np.float32))
        opt = ops.make_opt(1, lr=0.01 * (step + 1), beta1=0.9, beta2=0.999, eps=1e-8)
        ops.emb_sparse_apply(keys, src, B * F, N=F, E=E, total_rows=total, dz=dz, opt=opt, tables=Wa, slot0=ma, slot1=va, ws=ws)
        ops.emb_sparse_apply(keys, src, B * F, N=F, E=E, total_rows=total, dz=dz, opt=opt, tables=Wb, slot0=mb, slot1=vb, ws=ws)
    assert torch.equal(Wa, Wb) and torch.equal(ma, mb) and torch.equal(va, vb)
    assert float(ma.abs().sum()) > 0
    with pytest.raises(ValueError, match='one row pitch'):
        ops.emb_sparse_apply(keys, src, B * F, N=F, E=E, total_rows=total, dz=dz, opt=opt, tables=Wb, slot0=mb, slot1=va, ws=ws)


@pytest.mark.parametrize('kind', ['grad_only', 'adam'])
@pytest.mark.parametrize('E', [4, 16, 32])
def test_sparse_apply_with_the_fm_row_term(ops_ws, E, kind):
    """ctr_emb_sparse_apply_fm: dz without -g2 * Z, the row update subtracting (sum of g2 over the row's lookups) * row -
    against float64 (gradient of a row = sum over its lookups of dz[b,f] - g2[b] * W[row]) and against the plain update
    fed the complete gradient; hot rows whose runs cross many 64-pair blocks (the tail pass), out-of-range ids, g2 as
    a column block at an odd column offset of a wider buffer (8-byte aligned rows)."""
    ops, ws = ops_ws
    rng = np.random.default_rng(E + len(kind))
    cards = [3, 700, 40000]
    F, B = len(cards), 7000
    offs = np.zeros(F + 1, dtype=np.int64)
    offs[1:] = np.cumsum(cards)
    total = int(offs[-1])
    idx = np.stack([np.minimum(rng.pareto(0.8, B).astype(np.int64), v - 1) for v in cards], 1).astype(np.int32)
    idx[11, 1] = -1
    idx[12, 2] = cards[2]
    ok = (idx >= 0) & (idx < np.array(cards)[None])
    W = rng.standard_normal((total, E)).astype(np.float32)
    dz = rng.standard_normal((B, F * E + 4)).astype(np.float32)
    wide = rng.standard_normal((B, E + 10)).astype(np.float32)
    g2 = wide[:, 6:6 + E]
    rows = offs[:-1][None] + idx
    Z = np.where(ok[:, :, None], W[np.where(ok, rows, 0)], 0.0).astype(np.float64)
    full = dz[:, :F * E].astype(np.float64).reshape(B, F, E) - g2.astype(np.float64)[:, None, :] * Z
    gref = np.zeros((total, E))
    for f in range(F):
        np.add.at(gref, rows[ok[:, f], f], full[ok[:, f], f])
    keys = torch.zeros(B * F, dtype=torch.int32, device=DEV)
    src = torch.zeros(B * F, dtype=torch.int32, device=DEV)
    ops.emb_sort(dev(idx), dev(offs), N=F, total_rows=total, rmode=False, keys_out=keys, src_out=src, ws=ws)
    d_wide = dev(wide)
    d_g2 = d_wide[:, 6:6 + E]
    assert ops.emb_sparse_apply_fm_supported(E, d_g2)
    d_full = dev(np.concatenate([full.reshape(B, F * E), np.zeros((B, 4))], 1).astype(np.float32))
    if kind == 'grad_only':
        outs = []
        for _ in range(2):
            grad = torch.zeros(total, E, device=DEV)
            ops.emb_sparse_apply(keys, src, B * F, N=F, E=E, total_rows=total, dz=dev(dz), opt=ops.make_opt(0),
                                 tables=dev(W), slot0=grad, ws=ws, fm_g2=d_g2)
            outs.append(grad.cpu().numpy())
        assert np.array_equal(outs[0], outs[1])
        assert rel_err(outs[0], gref) < 1e-5
        return
    opt = ops.make_opt(1, lr=0.01, beta1=0.9, beta2=0.999, eps=1e-8)
    Wa, Wb = dev(W), dev(W)
    mv_a, mv_b = torch.zeros(total, 2 * E, device=DEV), torch.zeros(total, 2 * E, device=DEV)
    ops.emb_sparse_apply(keys, src, B * F, N=F, E=E, total_rows=total, dz=dev(dz), opt=opt, tables=Wa,
                         slot0=mv_a[:, :E], slot1=mv_a[:, E:], ws=ws, fm_g2=d_g2)
    ops.emb_sparse_apply(keys, src, B * F, N=F, E=E, total_rows=total, dz=d_full, opt=opt, tables=Wb,
                         slot0=mv_b[:, :E], slot1=mv_b[:, E:], ws=ws)
    m_ref = 0.1 * gref
    touched = np.zeros(total, dtype=bool)
    for f in range(F):
        touched[rows[ok[:, f], f]] = True
    assert rel_err(mv_a[:, :E].cpu().numpy()[touched], m_ref[touched]) < 1e-5
    assert np.array_equal(Wa.cpu().numpy()[~touched], W[~touched])
    # against the plain update on the complete gradient: the same first moment up to the rounding of one product
    assert rel_err(mv_a[:, :E].cpu().numpy(), mv_b[:, :E].cpu().numpy()) < 1e-5


GEMM_SHAPES = [(1, 1, 1), (5, 3, 2), (130, 70, 33), (257, 129, 19), (64, 1, 300), (429, 256, 5000),
               (1000, 429, 256), (33, 250, 4096)]


@pytest.mark.parametrize('ta', [False, True])
@pytest.mark.parametrize('tb', [False, True])
@pytest.mark.parametrize('shape', GEMM_SHAPES)
def test_gemm_f32_all_layouts(ops_ws, shape, ta, tb):
    ops, ws = ops_ws
    M, N, K = shape
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    Bm = rng.standard_normal((K, N)).astype(np.float32)
    # odd leading dimensions on purpose (scalar load path) and aligned ones (vector path)
    pad = 0 if (M + N + K) % 2 else 3
    As = np.zeros(((K, M + pad) if ta else (M, K + pad)), dtype=np.float32)
    Bs = np.zeros(((N, K + pad) if tb else (K, N + pad)), dtype=np.float32)
    if ta:
        As[:, :M] = A.T
    else:
        As[:, :K] = A
    if tb:
        Bs[:, :K] = Bm.T
    else:
        Bs[:, :N] = Bm
    dA, dB = dev(As), dev(Bs)
    vA = dA[:, :M] if ta else dA[:, :K]
    vB = dB[:, :K] if tb else dB[:, :N]
    Cbuf = torch.full((M, N + 5), -1.0, device=DEV)
    ops.gemm(vA, vB, Cbuf[:, :N], trans_a=ta, trans_b=tb, ws=ws)
    ref = A.astype(np.float64) @ Bm.astype(np.float64)
    out = Cbuf.cpu().numpy()
    assert rel_err(out[:, :N], ref) < 1e-5
    assert np.all(out[:, N:] == -1.0)                              # nothing written outside the view


def _sweep_shapes():
    """Seeded shapes on both sides of the kernel-plan switches (tile widths, pre-transposed weights, split-K, the
    streaming weight-gradient kernel), plus ragged edges."""
    rng = np.random.default_rng(2024)
    out = []
    for _ in range(28):
        M = int(rng.choice([1, 31, 127, 128, 129, 700, 2047, 2048, 2500, 5000]))
        N = int(rng.choice([1, 17, 64, 191, 192, 193, 256, 511, 512, 640, 1100]))
        K = int(rng.choice([1, 15, 16, 100, 511, 512, 513, 1024, 3000, 20000]))
        if M * N * K > 6e9:
            K = 512
        out.append((M, N, K, bool(rng.integers(2)), bool(rng.integers(2)), int(rng.integers(4))))
    return out


@pytest.mark.parametrize('M,N,K,ta,tb,epi', _sweep_shapes())
def test_gemm_plan_boundaries(ops_ws, M, N, K, ta, tb, epi):
    """fp32 products against float64 on the GPU, with a bias+relu, masked or accumulating epilogue."""
    ops, ws = ops_ws
    gen = torch.Generator(device=DEV)
    gen.manual_seed(M * 31 + N * 7 + K)

    def mk(r, c):
        return torch.randn(r, (c + 3) // 4 * 4 + (4 if (r + c) % 3 == 0 else 0), generator=gen, device=DEV)[:, :c]
    A = mk(K, M) if ta else mk(M, K)
    Bm = mk(N, K) if tb else mk(K, N)
    ref = (A.double().t() if ta else A.double()) @ (Bm.double().t() if tb else Bm.double())
    C = torch.full((M, N + 3), 0.5, device=DEV)
    kw = {}
    if epi == 1:
        bias = torch.randn(N, generator=gen, device=DEV)
        kw = dict(bias=bias, relu=True)
        ref = torch.relu(ref + bias.double())
    elif epi == 2:
        h = torch.relu(mk(M, N))
        kw = dict(mask_src=h, mask_scale=1.25)
        ref = ref * torch.where(h > 0, 1.25, 0.0)
    elif epi == 3:
        kw = dict(accumulate=True)
        ref = ref + 0.5
    ops.gemm(A, Bm, C[:, :N], trans_a=ta, trans_b=tb, ws=ws, **kw)
    scale = float(((A.double().abs().t() if ta else A.double().abs()) @
                   (Bm.double().abs().t() if tb else Bm.double().abs())).max())
    assert float((C[:, :N].double() - ref).abs().max()) < 1e-5 * max(scale, 1e-30) + 1e-6
    assert bool((C[:, N:] == 0.5).all())


def test_gemm_epilogues(ops_ws):
    ops, ws = ops_ws
    rng = np.random.default_rng(5)
    M, N, K = 300, 70, 45
    A = rng.standard_normal((M, K)).astype(np.float32)
    Bm = rng.standard_normal((K, N)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    C0 = rng.standard_normal((M, N)).astype(np.float32)
    ref = A.astype(np.float64) @ Bm + bias
    # bias + relu
    C = torch.zeros(M, N, device=DEV)
    ops.gemm(dev(A), dev(Bm), C, bias=dev(bias), relu=True, ws=ws)
    assert rel_err(C.cpu().numpy(), np.maximum(ref, 0)) < 1e-5
    # bias + relu + dropout through the shared hash
    drop = ops.make_dropout(0.3, seed=99, stream_id=4)
    ops.gemm(dev(A), dev(Bm), C, bias=dev(bias), relu=True, dropout=drop, ws=ws)
    keep = O.keep_mask(99, 4, M * N, 0.7).reshape(M, N)
    assert rel_err(C.cpu().numpy(), np.maximum(ref, 0) * keep / 0.7) < 1e-5
    # accumulate
    C = dev(C0)
    ops.gemm(dev(A), dev(Bm), C, accumulate=True, ws=ws)
    assert rel_err(C.cpu().numpy(), C0 + A.astype(np.float64) @ Bm) < 1e-5
    # relu/dropout-backward mask
    act = rng.standard_normal((M, N)).astype(np.float32)
    C = torch.zeros(M, N, device=DEV)
    ops.gemm(dev(A), dev(Bm), C, mask_src=dev(act), mask_scale=1.25, ws=ws)
    assert rel_err(C.cpu().numpy(), np.where(act > 0, (A.astype(np.float64) @ Bm) * 1.25, 0)) < 1e-5


@pytest.mark.parametrize('M,K,N_rows,E,C', [
    (4096, 256, 26, 16, 13),    # 128 x 128 tiles (N = 429: what the DeepFM step runs)
    (2048, 64, 5, 8, 3),        # narrow product: the strip kernel's per-element epilogue
    (130, 4096, 6, 4, 2),       # few rows under a long K: split-K slabs, epilogue in the reduction
    (300, 45, 3, 12, 0),        # ragged tile edges, no continuous tail, E that does not divide 32
    (700, 300, 9, 8, 5),        # several column tiles at E = 8
])
def test_gemm_fm_backward_epilogue_matches_the_separate_pass(ops_ws, M, K, N_rows, E, C):
    """dx0 = g W^T with the FM terms of dZ added in the epilogue (ctr_fm_bwd_t) against the same product followed by
    ctr_embed_bwd_dz over it (gradient of DeepFM/DeepFM.py:55-66): the same bits, dropout on and off, and the
    columns of the continuous tail untouched by the FM terms."""
    ops, ws = ops_ws
    g = torch.Generator(device=DEV); g.manual_seed(M + K)
    rnd = lambda *sh: torch.randn(*sh, device=DEV, generator=g)
    D = N_rows * E + C
    ld = (D + 1 + 3) // 4 * 4
    gact, W = rnd(M, K), rnd(D, K) * 0.1
    x0 = rnd(M, ld)
    dcat = rnd(M, N_rows + E + 7)                    # g_first / g_second as unaligned column blocks of one buffer
    g1, g2, s = dcat[:, :N_rows], dcat[:, N_rows:N_rows + E], rnd(M, E)
    for rate in (0.0, 0.5):
        d1 = ops.make_dropout(rate, seed=7, stream_id=1)
        d2 = ops.make_dropout(rate, seed=7, stream_id=2)
        sep = torch.full((M, ld), 0.25, device=DEV)
        ops.gemm(gact, W, sep[:, :D], trans_b=True, ws=ws)
        plain = sep.clone()
        ops.embed_bwd_dz(sep, B=M, N=N_rows, E=E, gz=sep, g_first=g1, g_second=g2, s=s, z=x0, drop_first=d1,
                         drop_second=d2)
        # the epilogue takes the gradients with their keep masks already on them
        kept = dcat.clone()
        k1, k2 = kept[:, :N_rows], kept[:, N_rows:N_rows + E]
        ops.dropout_apply2(kept[:, :N_rows + E], N_rows, d1, E, d2)
        ref1, ref2 = g1.clone(), g2.clone()
        ops.dropout_apply(ref1, d1)
        ops.dropout_apply(ref2, d2)
        assert torch.equal(k1, ref1) and torch.equal(k2, ref2) and torch.equal(kept[:, N_rows + E:], dcat[:, N_rows + E:])
        fused = torch.full((M, ld), 0.25, device=DEV)
        ops.gemm(gact, W, fused[:, :D], trans_b=True, ws=ws, fm_bwd=dict(N=N_rows, E=E, g_first=k1, g_second=k2, s=s, z=x0))
        assert torch.equal(fused, sep)
        assert torch.equal(fused[:, N_rows * E:], plain[:, N_rows * E:])
        assert not torch.equal(fused[:, :N_rows * E], plain[:, :N_rows * E])
        # each term alone
        for kw, kwf in ((dict(g_first=g1), dict(g_first=k1)), (dict(g_second=g2, s=s, z=x0), dict(g_second=k2, s=s, z=x0))):
            sep = plain.clone()
            ops.embed_bwd_dz(sep, B=M, N=N_rows, E=E, gz=sep, drop_first=d1, drop_second=d2, **kw)
            fused = torch.full((M, ld), 0.25, device=DEV)
            ops.gemm(gact, W, fused[:, :D], trans_b=True, ws=ws, fm_bwd=dict(N=N_rows, E=E, **kwf))
            assert torch.equal(fused, sep)
    # float64 statement of the whole expression, dropout off
    ref = gact.double() @ W.double().t()
    zz = x0[:, :N_rows * E].double().view(M, N_rows, E)
    ref[:, :N_rows * E] += (g1.double()[:, :, None] + g2.double()[:, None, :] * (s.double()[:, None, :] - zz)).reshape(M, -1)
    out = torch.zeros(M, ld, device=DEV)
    ops.gemm(gact, W, out[:, :D], trans_b=True, ws=ws, fm_bwd=dict(N=N_rows, E=E, g_first=g1, g_second=g2, s=s, z=x0))
    assert_close(out[:, :D].cpu().numpy(), ref.cpu().numpy(), 1e-5, "dx0 + FM terms")


def test_colsum_rowdot_loss(ops_ws):
    ops, ws = ops_ws
    rng = np.random.default_rng(6)
    B, K, E = 5000, 37, 4
    X = rng.standard_normal((B, K + 3)).astype(np.float32)
    w = rng.standard_normal((B, 10)).astype(np.float32)
    out = torch.zeros(K, device=DEV)
    ops.colsum(dev(X)[:, :K], out, ws=ws)
    assert rel_err(out.cpu().numpy(), X[:, :K].astype(np.float64).sum(0)) < 1e-5
    ops.colsum(dev(X)[:, :K], out, w=dev(w), w_group=E, ws=ws)
    ref = (X[:, :K].astype(np.float64) * np.repeat(w, E, axis=1)[:, :K]).sum(0)
    assert rel_err(out.cpu().numpy(), ref) < 1e-5
    wv = rng.standard_normal(K).astype(np.float32)
    bias = np.array([0.3], dtype=np.float32)
    o = torch.zeros(B, device=DEV)
    ops.rowdot_fwd(dev(X)[:, :K], dev(wv), dev(bias), o)
    assert rel_err(o.cpu().numpy(), X[:, :K].astype(np.float64) @ wv + 0.3) < 1e-5
    # loss heads vs torch (float64) incl. gradients
    x = rng.standard_normal(B).astype(np.float32)
    y = rng.integers(0, 2, B).astype(np.float32)
    for kind in range(5):
        xx = x if kind not in (1, 4) else (1 / (1 + np.exp(-x))).astype(np.float32)   # probabilities where needed
        t = torch.tensor(xx, dtype=torch.float64, requires_grad=True)
        yy = torch.tensor(y, dtype=torch.float64)
        if kind == 0:
            p = torch.sigmoid(t); loss = O.log_loss(yy, p)
        elif kind == 1:
            p = t; loss = O.log_loss(yy, p)
        elif kind == 2:
            p = torch.sigmoid(t); loss = O.sigmoid_ce_with_logits(yy, p).mean()
        elif kind == 3:
            p = torch.sigmoid(t); loss = O.sigmoid_ce_with_logits(yy, t).mean()
        else:
            p = t; loss = O.sigmoid_ce_with_logits(yy, t).mean()
        loss.backward()
        yo, lo, dx = torch.zeros(B, device=DEV), torch.zeros(1, device=DEV), torch.zeros(B, device=DEV)
        ops.loss_head(kind, dev(xx), dev(y), y_out=yo, loss_out=lo, dx=dx, ws=ws)
        assert rel_err(yo.cpu().numpy(), p.detach().numpy()) < 1e-5, kind
        assert abs(float(lo.item()) - float(loss.item())) < 1e-5, kind
        assert_close(dx.cpu().numpy(), t.grad.numpy(), 1e-5, kind)


@pytest.mark.parametrize('D,L,B', [(50, 3, 37), (429, 3, 37), (1677, 6, 37), (300, 1, 37), (2100, 8, 37),
                                   (1677, 6, 8192), (429, 3, 65536), (50, 2, 200)])
def test_cross_network_fwd_bwd(ops_ws, D, L, B):
    """B = 37: one chunk of samples; 200, 8192 and 65536: several chunks, the last two at the chunk-count cap and
    at the benchmark shapes, so the partial sums of dw/db go through the staged final reduction."""
    ops, ws = ops_ws
    rng = np.random.default_rng(D + L)
    x0 = (rng.standard_normal((B, D)) * 0.5).astype(np.float32)
    w = (rng.standard_normal((L, D)) * 0.05).astype(np.float32)
    b = (rng.standard_normal((L, D)) * 0.05).astype(np.float32)
    g = rng.standard_normal((B, D)).astype(np.float32)
    ld = (D + 3) // 4 * 4
    x0p = np.zeros((B, ld), dtype=np.float32); x0p[:, :D] = x0
    d_x0 = dev(x0p)
    xL = torch.zeros(B, D, device=DEV)
    aux = torch.zeros(B, 2 * L, device=DEV)
    ops.cross_fwd(d_x0, dev(w), dev(b), xL, aux, ws)
    tx = torch.tensor(x0, dtype=torch.float64, requires_grad=True)
    tw = [torch.tensor(w[l].reshape(D, 1), dtype=torch.float64, requires_grad=True) for l in range(L)]
    tb = [torch.tensor(b[l].reshape(D, 1), dtype=torch.float64, requires_grad=True) for l in range(L)]
    ref = O.dcn_cross(tx, tw, tb)
    assert rel_err(xL.cpu().numpy(), ref.detach().numpy()) < 1e-5
    ref.backward(torch.tensor(g, dtype=torch.float64))
    dx0 = torch.zeros(B, ld, device=DEV)
    dw, db = torch.zeros(L, D, device=DEV), torch.zeros(L, D, device=DEV)
    ops.cross_bwd(d_x0, dev(w), dev(b), aux, dev(g), dx0, dw, db, ws)
    assert rel_err(dx0.cpu().numpy()[:, :D], tx.grad.numpy()) < 1e-5
    assert rel_err(dw.cpu().numpy(), np.stack([t.grad.numpy().reshape(-1) for t in tw])) < 1e-5
    assert rel_err(db.cpu().numpy(), np.stack([t.grad.numpy().reshape(-1) for t in tb])) < 1e-5


@pytest.mark.parametrize('D,L,B,acc', [(1677, 6, 8192, True), (1677, 6, 37, False), (50, 3, 37, True), (64, 2, 33, False),
                                       (1024, 3, 100, True), (2047, 6, 19, True), (429, 3, 65536, True),
                                       (300, 1, 5, False), (7, 8, 3, True)])
def test_cross_network_aligned_rows(ops_ws, D, L, B, acc):
    """Rows 16 bytes aligned and padded, the layout the DCN model feeds: the kernels then address memory through
    buffer descriptors, a sample per 16 lanes.  The padding of every input row holds NaN and that of every output
    row a sentinel: neither may leak into a result or be overwritten.  Shapes: the benchmark's, rows that end
    inside a lane's last quad, exactly at one, 1 and 32 quads per lane, batches that are not multiples of the four
    samples of a wave, with and without accumulation into dx0."""
    ops, ws = ops_ws
    rng = np.random.default_rng(7 * D + L)
    DP = (D + 3) // 4 * 4
    x0 = (rng.standard_normal((B, D)) * 0.5).astype(np.float32)
    w = (rng.standard_normal((L, D)) * 0.05).astype(np.float32)
    b = (rng.standard_normal((L, D)) * 0.05).astype(np.float32)
    g = rng.standard_normal((B, D)).astype(np.float32)
    old = rng.standard_normal((B, D)).astype(np.float32)

    def padded(a, width, fill):
        out = np.full((a.shape[0], width), fill, dtype=np.float32)
        out[:, :a.shape[1]] = a
        return dev(out)
    d_x0 = padded(x0, DP + 4, np.nan)[:, :D]
    d_g = padded(g, DP + 8, np.nan)[:, :D]
    xL_buf = torch.full((B, DP + 8), 7.0, device=DEV)
    dx_buf = padded(old if acc else np.zeros_like(old), DP, 7.0)
    aux = torch.zeros(B, 2 * L, device=DEV)
    ops.cross_fwd(d_x0, dev(w), dev(b), xL_buf[:, :D], aux, ws)
    tx = torch.tensor(x0, dtype=torch.float64, requires_grad=True)
    tw = [torch.tensor(w[l].reshape(D, 1), dtype=torch.float64, requires_grad=True) for l in range(L)]
    tb = [torch.tensor(b[l].reshape(D, 1), dtype=torch.float64, requires_grad=True) for l in range(L)]
    ref = O.dcn_cross(tx, tw, tb)
    out = xL_buf.cpu().numpy()
    assert (out[:, D:] == 7.0).all()
    assert rel_err(out[:, :D], ref.detach().numpy()) < 1e-5
    ref.backward(torch.tensor(g, dtype=torch.float64))
    dw, db = torch.zeros(L, D, device=DEV), torch.zeros(L, D, device=DEV)
    ops.cross_bwd(d_x0, dev(w), dev(b), aux, d_g, dx_buf[:, :D], dw, db, ws, accumulate=acc)
    dx = dx_buf.cpu().numpy()
    assert (dx[:, D:] == 7.0).all()
    want = tx.grad.numpy() + (old if acc else 0.0)
    assert rel_err(dx[:, :D], want) < 1e-5
    assert rel_err(dw.cpu().numpy(), np.stack([t.grad.numpy().reshape(-1) for t in tw])) < 1e-5
    assert rel_err(db.cpu().numpy(), np.stack([t.grad.numpy().reshape(-1) for t in tb])) < 1e-5


def test_cross_network_short_rows_ignore_stale_lds(ops_ws):
    """Rows much shorter than a wave's 256-column reach (D = 43): the lanes past the row must not read the LDS
    behind the staged weights.  A GEMM over NaN operands first leaves NaN bit patterns in every CU's LDS."""
    ops, ws = ops_ws
    rng = np.random.default_rng(43)
    B, D, L = 300, 43, 3
    x0 = (rng.standard_normal((B, D)) * 0.5).astype(np.float32)
    w = (rng.standard_normal((L, D)) * 0.05).astype(np.float32)
    b = (rng.standard_normal((L, D)) * 0.05).astype(np.float32)
    x0p = np.zeros((B, 44), dtype=np.float32); x0p[:, :D] = x0
    d_x0, d_w, d_b = dev(x0p), dev(w), dev(b)
    tx = torch.tensor(x0, dtype=torch.float64)
    ref = O.dcn_cross(tx, [torch.tensor(w[l].reshape(D, 1), dtype=torch.float64) for l in range(L)],
                      [torch.tensor(b[l].reshape(D, 1), dtype=torch.float64) for l in range(L)]).numpy()
    nan_a = torch.full((4096, 256), float('nan'), device=DEV)
    nan_b = torch.full((256, 256), float('nan'), device=DEV)
    nan_c = torch.zeros(4096, 256, device=DEV)
    for _ in range(5):
        ops.gemm(nan_a, nan_b, nan_c, ws=ws)
        xL = torch.zeros(B, D, device=DEV)
        aux = torch.zeros(B, 2 * L, device=DEV)
        ops.cross_fwd(d_x0, d_w, d_b, xL, aux, ws)
        out = xL.cpu().numpy()
        assert np.isfinite(out).all() and np.isfinite(aux.cpu().numpy()).all()
        assert rel_err(out, ref) < 1e-5


@pytest.mark.parametrize('B,N,E,A,keep', [(5, 2, 4, 1, 1.0), (70, 7, 8, 10, 1.0), (33, 39, 16, 16, 0.7),
                                          (9, 20, 64, 32, 1.0), (1100, 12, 16, 5, 0.5)])
def test_afm_attention_over_pairs_fwd_bwd(ops_ws, B, N, E, A, keep):
    """softmax over the pair axis: forward, dZ and the attention-parameter gradients against
    autograd in float64 on the same dropout masks."""
    ops, ws = ops_ws
    rng = np.random.default_rng(B * 131 + N)
    Z = (rng.standard_normal((B, N, E)) * 0.8).astype(np.float32)
    W = (rng.standard_normal((E, A)) * 0.5).astype(np.float32)
    b = (rng.standard_normal(A) * 0.3).astype(np.float32)
    h = rng.standard_normal(A).astype(np.float32)
    g = rng.standard_normal((B, E)).astype(np.float32)
    npairs = N * (N - 1) // 2
    drop = ops.make_dropout(1 - keep, 77, 4)
    m = torch.as_tensor(O.keep_mask(77, 4, B * npairs, keep).reshape(B, npairs, 1)) if keep < 1 else None
    tz, tW, tb, th = [torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (Z, W, b, h)]
    ii, jj = torch.triu_indices(N, N, offset=1)
    pw = tz[:, ii, :] * tz[:, jj, :]
    e = (torch.relu(pw @ tW + tb) * th).sum(2, keepdim=True)
    a = torch.softmax(e, dim=1)
    if m is not None:
        a = a * m / keep
    x_ref = (a * pw).sum(1)
    x_ref.backward(torch.tensor(g, dtype=torch.float64))
    ld = N * E + 8
    zp = torch.zeros(B, ld, device=DEV); zp[:, :N * E] = dev(Z.reshape(B, -1))
    x = torch.zeros(B, E, device=DEV)
    ops.afm_attn_fwd(zp, dev(W), dev(b), dev(h), x, N=N, E=E, pair_drop=drop)
    assert rel_err(x.cpu().numpy(), x_ref.detach().numpy()) < 1e-5
    dz = torch.full((B, ld), 7.0, device=DEV)
    dW, db, dh = torch.zeros(E, A, device=DEV), torch.zeros(A, device=DEV), torch.zeros(A, device=DEV)
    for _ in range(2):      # second call must overwrite, not accumulate
        ops.afm_attn_bwd(zp, dev(g), dev(W), dev(b), dev(h), dz, dW, db, dh, N=N, E=E, ws=ws, pair_drop=drop)
    assert_close(dz.cpu().numpy()[:, :N * E], tz.grad.numpy().reshape(B, -1), 1e-5, 'dz')
    assert (dz.cpu().numpy()[:, N * E:] == 7.0).all()
    assert_close(dW.cpu().numpy(), tW.grad.numpy(), 1e-5, 'dW')
    assert_close(db.cpu().numpy(), tb.grad.numpy(), 1e-5, 'db')
    assert_close(dh.cpu().numpy(), th.grad.numpy(), 1e-5, 'dh')


def test_afm_attention_over_pairs_at_the_benchmark_shape(ops_ws):
    """B = 65536, N = 39, E = 16, A = 10 (741 pairs): forward, dZ and the parameter gradients of the rows of a
    2048-sample slice against float64 autograd (the whole batch does not fit a float64 graph in seconds), the slice's
    rows taken from a run over the whole batch, so that grid caps and the per-wave partial sums are exercised; the
    parameter gradients of the whole batch against the sum of eight such slices' worth is replaced by linearity: the
    gradients of two half-batches add up to the gradient of the batch."""
    ops, ws = ops_ws
    B, N, E, A = 65536, 39, 16, 10
    gen = torch.Generator(device=DEV)
    gen.manual_seed(39)
    Z = torch.randn(B, N * E, generator=gen, device=DEV) * 0.5
    W = torch.randn(E, A, generator=gen, device=DEV) * 0.4
    b = torch.randn(A, generator=gen, device=DEV) * 0.2
    h = torch.randn(A, generator=gen, device=DEV)
    g = torch.randn(B, E, generator=gen, device=DEV)
    x = torch.zeros(B, E, device=DEV)
    ops.afm_attn_fwd(Z, W, b, h, x, N=N, E=E)
    dz = torch.zeros(B, N * E, device=DEV)
    dW, db, dh = torch.zeros(E, A, device=DEV), torch.zeros(A, device=DEV), torch.zeros(A, device=DEV)
    ops.afm_attn_bwd(Z, g, W, b, h, dz, dW, db, dh, N=N, E=E, ws=ws)
    sl = slice(30000, 32048)
    tz = Z[sl].double().reshape(-1, N, E).cpu().requires_grad_(True)
    tW, tb, th = [t.double().cpu().requires_grad_(True) for t in (W, b, h)]
    ii, jj = torch.triu_indices(N, N, offset=1)
    pw = tz[:, ii, :] * tz[:, jj, :]
    a = torch.softmax((torch.relu(pw @ tW + tb) * th).sum(2, keepdim=True), dim=1)
    x_ref = (a * pw).sum(1)
    x_ref.backward(g[sl].double().cpu())
    assert rel_err(x[sl].cpu().numpy(), x_ref.detach().numpy()) < 1e-5
    assert_close(dz[sl].cpu().numpy(), tz.grad.numpy().reshape(-1, N * E), 1e-5, 'dz')
    # parameter gradients: the slice alone against autograd, then additivity over the two halves of the batch
    dWs, dbs, dhs = torch.zeros(E, A, device=DEV), torch.zeros(A, device=DEV), torch.zeros(A, device=DEV)
    dzs = torch.zeros(2048, N * E, device=DEV)
    ops.afm_attn_bwd(Z[sl].contiguous(), g[sl].contiguous(), W, b, h, dzs, dWs, dbs, dhs, N=N, E=E, ws=ws)
    assert_close(dWs.cpu().numpy(), tW.grad.numpy(), 1e-5, 'dW')
    assert_close(dbs.cpu().numpy(), tb.grad.numpy(), 1e-5, 'db')
    assert_close(dhs.cpu().numpy(), th.grad.numpy(), 1e-5, 'dh')
    assert torch.equal(dzs, dz[sl])
    halves = []
    for lo in (0, B // 2):
        o = [torch.zeros(E, A, device=DEV), torch.zeros(A, device=DEV), torch.zeros(A, device=DEV)]
        ops.afm_attn_bwd(Z[lo:lo + B // 2], g[lo:lo + B // 2], W, b, h, torch.zeros(B // 2, N * E, device=DEV), *o, N=N,
                         E=E, ws=ws)
        halves.append(o)
    for whole, p0, p1, nm in zip((dW, db, dh), halves[0], halves[1], ('dW', 'db', 'dh')):
        assert_close(whole.cpu().numpy(), (p0.double() + p1.double()).cpu().numpy(), 1e-5, nm)


@pytest.mark.parametrize('B,E', [(7, 4), (300, 16), (65536, 16)])
def test_outer_product_fwd_bwd(ops_ws, B, E):
    """PNN's outer-product term (PNN/PNN.py:83-84): ss = s (x) s and its gradient, against float64."""
    ops, _ = ops_ws
    gen = torch.Generator(device=DEV)
    gen.manual_seed(B + E)
    s = torch.randn(B, E, generator=gen, device=DEV)
    ss = torch.zeros(B, E * E, device=DEV)
    ops.outer_fwd(s, ss)
    assert torch.equal(ss.view(B, E, E), s[:, :, None] * s[:, None, :])        # one rounding per element either way
    dss = torch.randn(B, E * E, generator=gen, device=DEV)
    ds = torch.zeros(B, E, device=DEV)
    ops.outer_bwd(dss, s, ds)
    d = dss.double().view(B, E, E)
    ref = ((d + d.transpose(1, 2)) * s.double()[:, None, :]).sum(2)
    assert_close(ds.cpu().numpy(), ref.cpu().numpy(), 1e-5, 'ds')


def test_bad_arguments_are_rejected(ops_ws):
    ops, ws = ops_ws
    from recommendation_amd._lib import CtrError
    W = torch.zeros(10, 12, device=DEV)
    idx = torch.zeros(4, 1, dtype=torch.int32, device=DEV)
    offs = dev(np.array([0, 10], dtype=np.int64))
    with pytest.raises(CtrError):       # E = 12 is not a power of two
        ops.embed_fwd(W, offs, idx, E=12, z_out=torch.zeros(4, 12, device=DEV))
    with pytest.raises(ValueError):     # GEMM inner dimensions disagree
        ops.gemm(torch.zeros(4, 5, device=DEV), torch.zeros(6, 3, device=DEV), torch.zeros(4, 3, device=DEV), ws=ws)
    with pytest.raises(ValueError):     # CPU tensors are refused: there is no CPU path
        ops.gemm(torch.zeros(4, 5), torch.zeros(5, 3), torch.zeros(4, 3), ws=ws)


@pytest.mark.parametrize('n,ties', [(1, False), (1000, False), (70001, True), (300000, True)])
def test_device_auc_and_logloss_match_the_host_metrics(ops_ws, n, ties):
    from recommendation_amd import train_loop as tl
    ops, ws = ops_ws
    rng = np.random.default_rng(n)
    p = rng.random(n).astype(np.float32)
    if ties:
        p = np.round(p, 2)                       # many equal scores
        p[:5] = [0.0, -0.0, 1.0, 0.5, 0.5]
    y = (rng.random(n) < 0.3).astype(np.float32)
    auc, ll = ops.auc_logloss(dev(p), dev(y), ws)
    ref = tl.auc_score(y, p)
    if np.isnan(ref):
        assert np.isnan(auc.item())
    else:
        assert abs(auc.item() - ref) < 1e-6
    assert abs(ll.item() - tl.log_loss(y, np.asarray(p, np.float64))) < 1e-5 * max(1.0, abs(tl.log_loss(y, p)))


@pytest.mark.parametrize('M,K0,widths,rate', [(1000, 256, (64, 32, 16), 0.0), (77, 128, (64, 32), 0.2),
                                               (4099, 64, (64, 32, 16), 0.3), (513, 128, (64, 32), 0.2),
                                               (65, 48, (64, 32), 0.0), (65536, 128, (64, 32), 0.2)])
def test_chained_tower_layers_match_layer_by_layer_gemms(ops_ws, M, K0, widths, rate):
    """one launch for the narrow layers == relu(h W + b) with the counter-hash dropout, layer by layer in float64"""
    ops, ws = ops_ws
    rng = np.random.default_rng(M + K0)
    x = (rng.standard_normal((M, K0)) * 0.5).astype(np.float32)
    Ws, bs, kin = [], [], K0
    for w in widths:
        Ws.append((rng.standard_normal((kin, w)) / np.sqrt(kin)).astype(np.float32))
        bs.append((rng.standard_normal(w) * 0.1).astype(np.float32))
        kin = w
    assert ops.mlp_chain_supported(K0, widths)
    assert not ops.mlp_chain_supported(K0 + 1, widths) and not ops.mlp_chain_supported(K0, (128, 64))
    xd = torch.zeros(M, K0 + 4, device=DEV)
    xd[:, :K0] = dev(x)
    big = torch.full((M, sum(widths) + 12), 5.0, device=DEV)          # outputs as strided views of one buffer
    outs, c = [], 4
    for i, w in enumerate(widths):
        if i == len(widths) - 1 and M % 2:
            c += 3                                                    # the last output may start at any column
        outs.append(big[:, c:c + w])
        c += w
    drops = [ops.make_dropout(rate, 99, 10 + i) for i in range(len(widths))]
    ops.mlp_chain_fwd(xd[:, :K0], [dev(W) for W in Ws], [dev(b) for b in bs], drops, outs)
    h = x.astype(np.float64)
    for i, w in enumerate(widths):
        h = np.maximum(h @ Ws[i].astype(np.float64) + bs[i], 0.0)
        if rate:
            h = h * O.keep_mask(99, 10 + i, M * w, 1 - rate).reshape(M, w) / (1 - rate)
        assert rel_err(outs[i].cpu().numpy(), h) < 1e-5, i
    assert (big[:, :4] == 5.0).all() and (big[:, c:] == 5.0).all()
    if M % 2:
        assert (big[:, c - widths[-1] - 3:c - widths[-1]] == 5.0).all()


@pytest.mark.parametrize('B,K,mask_from', [(1000, 74, 42), (65, 255, -1), (4099, 17, 0), (3, 1, 1), (515, 2701, 1677), (33, 256, 100),
                                           (70, 4096, -1), (65536, 74, 42), (8192, 2701, 1677)])
def test_fused_output_unit_backward(ops_ws, B, K, mask_from):
    """dX = dl w^T with the relu/dropout mask on the tail columns, dw = X^T dl, db = sum dl, in one pass"""
    ops, ws = ops_ws
    rng = np.random.default_rng(B + K)
    x = rng.standard_normal((B, K)).astype(np.float32)
    dl = (rng.standard_normal(B) / B).astype(np.float32)
    w = rng.standard_normal(K).astype(np.float32)
    xd = torch.zeros(B, K + 3, device=DEV)
    xd[:, :K] = dev(x)
    dxd = torch.full((B, K + 5), 7.0, device=DEV)
    dw, db = torch.zeros(K, device=DEV), torch.zeros(1, device=DEV)
    assert ops.unit_bwd_supported(K) and not ops.unit_bwd_supported(4097)
    amax = torch.zeros(ops.AMAX_FLOATS, device=DEV)
    ops.unit_bwd(dev(dl), dev(w), xd[:, :K], dxd[:, :K], dw, db, mask_from=mask_from, mask_scale=1.25, masked_amax=amax,
                 ws=ws)
    ref = dl[:, None].astype(np.float64) * w[None, :]
    if mask_from >= 0:
        ref[:, mask_from:] *= np.where(x[:, mask_from:] > 0, 1.25, 0.0)
    assert rel_err(dxd[:, :K].cpu().numpy(), ref) < 1e-6
    # the record holds exactly the largest magnitude written to the masked columns (what ctr_absmax would find there)
    got = dxd[:, :K].cpu().numpy()
    want = float(np.abs(got[:, mask_from:]).max()) if 0 <= mask_from < K else 0.0
    assert float(amax.max().item()) == want
    assert (dxd[:, K:] == 7.0).all()
    assert rel_err(dw.cpu().numpy(), x.astype(np.float64).T @ dl) < 1e-5
    assert abs(float(db.item()) - float(dl.astype(np.float64).sum())) < 1e-6


@pytest.mark.parametrize('rows,ca,cb,extra', [(777, 26, 16, 5), (50, 200, 100, 0), (300, 1, 3, 2), (64, 0, 40, 1)])
def test_dropout_apply2_matches_two_single_block_passes(ops_ws, rows, ca, cb, extra):
    """Two neighbouring column blocks under their own dropouts in one launch (wide blocks: one launch each) against
    ctr_dropout_apply per block; columns past the blocks untouched; a missing descriptor leaves its block alone."""
    ops, ws = ops_ws
    g = torch.Generator(device=DEV); g.manual_seed(rows + ca)
    x = torch.randn(rows, ca + cb + extra, device=DEV, generator=g)
    da, db = ops.make_dropout(0.4, seed=11, stream_id=3), ops.make_dropout(0.25, seed=11, stream_id=4)
    for a, b in ((da, db), (None, db), (da, None)):
        ref = x.clone()
        if a is not None and ca:
            ops.dropout_apply(ref[:, :ca], a)
        if b is not None:
            ops.dropout_apply(ref[:, ca:ca + cb], b)
        out = x.clone()
        ops.dropout_apply2(out[:, :ca + cb], ca, a, cb, b)
        assert torch.equal(out, ref)


@pytest.mark.parametrize('B,K,ca,cb', [(3000, 74, 26, 16), (257, 40, 0, 8), (1000, 600, 26, 16)])
def test_output_unit_backward_with_keep_masks(ops_ws, B, K, ca, cb):
    """ctr_unit_bwd_keep == ctr_unit_bwd followed by ctr_dropout_apply2 on dX's leading column blocks, bit for bit
    (narrow kernel: masks inside the pass; wide kernel: the pass of their own), dw / db untouched by the masks."""
    ops, ws = ops_ws
    g = torch.Generator(device=DEV); g.manual_seed(B + K)
    x = torch.randn(B, K, device=DEV, generator=g)
    dl = torch.randn(B, device=DEV, generator=g)
    w = torch.randn(K, device=DEV, generator=g)
    da = ops.make_dropout(0.5, seed=3, stream_id=1) if ca else None
    db = ops.make_dropout(0.3, seed=3, stream_id=2)
    out = []
    for keep in (None, (ca, da, cb, db)):
        dx = torch.zeros(B, K, device=DEV)
        dw, dbias = torch.zeros(K, device=DEV), torch.zeros(1, device=DEV)
        ops.unit_bwd(dl, w, x, dx, dw, dbias, mask_from=ca + cb, mask_scale=1.25, keep=keep, ws=ws)
        out.append((dx, dw, dbias))
    ref = out[0][0].clone()
    ops.dropout_apply2(ref[:, :ca + cb], ca, da, cb, db)
    assert torch.equal(out[1][0], ref) and not torch.equal(out[1][0], out[0][0])
    assert torch.equal(out[1][1], out[0][1]) and torch.equal(out[1][2], out[0][2])


@pytest.mark.parametrize('M,widths,offset', [(1000, (32, 64, 128), 0), (77, (16, 32, 64), 3), (4099, (32, 64, 128), 2),
                                             (65536, (32, 64, 128), 0)])
def test_chained_activation_gradients_match_layer_by_layer(ops_ws, M, widths, offset):
    """two dX steps in one launch == (g W^T) * relu/dropout mask, layer by layer in float64"""
    ops, ws = ops_ws
    n_in, mid, out = widths
    rng = np.random.default_rng(M)
    g_in = rng.standard_normal((M, n_in)).astype(np.float32)
    Wb = (rng.standard_normal((mid, n_in)) / 8).astype(np.float32)
    Wa = (rng.standard_normal((out, mid)) / 8).astype(np.float32)
    h_mid = np.maximum(rng.standard_normal((M, mid)), 0).astype(np.float32)
    h_out = np.maximum(rng.standard_normal((M, out)), 0).astype(np.float32)
    big = torch.zeros(M, n_in + 7, device=DEV)                # the incoming gradient may start at any column
    big[:, offset:offset + n_in] = dev(g_in)
    g_mid, g_out = torch.full((M, mid), 9.0, device=DEV), torch.full((M, out + 4), 9.0, device=DEV)
    assert ops.mlp_chain_bwd_supported(*widths) and not ops.mlp_chain_bwd_supported(32, 64, 256)
    ops.mlp_chain_bwd(big[:, offset:offset + n_in], dev(Wb), dev(h_mid), 1.25, g_mid, dev(Wa), dev(h_out), 2.0,
                      g_out[:, :out])
    r_mid = (g_in.astype(np.float64) @ Wb.astype(np.float64).T) * np.where(h_mid > 0, 1.25, 0.0)
    r_out = (r_mid @ Wa.astype(np.float64).T) * np.where(h_out > 0, 2.0, 0.0)
    assert rel_err(g_mid.cpu().numpy(), r_mid) < 1e-5
    assert rel_err(g_out[:, :out].cpu().numpy(), r_out) < 1e-5
    assert (g_out[:, out:] == 9.0).all()


def test_embed_assemble_and_routed_dz_match_the_plain_layout(ops_ws):
    """Z built from per-owner column blocks == the FM front end on the dense Z; dZ routed into the block layout
    == dZ in the plain layout copied block by block (bit-exact: same arithmetic, other addresses)."""
    ops, ws = ops_ws
    B, F, C, E = 130, 7, 3, 16
    ranges = [(0, 3), (3, 3), (3, 7)]                       # an owner without fields in the middle
    rng = np.random.default_rng(3)
    z = rng.standard_normal((B, F * E)).astype(np.float32)
    xc = rng.standard_normal((B, C)).astype(np.float32)
    flat, fbase, fstride, off = [], [], [], 0
    for a, b in ranges:
        flat.append(z[:, a * E:b * E].reshape(-1))
        for n in range(a, b):
            fbase.append(off + (n - a) * E)
            fstride.append((b - a) * E)
        off += B * (b - a) * E
    blocks = dev(np.concatenate(flat))
    fb, fs = torch.tensor(fbase, device=DEV), torch.tensor(fstride, device=DEV)
    ld = F * E + 4
    d1, d2 = ops.make_dropout(0.5, 3, 1), ops.make_dropout(0.5, 3, 2)
    outs = []
    for mode in ('plain', 'blocks'):
        zo = torch.zeros(B, ld, device=DEV)
        first, s, second = torch.zeros(B, F, device=DEV), torch.zeros(B, E, device=DEV), torch.zeros(B, E, device=DEV)
        if mode == 'plain':
            zo[:, :F * E] = dev(z)
            zo[:, F * E:F * E + C] = dev(xc)
            ops.fm_fwd(zo, N=F, E=E, first_out=first, s_out=s, second_out=second, drop_first=d1, drop_second=d2)
        else:
            ops.embed_assemble_fwd(blocks, fb, fs, dev(xc), None, F=F, E=E, z_out=zo, cont_tail=True, first_out=first,
                                   s_out=s, second_out=second, drop_first=d1, drop_second=d2)
        outs.append((zo, first, s, second))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    zo, first, s, second = outs[0]
    gz = dev(rng.standard_normal((B, ld)).astype(np.float32))
    g1, g2 = dev(rng.standard_normal((B, F)).astype(np.float32)), dev(rng.standard_normal((B, E)).astype(np.float32))
    dz_plain = torch.zeros(B, ld, device=DEV)
    ops.embed_bwd_dz(dz_plain, B=B, N=F, E=E, gz=gz, g_first=g1, g_second=g2, s=s, z=zo, drop_first=d1, drop_second=d2)
    dz_other, routed = torch.zeros(B, ld, device=DEV), torch.zeros_like(blocks)
    ops.embed_bwd_dz(dz_other, B=B, N=F, E=E, gz=gz, g_first=g1, g_second=g2, s=s, z=zo, drop_first=d1, drop_second=d2,
                     blocks_out=routed, fbase=fb, fstride=fs, F=F)
    want = torch.cat([dz_plain[:, a * E:b * E].reshape(-1) for a, b in ranges])
    assert torch.equal(routed, want)
    assert (dz_other == 0).all()                           # every row of this Z belongs to a sharded field


@pytest.mark.parametrize('B,N,E,P,per_sample', [(1001, 39, 16, 50, True), (65536, 39, 16, 50, True), (7, 7, 16, 5, False), (130, 64, 16, 64, True),
                                                (66, 17, 16, 33, False), (258, 39, 8, 50, True), (33, 5, 32, 9, True)])
def test_pnn_inner_product_term_forward_and_backward(ops_ws, B, N, E, P, per_sample):
    """u = W Z per sample: usq = sum_e u^2; given coef: dZ += W^T (coef * u), dW = sum_b (coef * u) Z^T.
    E = 16 takes the all-MFMA kernel, the other widths the LDS kernel; both against float64."""
    ops, ws = ops_ws
    rng = np.random.default_rng(B + N + P)
    z = rng.standard_normal((B, N, E)).astype(np.float32)
    w = (rng.standard_normal((P, N)) / np.sqrt(N)).astype(np.float32)
    coef = rng.standard_normal((B, P) if per_sample else (P,)).astype(np.float32)
    dz0 = rng.standard_normal((B, N, E)).astype(np.float32)
    ld = N * E + 4
    zd = torch.zeros(B, ld, device=DEV)
    zd[:, :N * E] = dev(z.reshape(B, -1))
    usq = torch.zeros(B, P, device=DEV)
    ops.pnn_inner_fwd(zd[:, :N * E], dev(w), usq, N=N, E=E)
    u = np.einsum('pn,bne->bpe', w.astype(np.float64), z.astype(np.float64))
    assert rel_err(usq.cpu().numpy(), (u * u).sum(-1)) < 1e-5
    dzd = torch.full((B, ld), 3.0, device=DEV)
    dzd[:, :N * E] = dev(dz0.reshape(B, -1))
    dw = torch.zeros(P, N, device=DEV)
    ops.pnn_inner_bwd(zd[:, :N * E], dev(w), dev(coef), dzd[:, :N * E], dw, N=N, E=E, ws=ws)
    du = u * (coef.astype(np.float64)[:, :, None] if per_sample else coef.astype(np.float64)[None, :, None])
    dz_ref = dz0 + np.einsum('pn,bpe->bne', w.astype(np.float64), du)
    dw_ref = np.einsum('bpe,bne->pn', du, z.astype(np.float64))
    assert rel_err(dzd[:, :N * E].cpu().numpy().reshape(B, N, E), dz_ref) < 1e-5
    assert rel_err(dw.cpu().numpy(), dw_ref) < 1e-5
    assert (dzd[:, N * E:] == 3.0).all()


@pytest.mark.parametrize('B,N,E,rate', [(1001, 39, 16, 0.3), (65536, 39, 16, 0.3), (5, 2, 16, 0.0), (130, 64, 16, 0.5), (66, 17, 16, 0.0),
                                        (77, 39, 8, 0.3), (9, 5, 4, 0.0)])
def test_afm_pair_layer_forward_and_backward(ops_ws, B, N, E, rate):
    """x[e] = sum_{i<j} m_ij z_i[e] z_j[e] with the per-pair dropout multipliers of the counter hash; backward
    dZ_i[e] = dx[e] sum_{j != i} m_ij z_j[e].  E = 16 takes the matrix-core kernel, other widths the LDS kernel."""
    ops, ws = ops_ws
    rng = np.random.default_rng(B + N)
    z = rng.standard_normal((B, N, E)).astype(np.float32)
    dx = rng.standard_normal((B, E)).astype(np.float32)
    P = N * (N - 1) // 2
    drop = ops.make_dropout(rate, 21, 4)
    m = np.ones((B, P))
    if rate:
        m = O.keep_mask(21, 4, B * P, 1 - rate).reshape(B, P) / (1 - rate)
    M = np.zeros((B, N, N))
    iu = np.triu_indices(N, 1)                                # lexicographic (i, j), i < j: the kernel's pair order
    M[:, iu[0], iu[1]] = m
    M = M + M.transpose(0, 2, 1)
    y = np.einsum('bij,bje->bie', M, z.astype(np.float64))
    ld = N * E + 4
    zd = torch.zeros(B, ld, device=DEV)
    zd[:, :N * E] = dev(z.reshape(B, -1))
    x = torch.zeros(B, E, device=DEV)
    ops.afm_pair_fwd(zd[:, :N * E], x, N=N, E=E, pair_drop=drop)
    assert rel_err(x.cpu().numpy(), 0.5 * (z * y).sum(1)) < 1e-5
    dzd = torch.full((B, ld), 2.0, device=DEV)
    ops.afm_pair_bwd(zd[:, :N * E], dev(dx), dzd[:, :N * E], N=N, E=E, pair_drop=drop)
    assert rel_err(dzd[:, :N * E].cpu().numpy().reshape(B, N, E), dx[:, None, :] * y) < 1e-5
    assert (dzd[:, N * E:] == 2.0).all()


@pytest.mark.parametrize('M,ka,na,kb,nb,off', [(65536, 128, 64, 64, 32, 3), (1000, 128, 64, 64, 32, 0),
                                               (37, 64, 32, 32, 16, 1), (5000, 64, 32, 32, 16, 0)])
def test_tail_wgrad2_matches_two_float64_products(ops_ws, M, ka, na, kb, nb, off):
    """Both weight gradients and both bias gradients of the last two tower layers from one launch; gb sits at a
    column offset of a wider buffer (the output unit's concat gradient), rows not 16-byte aligned when off % 4."""
    ops, ws = ops_ws
    assert ops.tail_wgrad2_supported(ka, na, kb, nb) and not ops.tail_wgrad2_supported(256, 128, 128, 64)
    g = torch.Generator().manual_seed(M + ka)
    xa, ga, xb = (torch.randn(M, c, generator=g).to(DEV) for c in (ka, na, kb))
    wide = torch.randn(M, nb + off + 2, generator=g).to(DEV)
    gb = wide[:, off:off + nb]
    outs = []
    for rep in range(2):
        dWa, dba = torch.full((ka, na), 7.0, device=DEV), torch.full((na,), 7.0, device=DEV)
        dWb, dbb = torch.full((kb, nb), 7.0, device=DEV), torch.full((nb,), 7.0, device=DEV)
        ops.tail_wgrad2(xa, ga, dWa, dba, xb, gb, dWb, dbb, ws=ws)
        outs.append((dWa, dba, dWb, dbb))
    d = lambda t: t.double().cpu()
    refs = (d(xa).t() @ d(ga), d(ga).sum(0), d(xb).t() @ d(gb), d(gb).sum(0))
    for got, ref, name in zip(outs[0], refs, ('dWa', 'dba', 'dWb', 'dbb')):
        assert_close(got.cpu().numpy(), ref.numpy(), 1e-5, name)
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)                                   # overwrites, fixed summation order


@pytest.mark.parametrize('M,n_in,mid,out,off', [(65536, 32, 64, 128, 55), (1000, 32, 64, 128, 0), (37, 16, 32, 64, 1),
                                                (5000, 16, 32, 64, 0)])
def test_tail_bwd_fused_matches_chain_backward_and_float64_weight_gradients(ops_ws, M, n_in, mid, out, off):
    """One launch = mlp_chain_bwd's output gradient (bit for bit: same products in the same order) + the two
    layers' weight and bias gradients against float64; g_in sits at a column offset of a wider buffer."""
    ops, ws = ops_ws
    g = torch.Generator().manual_seed(M + out)
    Wb = (torch.randn(mid, n_in, generator=g) * 0.2).to(DEV)
    Wa = (torch.randn(out, mid, generator=g) * 0.2).to(DEV)
    h_mid = torch.relu(torch.randn(M, mid, generator=g)).to(DEV)
    h_out = torch.relu(torch.randn(M, out, generator=g)).to(DEV)
    wide = torch.randn(M, n_in + off + 1, generator=g).to(DEV)
    g_in = wide[:, off:off + n_in]
    sm, so = 1.25, 2.0
    g_mid_ref, g_out_ref = torch.zeros(M, mid, device=DEV), torch.zeros(M, out, device=DEV)
    ops.mlp_chain_bwd(g_in, Wb, h_mid, sm, g_mid_ref, Wa, h_out, so, g_out_ref)
    runs = []
    for rep in range(2):
        g_out = torch.full((M, out), 7.0, device=DEV)
        dWa, dba = torch.full((out, mid), 7.0, device=DEV), torch.full((mid,), 7.0, device=DEV)
        dWb, dbb = torch.full((mid, n_in), 7.0, device=DEV), torch.full((n_in,), 7.0, device=DEV)
        ops.tail_bwd_fused(g_in, Wb, h_mid, sm, Wa, h_out, so, g_out, dWa, dba, dWb, dbb, ws=ws)
        runs.append((g_out, dWa, dba, dWb, dbb))
    g_out, dWa, dba, dWb, dbb = runs[0]
    assert torch.equal(g_out, g_out_ref)
    d = lambda t: t.double().cpu()
    gm64 = (d(g_in) @ d(Wb).t()) * (d(h_mid) > 0) * sm
    assert_close(dWb.cpu().numpy(), (d(h_mid).t() @ d(g_in)).numpy(), 1e-5, 'dWb')
    assert_close(dbb.cpu().numpy(), d(g_in).sum(0).numpy(), 1e-5, 'dbb')
    assert_close(dWa.cpu().numpy(), (d(h_out).t() @ gm64).numpy(), 1e-5, 'dWa')
    assert_close(dba.cpu().numpy(), gm64.sum(0).numpy(), 1e-5, 'dba')
    for a, b in zip(runs[0], runs[1]):
        assert torch.equal(a, b)

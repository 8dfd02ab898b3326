"""Host side of the entry points (no GPU): loader contract, batching quirk, DIN iterators, AUC."""
import os
import sys

import numpy as np
import pytest

import recommendation_amd as ra
from recommendation_amd import train_loop as tl

ra.enable_zoo_imports()


def _write_csvs(tmp_path, n_train=37, n_test=11):
    import pandas as pd
    rng = np.random.default_rng(0)
    for name, n in (('train', n_train), ('test', n_test)):
        df = pd.DataFrame({'userid': rng.integers(0, 100, n), 'itemid': rng.integers(0, 100, n),
                           'label': rng.integers(0, 2, n), 'user_type': rng.integers(0, 4, n),
                           'item_type': rng.integers(2, 9, n), 'c1': rng.standard_normal(n) * 3 + 1,
                           'c2': rng.standard_normal(n)})
        df.to_csv(os.path.join(tmp_path, name + '.csv'), index=False)


def test_load_dataset_contract(tmp_path):
    _write_csvs(str(tmp_path))
    Xtr_c, Xtr_f, ytr, Xte_c, Xte_f, yte, cate_list = tl.load_dataset(str(tmp_path))
    assert Xtr_c.shape == (37, 2) and Xte_c.shape == (11, 2) and Xtr_f.shape == (37, 2)
    allc = np.concatenate([Xtr_c, Xte_c])
    assert cate_list == [len(np.unique(allc[:, 0])) + 1, len(np.unique(allc[:, 1])) + 1]
    allf = np.concatenate([Xtr_f, Xte_f])                 # scaler is fit over train + test together
    assert np.allclose(allf.mean(0), 0, atol=1e-12) and np.allclose(allf.std(0), 1, atol=1e-12)


def test_batches_reproduce_the_reference_loop():
    assert tl.batches(10, 4) == [(0, 4), (4, 8), (8, 10)]
    assert tl.batches(8, 4) == [(0, 4), (4, 8)]
    assert tl.batches(4, 4) == [] and tl.batches(3, 4) == []     # total <= batch_size: no step at all
    assert tl.batches(5, 4) == [(0, 4), (4, 5)]


def test_flags_defaults():
    from DeepFM import main as dm
    from PNN import main as pm
    a = dm.make_parser().parse_args([])
    assert (a.embed_size, a.batch_size, a.epoch, a.learning_rate, a.decay_steps, a.decay_rate) == (8, 128, 10, 0.001, 10000, 0.99)
    assert tl.int_list(a.deep_layers) == [256, 128, 64, 32] and a.dropout_fm == 0.5 and a.dropout_deep == 0.2
    p = pm.make_parser().parse_args(['--no-inner_product', '--hidden_layers', '8,4'])
    assert p.inner_product is False and p.hidden_layers == [8, 4] and p.product_size == 50


def test_din_iterators_and_auc():
    from DIN import data
    from DIN.main import calc_auc, score_records
    train = [(1, [3, 4], 5, 1), (2, [6], 7, 0), (3, [1, 2, 3], 9, 1)]
    it = data.DataInput(train, 2)
    k, (u, i, y, hist, sl) = next(it)
    assert k == 1 and u == [1, 2] and i == [5, 7] and y == [1, 0] and sl == [2, 1]
    assert hist.dtype == np.int64 and hist.tolist() == [[3, 4], [6, 0]]
    k, (u, i, y, hist, sl) = next(it)
    assert hist.tolist() == [[1, 2, 3]]
    with pytest.raises(StopIteration):
        next(it)
    test = [(1, [3], (5, 6)), (2, [4, 4], (7, 8))]
    _, (u, i, j, hist, sl) = next(data.DataInputTest(test, 8))
    assert i == [5, 7] and j == [6, 8]
    pn = np.array([[0.9, 0.1], [0.8, 0.3], [0.2, 0.7]])
    auc = calc_auc(score_records(pn))
    pos, neg = pn[:, 0], pn[:, 1]
    exact = np.mean([(p > n) + 0.5 * (p == n) for p in pos for n in neg])
    assert abs(auc - exact) < 1e-12
    assert calc_auc([[0, 1, 0.5]] * 4) == -0.5


def test_auc_score_matches_pair_counting():
    rng = np.random.default_rng(1)
    y = rng.integers(0, 2, 200)
    p = np.round(rng.random(200), 2)                      # ties on purpose
    pos, neg = p[y == 1], p[y == 0]
    exact = np.mean([(a > b) + 0.5 * (a == b) for a in pos for b in neg])
    assert abs(tl.auc_score(y, p) - exact) < 1e-12


def test_wide_deep_loader_builds_the_cross_feature(tmp_path):
    """Wide_Deep/main.py:32-85: cross id = index of the '-'-joined type strings in sorted order,
    cardinalities = distinct + 1, numeric columns standardised over train + test."""
    import importlib.util
    import pandas as pd
    rng = np.random.default_rng(0)

    def frame(n):
        return pd.DataFrame(dict(userid=rng.integers(0, 50, n), itemid=rng.integers(0, 80, n),
                                 label=rng.integers(0, 2, n), user_type=rng.integers(0, 3, n),
                                 item_type=rng.integers(0, 4, n), f0=rng.standard_normal(n) * 3 + 1,
                                 f1=rng.standard_normal(n)))
    tr, te = frame(40), frame(15)
    tr.to_csv(tmp_path / 'train.csv', index=False)
    te.to_csv(tmp_path / 'test.csv', index=False)
    spec = importlib.util.spec_from_file_location(
        'wd_main', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'recommendation_amd',
                                'zoo', 'Wide_Deep', 'main.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    Xr, Xc, Xf, y, Xr2, Xc2, Xf2, y2, cross_list, cate_list = mod.load_dataset(str(tmp_path))
    assert Xr.shape == (40, 1) and Xc.shape == (40, 2) and Xf.shape == (40, 2) and Xr2.shape == (15, 1)
    both = pd.concat([tr, te])
    joined = (both['user_type'].astype(str) + '-' + both['item_type'].astype(str)).values
    uniq = np.unique(joined)
    assert cross_list == [len(uniq) + 1] and cate_list == [both['user_type'].nunique() + 1, both['item_type'].nunique() + 1]
    assert np.array_equal(Xr[:, 0], np.searchsorted(uniq, joined[:40]))
    allf = np.concatenate([Xf, Xf2])
    assert np.allclose(allf.mean(0), 0, atol=1e-9) and np.allclose(allf.std(0), 1, atol=1e-9)
    assert mod.make_parser().parse_args([]).hidden_layers in ('[32, 64, 128]', [32, 64, 128])


def test_ncf_loader_reindexes_in_first_appearance_order(tmp_path):
    """NCF/main.py:12-50: ids are mapped to their rank of first appearance over train then test."""
    import importlib.util
    import pandas as pd
    tr = pd.DataFrame(dict(userid=[30, 10, 30, 20], itemid=[5, 5, 9, 7], label=[1, 0, 1, 0]))
    te = pd.DataFrame(dict(userid=[20, 40], itemid=[9, 1], label=[0, 1]))
    tr.to_csv(tmp_path / 'train.csv', index=False)
    te.to_csv(tmp_path / 'test.csv', index=False)
    spec = importlib.util.spec_from_file_location(
        'ncf_main', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'recommendation_amd',
                                 'zoo', 'NCF', 'main.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    Xtr, ytr, Xte, yte, nu, ni = mod.load_dataset(str(tmp_path))
    assert (nu, ni) == (4, 4)
    assert Xtr.tolist() == [[0, 0], [1, 0], [0, 1], [2, 2]] and Xte.tolist() == [[2, 1], [3, 3]]
    assert ytr.tolist() == [1, 0, 1, 0] and yte.tolist() == [0, 1]
    assert mod.make_parser().parse_args([]).embed_size == 16

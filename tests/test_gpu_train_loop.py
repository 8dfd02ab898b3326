"""End-to-end on the GPU: the main.py-style loop (CSV in, epochs, checkpoint round trip) and the
N-step AUC / log-loss parity against the CPU oracle (bar: within 1e-3)."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import recommendation_amd as ra
from oracle import ctr_oracle as O
from recommendation_amd import train_loop as tl

pytestmark = pytest.mark.gpu
ra.enable_zoo_imports()


def _signal_data(n, rng, cards, C):
    Xc = np.stack([rng.integers(0, v - 1, n) for v in cards], 1)
    Xf = rng.standard_normal((n, C))
    score = 0.8 * Xf[:, 0] - 0.5 * Xf[:, 1] + 0.6 * (Xc[:, 0] % 2) - 0.4 * (Xc[:, 1] % 3 == 0)
    y = (score + 0.3 * rng.standard_normal(n) > 0).astype(np.int64)
    return Xc, Xf, y


@pytest.mark.parametrize('mode', ['R', 'S'])
def test_fit_loop_learns_and_checkpoint_round_trips(sess, tmp_path, mode):
    import pandas as pd
    from DeepFM import DeepFM
    rng = np.random.default_rng(0)
    for name, n in (('train', 1500), ('test', 400)):
        Xc, Xf, y = _signal_data(n, rng, [5, 7], 3)
        pd.DataFrame({'userid': 0, 'itemid': 0, 'label': y, 'user_type': Xc[:, 0], 'item_type': Xc[:, 1],
                      'c1': Xf[:, 0], 'c2': Xf[:, 1], 'c3': Xf[:, 2]}).to_csv(tmp_path / (name + '.csv'), index=False)
    args = SimpleNamespace(input_dir=str(tmp_path), model_name='DeepFM.ckpt', embed_size=8, deep_layers=[32, 16],
                           batch_size=128, epoch=4, regular_rate=0.1, learning_rate=0.01, dropout_fm=0.0,
                           dropout_deep=0.0, decay_steps=10000, decay_rate=0.99, mode=mode, seed=1, init_std=0.05)
    lines = []
    model = tl.fit(DeepFM.DeepFM, args, session=sess, log=lines.append)
    train_lines = [l for l in lines if 'training: loss' in l]
    losses = [float(l.split('loss = ')[1].split(',')[0]) for l in train_lines]
    assert len(losses) == 4 and losses[-1] < losses[0] - 0.05
    assert any('Accuracy = 1.0000' in l for l in lines)          # the reference's expression is 1.0 by construction
    data = tl.load_dataset(str(tmp_path))
    p1 = model.predict_proba(sess, data[3], data[4])
    assert tl.auc_score(data[5], p1) > 0.75
    # a fresh model restored from the checkpoint predicts identically and keeps global_step / Adam slots
    args2 = SimpleNamespace(**vars(args))
    m2 = DeepFM.DeepFM(args2, 2, 3, data[6])
    m2.build()
    from recommendation_amd.model_base import get_checkpoint_state
    ck = get_checkpoint_state(os.path.join(str(tmp_path), 'DeepFM.ckpt'))
    assert ck is not None
    m2.saver.restore(sess, ck)
    assert m2.global_step == model.global_step
    assert np.array_equal(m2.predict_proba(sess, data[3], data[4]), p1)
    l_a = model.run(sess, data[0][:128], data[1][:128], data[2][:128], True)
    l_b = m2.run(sess, data[0][:128], data[1][:128], data[2][:128], True)
    assert l_a == l_b
    for (k, a), b in zip(model.get_params().items(), m2.get_params().values()):
        assert np.array_equal(a, b), k


@pytest.mark.parametrize('name', ['DeepFM', 'DCN'])
def test_n_step_auc_logloss_parity(sess, name):
    """40 training steps from identical weights, dropout off: held-out log-loss and AUC of the HIP path
    stay within 1e-3 of the oracle's (float64)."""
    import importlib
    rng = np.random.default_rng(3)
    cards, C, B = [11, 23, 5, 40], 4, 256
    cfg = O.default_cfg(name, embed_size=8, mode='S', learning_rate=0.01)
    cfg.cate_num, cfg.cont_num, cfg.cate_list = len(cards), C, cards
    cfg.deep_layers = [32, 16]
    for k in ('dropout_fm', 'dropout_deep', 'dropout'):
        if hasattr(cfg, k):
            setattr(cfg, k, 0.0)
    cfg.adam_mode = 'tf1'
    cfg.init_std = 0.05
    P = O.init_params(name, cfg, seed=2, std=0.05)
    orc = O.OracleModel(name, cfg, P, dtype=torch.float64)
    m = getattr(importlib.import_module('%s.%s' % (name, name)), name)(cfg, len(cards), C, cards)
    m.build()
    m.initialize(sess, params=P)
    for _ in range(40):
        Xc, Xf, y = _signal_data(B, rng, cards, C)
        lo = orc.run(Xc, Xf, y, True)
        lg = m.run(sess, Xc, Xf, y, True)
        assert abs(float(lo) - float(lg)) < 1e-3
    Xc, Xf, y = _signal_data(4000, rng, cards, C)
    p_g = m.predict_proba(sess, Xc, Xf)
    p_o = orc.forward(Xc, Xf).y_out.numpy()
    assert abs(tl.log_loss(y, p_g) - tl.log_loss(y, p_o)) < 1e-3
    assert abs(tl.auc_score(y, p_g) - tl.auc_score(y, p_o)) < 1e-3
    assert tl.auc_score(y, p_g) > 0.7


def test_feeder_delivers_the_reference_batches_in_order(sess):
    """Pinned, multi-slot host -> device feed: same ranges as the reference loop, narrowed dtypes,
    exact values, more batches than staging slots (slots are reused)."""
    from recommendation_amd.feeder import Feeder
    from recommendation_amd.train_loop import batches
    rng = np.random.default_rng(3)
    n, F, C = 1003, 5, 3
    Xc = rng.integers(0, 1 << 20, (n, F))                  # int64, as pandas delivers it
    Xf = rng.standard_normal((n, C))                       # float64
    y = rng.integers(0, 2, n)
    ranges = batches(n, 100)
    assert ranges[-1] == (900, 1000) or ranges[-1][1] <= n
    got = []
    for idx, xc, yy in Feeder(sess, Xc, Xf, y, ranges, depth=2):
        assert idx.dtype == torch.int32 and xc.dtype == torch.float32 and yy.dtype == torch.float32
        got.append((idx.cpu().numpy().copy(), xc.cpu().numpy().copy(), yy.cpu().numpy().copy()))
    assert len(got) == len(ranges)
    for (s, e), (a, b, c) in zip(ranges, got):
        assert np.array_equal(a, Xc[s:e].astype(np.int32))
        assert np.array_equal(b, Xf[s:e].astype(np.float32))
        assert np.array_equal(c, y[s:e].astype(np.float32))

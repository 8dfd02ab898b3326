"""Shared builders for the parity tests (oracle = checker, never the thing under test)."""
from types import SimpleNamespace

import numpy as np

from oracle import ctr_oracle as O


def tiny_cfg(model, mode, E=4, **kw):
    cfg = O.default_cfg(model, embed_size=E, mode=mode)
    cfg.cate_num, cfg.cont_num, cfg.cate_list = 3, 2, [3, 5, 4]
    if hasattr(cfg, 'deep_layers'):
        cfg.deep_layers = [8, 4]
    if model == 'PNN':
        cfg.hidden_layers = [8, 4]
        cfg.product_size = 5
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


def tiny_batch(cfg, B=8, seed=0, oor=True):
    rng = np.random.default_rng(seed)
    Xc = np.stack([rng.integers(0, v, B) for v in cfg.cate_list], 1).astype(np.int64)
    if oor and B >= 4:
        Xc[0, 0] = cfg.cate_list[0] + 4      # out of range (high)
        Xc[1, 1] = -1                        # out of range (negative)
        Xc[2, 2] = 0
        Xc[3, 2] = 1
    Xf = rng.standard_normal((B, cfg.cont_num))
    y = rng.integers(0, 2, B).astype(np.int64)
    return Xc, Xf, y


def no_dropout(cfg):
    for k in ('dropout_fm', 'dropout_deep', 'dropout_bi', 'dropout'):
        if hasattr(cfg, k):
            setattr(cfg, k, 0.0)
    return cfg


def rel_err(a, b, floor=1e-30):
    """max-norm relative error; `floor` bounds the denominator from below for quantities that are
    analytically zero (a sum of cancelling fp32 terms is then compared on an absolute scale)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + floor))

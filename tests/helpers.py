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


def elem_err(a, b, atol_scale=0.1):
    """Element-wise companion of rel_err: max over elements of |a-b| / (|b| + atol_scale * max|b|).
    An element much smaller than the tensor's largest is still held to a relative bound (against
    atol_scale * max|b| at worst) instead of disappearing under the max-norm."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if b.size == 0:
        return 0.0
    den = np.abs(b) + atol_scale * np.max(np.abs(b)) + 1e-300
    return float(np.max(np.abs(a - b) / den))


def assert_close(a, b, tol, name='', elem_tol=None, atol_scale=0.1):
    """Both bars: max-norm relative error < tol, and element-wise |a-b| < elem_tol * (|b| + atol_scale*max|b|)
    (elem_tol defaults to 10 * tol: with atol_scale = 0.1 the element-wise bar is never looser than the
    max-norm one at the small elements and holds every element above 10 % of the maximum to ~tol * 10)."""
    r = rel_err(a, b)
    assert r < tol, '%s: max-norm rel err %.3g >= %.3g' % (name, r, tol)
    e = elem_err(a, b, atol_scale)
    et = 10 * tol if elem_tol is None else elem_tol
    assert e < et, '%s: element-wise err %.3g >= %.3g' % (name, e, et)


def deepfm_masks(cfg, B, seed, N=None):
    """Keep masks of one DeepFM training step from the counter hash the kernels use (streams 1, 2, 10+i)."""
    import torch
    N = cfg.cate_num if N is None else N
    E = cfg.embed_size
    masks = {'fm_first': O.keep_mask(seed, 1, B * N, 1 - cfg.dropout_fm).reshape(B, N),
             'fm_second': O.keep_mask(seed, 2, B * E, 1 - cfg.dropout_fm).reshape(B, E)}
    for i, L in enumerate(cfg.deep_layers):
        masks['deep_%d' % i] = O.keep_mask(seed, 10 + i, B * L, 1 - cfg.dropout_deep).reshape(B, L)
    return {k: torch.as_tensor(v) for k, v in masks.items()}


def sync_oracle_from_gpu(orc, m):
    """Teacher forcing for multi-step checks: copy the GPU model's weights, Adam/Adagrad slots and step count into
    the oracle, so that the next step is compared from identical state.  Free-running trajectories of a relu network
    separate chaotically (a 1e-7 weight difference flips a relu decision and moves a gradient column by per cents),
    which says nothing about the arithmetic of a step."""
    import torch
    for k, v in m.get_params().items():
        orc.P[k] = torch.as_tensor(v).to(orc.dtype)
    opt = orc.opt
    slots = [('m', 'slot0'), ('v', 'slot1')] if hasattr(opt, 'm') else [('acc', 'slot0')] if hasattr(opt, 'acc') else []
    for attr, slot in slots:
        dst = getattr(opt, attr)
        for k in m.dense.shapes:
            dst[k] = m.dense._view(getattr(m.dense, slot), k).cpu().to(orc.dtype)
        for f in range(m.cate_num):
            dst['embedding/embed_cate_w_%d' % f] = m.tables.table(f, getattr(m.tables, slot)).cpu().to(orc.dtype)
    if hasattr(opt, 't'):
        opt.t = m.global_step
    orc.global_step = m.global_step


def assert_weights_track(P_gpu, P_ref, name='', tol=1e-4):
    """Weights after a few free-running optimizer steps from identical starts: max-norm relative `tol` (1e-4).
    Not 1e-5: an optimizer step divides by sqrt(accumulated g^2) (Adam also adds eps = 1e-8 there), which turns the
    gradient's fp32 rounding (held to 1e-5 of the tensor's largest element by the gradient checks) into a larger
    relative change of the step for the elements whose gradient is small, and three steps add up.  The one-step,
    identical-state form of this check (test_gpu_configs._step_and_compare) is the tight one."""
    for k, v in P_gpu.items():
        ref = P_ref[k].numpy() if hasattr(P_ref[k], 'numpy') else np.asarray(P_ref[k])
        r = rel_err(v, ref)
        assert r < tol, '%s %s: weights differ by %.3g (max-norm relative)' % (name, k, r)


def assert_sorted_pairs(keys, src, key_ref, src_ref, total):
    """keys / src (uint32 arrays from ctr_emb_sort) against the definition: the pairs (key_ref[i], src_ref[i]) in
    stable key order.  Pairs with the sentinel key `total` (out-of-range ids) sit behind everything; their order
    among themselves is by sample or by field depending on the sort path, so the tail is compared as a set."""
    key_ref = np.asarray(key_ref).astype(np.uint32)
    src_ref = np.asarray(src_ref).astype(np.uint32)
    order = np.argsort(key_ref, kind='stable')
    nv = int((key_ref != total).sum())
    assert np.array_equal(keys[:nv], key_ref[order][:nv])
    assert np.array_equal(src[:nv], src_ref[order][:nv])
    assert np.all(keys[nv:] == total)
    assert np.array_equal(np.sort(src[nv:]), np.sort(src_ref[order][nv:]))

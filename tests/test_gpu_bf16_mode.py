"""The opt-in reduced-precision mode of the tile GEMM (bf16 operands, one MFMA per product, fp32 accumulate): the
product is what its definition says - the float64 product of the bf16-rounded operands, to fp32 accumulation
accuracy - and a DeepFM trained with it reaches the same AUC / log loss as the fp32-accurate default within 1e-3
(the end-to-end bar for reduced-precision towers).  The default mode is restored whatever happens."""
import numpy as np
import pytest
import torch

from oracle import ctr_oracle as O
from tests.helpers import rel_err
from tests.test_gpu_models import _build

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture
def bf16_mode(sess):
    from recommendation_amd import ops
    prev = ops.set_gemm_mode('bf16')
    yield ops
    ops.set_gemm_mode(prev)


@pytest.mark.parametrize('M,N,K,ta,tb', [(4096, 256, 429, 0, 0), (4096, 429, 256, 0, 1), (429, 256, 4096, 1, 0),
                                          (8192, 1024, 1000, 0, 0), (300, 70, 33, 1, 1)])
def test_bf16_product_is_the_product_of_the_rounded_operands(sess, bf16_mode, M, N, K, ta, tb):
    ops = bf16_mode
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn((K, M) if ta else (M, K), generator=g).to(DEV)
    B = torch.randn((N, K) if tb else (K, N), generator=g).to(DEV)
    C = torch.zeros(M, N, device=DEV)
    ops.gemm(A, B, C, trans_a=bool(ta), trans_b=bool(tb), ws=sess.ws)
    r = lambda t: t.to(torch.bfloat16).to(torch.float64).cpu()
    a, b = r(A), r(B)
    ref = (a.t() if ta else a) @ (b.t() if tb else b)
    assert rel_err(C.cpu().numpy(), ref.numpy()) < 2e-6            # exact bf16 products, fp32 accumulation
    exact = (A.double().cpu().t() if ta else A.double().cpu()) @ (B.double().cpu().t() if tb else B.double().cpu())
    assert 1e-4 < rel_err(C.cpu().numpy(), exact.numpy()) < 2e-2   # and really reduced precision


def _auc(score, y):
    order = np.argsort(score, kind='stable')
    ranks = np.empty(len(score)); ranks[order] = np.arange(1, len(score) + 1)
    pos = y > 0.5
    return (ranks[pos].sum() - pos.sum() * (pos.sum() + 1) / 2) / (pos.sum() * (~pos).sum())


def test_training_in_bf16_mode_reaches_the_same_auc_and_logloss(sess):
    """40 steps of DeepFM at the reference's default tower (its big layers go through the tile kernel) on labels a
    logistic teacher draws from the inputs, once per mode from the same initial weights and the same batches."""
    from recommendation_amd import ops
    cfg = O.default_cfg('DeepFM', embed_size=16, mode='S')
    rng = np.random.default_rng(17)
    cfg.cate_list = [int(v) for v in rng.integers(20, 400, 26)]
    cfg.cate_num, cfg.cont_num = 26, 13
    cfg.adam_mode = 'lazy'
    P = O.init_params('DeepFM', cfg, seed=5, std=0.05)
    B, steps = 4096, 40
    teacher = [rng.standard_normal(v) for v in cfg.cate_list]
    wc = rng.standard_normal(13) * 0.5

    def batch():
        Xc = np.stack([rng.integers(0, v, B) for v in cfg.cate_list], 1)
        Xf = rng.standard_normal((B, 13))
        logit = sum(teacher[f][Xc[:, f]] for f in range(26)) * 0.35 + Xf @ wc
        y = (rng.random(B) < 1 / (1 + np.exp(-logit))).astype(np.int64)
        return Xc, Xf, y
    data = [batch() for _ in range(steps)]
    test = batch()
    res = {}
    for mode in ('split', 'bf16'):
        prev = ops.set_gemm_mode(mode)
        try:
            m = _build(sess, 'DeepFM', cfg, P)
            for Xc, Xf, y in data:
                m.run(sess, Xc, Xf, y, True)
            p = np.clip(m.predict_proba(sess, test[0], test[1]).astype(np.float64), 1e-7, 1 - 1e-7)
        finally:
            ops.set_gemm_mode(prev)
        yt = test[2].astype(np.float64)
        res[mode] = (_auc(p, yt), float(-np.mean(yt * np.log(p) + (1 - yt) * np.log(1 - p))))
    assert res['split'][0] > 0.6                                   # the model did learn
    assert abs(res['split'][0] - res['bf16'][0]) < 1e-3, res
    assert abs(res['split'][1] - res['bf16'][1]) < 1e-3, res
    assert ops.get_gemm_mode() == 'split'

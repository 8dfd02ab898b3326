"""The two-term fp16 split of the tile GEMM (ctr_epilogue_t.a_amax / b_amax): accuracy against float64 over
magnitudes far outside fp16's own range, the largest-magnitude words (ctr_absmax, ctr_absmax_tensors, c_amax_out),
and that the bf16 split is what runs without them."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def ops_ws(sess):
    from recommendation_amd import ops
    return ops, sess.ws


def _mk(gen, r, c, scale):
    return (torch.randn(r, (c + 3) // 4 * 4, generator=gen, device=DEV) * scale)[:, :c]


def _amax(ops, *ts):
    """One largest-magnitude record per tensor, plus a spare one for an output."""
    am = torch.zeros(len(ts) + 1, ops.AMAX_FLOATS, device=DEV)
    for i, t in enumerate(ts):
        ops.absmax(t, am[i])
    return am


@pytest.mark.parametrize('M,N,K,ta,tb', [(1000, 256, 429, False, False), (1000, 429, 256, False, True),
                                         (429, 256, 5000, True, False), (300, 70, 45, False, False),
                                         (2100, 1024, 700, False, True), (2048, 640, 1024, False, False),
                                         (257, 33, 130, True, True), (4096, 64, 256, False, True)])
@pytest.mark.parametrize('sa,sb', [(1.0, 1.0), (3e-7, 2e5), (4e4, 1e-3)])
def test_fp16_split_matches_float64(ops_ws, M, N, K, ta, tb, sa, sb):
    """Operands scaled far below and above what fp16 holds: the per-matrix power of two keeps the split exact, and
    the product stays within 2e-6 of sum |a||b| (the bf16 split's bar in test_gpu_kernels is 1e-5)."""
    ops, ws = ops_ws
    gen = torch.Generator(device=DEV)
    gen.manual_seed(M + 3 * N + 7 * K)
    A = _mk(gen, K, M, sa) if ta else _mk(gen, M, K, sa)
    Bm = _mk(gen, N, K, sb) if tb else _mk(gen, K, N, sb)
    am = _amax(ops, A, Bm)
    assert float(am[0].max()) == float(A.abs().max()) and float(am[1].max()) == float(Bm.abs().max())
    C = torch.full((M, N + 4), 7.0, device=DEV)
    ops.gemm(A, Bm, C[:, :N], trans_a=ta, trans_b=tb, ws=ws, a_amax=am[0], b_amax=am[1], amax_out=am[2])
    Ad, Bd = (A.double().t() if ta else A.double()), (Bm.double().t() if tb else Bm.double())
    ref = Ad @ Bd
    scale = float((Ad.abs() @ Bd.abs()).max())
    err = float((C[:, :N].double() - ref).abs().max())
    assert err < 2e-6 * scale, (err / scale)
    assert err < 2e-5 * float(ref.abs().max())
    assert bool((C[:, N:] == 7.0).all())
    assert float(am[2].max()) == float(C[:, :N].abs().max())           # c_amax_out: exactly the largest stored magnitude
    # the same call with the fp16 split switched off takes the bf16 split (other bits, same bar)
    C2 = torch.zeros(M, N, device=DEV)
    prev = ops.set_gemm_f16_split(False)
    try:
        ops.gemm(A, Bm, C2, trans_a=ta, trans_b=tb, ws=ws, a_amax=am[0], b_amax=am[1])
    finally:
        ops.set_gemm_f16_split(prev)
    assert float((C2.double() - ref).abs().max()) < 1e-6 * scale
    if M * N > 10000 and K > 256:                                # (narrow layers with K <= 256 stream in plain fp32)
        assert not torch.equal(C2, C[:, :N])


def test_fp16_split_with_crest_factor_and_epilogues(ops_ws):
    """A few huge elements beside many small ones (gradient-like), a column of ones covered by a_amax_floor, and the
    bias + relu / mask epilogues on the unscaled result."""
    ops, ws = ops_ws
    gen = torch.Generator(device=DEV)
    gen.manual_seed(11)
    M, N, K = 3000, 256, 300
    A = _mk(gen, M, K, 1e-4)
    A[::97, ::13] *= 3e4                                            # crest factor ~1e5
    Bm = _mk(gen, K, N, 0.05)
    bias = torch.randn(N, generator=gen, device=DEV) * 1e-3
    am = _amax(ops, A, Bm)
    C = torch.zeros(M, N, device=DEV)
    ops.gemm(A, Bm, C, bias=bias, relu=True, ws=ws, a_amax=am[0], b_amax=am[1])
    ref = torch.relu(A.double() @ Bm.double() + bias.double())
    scale = float((A.double().abs() @ Bm.double().abs()).max())
    assert float((C.double() - ref).abs().max()) < 2e-6 * scale
    # rows that hold only small elements keep their own relative accuracy (the split is exact down to 2^-40 of the
    # matrix maximum, not 2^-22 of it)
    small = torch.ones(M, dtype=torch.bool, device=DEV)
    small[::97] = False
    row_scale = (A.double().abs() @ Bm.double().abs()).max(dim=1).values
    assert float(((C.double() - ref).abs().max(dim=1).values / row_scale)[small].max()) < 5e-6
    # A with a column of ones its producer never saw: the floor keeps the scale valid
    Aext = torch.zeros(M, K + 4, device=DEV)
    Aext[:, :K] = A * 1e-3
    Aext[:, K] = 1.0
    am2 = _amax(ops, Aext[:, :K], Bm)
    G = _mk(gen, M, N, 1e-5)
    am3 = _amax(ops, G)
    dW = torch.zeros(K + 1, N, device=DEV)
    ops.dense_wgrad(Aext[:, :K + 1], G, dW, None, ws=ws, x_amax=am2[0], g_amax=am3[0], x_amax_floor=1.0)
    refw = Aext[:, :K + 1].double().t() @ G.double()
    assert float((dW.double() - refw).abs().max()) < 2e-6 * float((Aext[:, :K + 1].double().abs().t() @ G.double().abs()).max())
    assert bool(torch.isfinite(dW).all())


def test_absmax_tensors(ops_ws):
    ops, _ = ops_ws
    gen = torch.Generator(device=DEV)
    gen.manual_seed(3)
    flat = torch.randn(100000, generator=gen, device=DEV)
    offs = torch.tensor([0, 16, 5000, 99999], dtype=torch.int64, device=DEV)
    cnts = torch.tensor([16, 4000, 90000, 1], dtype=torch.int64, device=DEV)
    out = torch.zeros(4, ops.AMAX_FLOATS, device=DEV)
    ops.absmax_tensors(flat, offs, cnts, out)
    for i in range(4):
        o, c = int(offs[i]), int(cnts[i])
        assert float(out[i].max()) == float(flat[o:o + c].abs().max())

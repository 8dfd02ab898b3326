"""The fused DIN attention scoring network (ctr_din_attn_mlp_fwd / _bwd / _wgrad) against the literal graph of
DIN/DIN.py:169-185 - concat [q, k, q-k, q*k], dense 80 sigmoid, dense 40 sigmoid, dense 1 - in float64 with
autograd, through the C ABI.  1e-5 relative (max-norm plus the element-wise bar) on every output and gradient;
two runs are bitwise equal (fixed summation order)."""
import numpy as np
import pytest
import torch

from tests.helpers import assert_close

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TOL = 1e-5


def _case(B, T, H, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g, dtype=torch.float64) * sc)
    return dict(q=r(B, H), k=r(B * T, H), W1=r(4 * H, 80, sc=0.2), b1=r(80, sc=0.1), W2=r(80, 40, sc=0.2),
                b2=r(40, sc=0.1), w3=r(40, sc=0.3), b3=r(1, sc=0.1), ds=r(B * T), dk0=r(B * T, H))


def _reference(c, B, T, H):
    P = {n: c[n].clone().requires_grad_(True) for n in ('q', 'k', 'W1', 'b1', 'W2', 'b2', 'w3', 'b3')}
    qq = P['q'].view(B, 1, H).expand(B, T, H).reshape(B * T, H)
    din_all = torch.cat([qq, P['k'], qq - P['k'], qq * P['k']], 1)
    p1 = din_all @ P['W1'] + P['b1']
    d1 = torch.sigmoid(p1)
    p2 = d1 @ P['W2'] + P['b2']
    d2 = torch.sigmoid(p2)
    score = d2 @ P['w3'] + P['b3']
    names = list(P)
    grads = torch.autograd.grad(score, [P[n] for n in names] + [p1, p2], grad_outputs=c['ds'])
    G = dict(zip(names + ['p1', 'p2'], grads))
    return d1.detach(), d2.detach(), score.detach(), G


@pytest.mark.parametrize('B,T,H', [(5, 200, 32), (64, 33, 16), (3, 7, 32), (9, 16, 32), (2, 50, 64), (11, 40, 64),
                                   (6, 200, 128), (1500, 3, 128)])
def test_din_attn_mlp_matches_the_literal_graph(sess, B, T, H):
    from recommendation_amd import ops
    c = _case(B, T, H, 100 * B + T + H)
    d1r, d2r, sr, G = _reference(c, B, T, H)
    f = {n: v.to(torch.float32).to(DEV).contiguous() for n, v in c.items()}
    z = lambda *s: torch.zeros(*s, device=DEV)
    d1, d2, score = z(B * T, 80), z(B * T, 40), z(B * T)
    ops.din_attn_mlp_fwd(f['q'], f['k'], f['W1'], f['b1'], f['W2'], f['b2'], f['w3'], f['b3'], score, B=B, T=T, H=H,
                         d1=d1, d2=d2)
    assert_close(d1.cpu().numpy(), d1r.numpy(), TOL, 'd1')
    assert_close(d2.cpu().numpy(), d2r.numpy(), TOL, 'd2')
    assert_close(score.cpu().numpy(), sr.numpy(), TOL, 'score')
    s2 = z(B * T)
    ops.din_attn_mlp_fwd(f['q'], f['k'], f['W1'], f['b1'], f['W2'], f['b2'], f['w3'], f['b3'], s2, B=B, T=T, H=H)
    assert torch.equal(score, s2)                              # inference form (nothing kept) gives the same bits
    g1, g2, dq = z(B * T, 80), z(B * T, 40), z(B, H)
    dk = f['dk0'].clone()
    ops.din_attn_mlp_bwd(f['q'], f['k'], f['W1'], f['W2'], f['w3'], d1, d2, f['ds'], g1, g2, dq, dk, B=B, T=T, H=H,
                         acc_k=True)
    assert_close(g2.cpu().numpy(), G['p2'].numpy(), TOL, 'g2')
    assert_close(g1.cpu().numpy(), G['p1'].numpy(), TOL, 'g1')
    assert_close(dq.cpu().numpy(), G['q'].numpy(), TOL, 'dq')
    assert_close((dk - f['dk0']).cpu().numpy(), G['k'].numpy(), 2 * TOL, 'dk (accumulated onto a unit-scale buffer)')
    dk2 = z(B * T, H)
    ops.din_attn_mlp_bwd(f['q'], f['k'], f['W1'], f['W2'], f['w3'], d1, d2, f['ds'], g1, g2, dq, dk2, B=B, T=T, H=H)
    assert_close(dk2.cpu().numpy(), G['k'].numpy(), TOL, 'dk')
    # H = 64 / 128: the weight-gradient launch runs in 2 / 4 column slices; (1500, 3, 128) has more samples than
    # waves, so every slice's waves walk several samples
    out = dict(W1=z(4 * H, 80), b1=z(80), W2=z(80, 40), b2=z(40), w3=z(40), b3=z(1))
    for rep in range(2):
        o = out if rep == 0 else {n: torch.full_like(v, 7.0) for n, v in out.items()}
        ops.din_attn_mlp_wgrad(f['q'], f['k'], d1, d2, f['ds'], g1, g2, o['W1'], o['b1'], o['W2'], o['b2'], o['w3'],
                               o['b3'], B=B, T=T, H=H, ws=sess.ws)
        for n in out:
            if rep == 0:
                assert_close(o[n].cpu().numpy(), G[n].numpy().reshape(o[n].shape), TOL, 'd' + n)
            else:
                assert torch.equal(o[n], out[n]), n               # overwrites, and the same bits again


@pytest.mark.parametrize('B,T,H', [(37, 200, 32), (9, 50, 64), (300, 21, 128), (6, 16, 16)])
def test_din_attn_mlp_with_sequence_lengths_skips_only_exact_zeros(sess, B, T, H):
    """The _sl forms (strips of 16 positions from a sample's length on are not computed) against the plain calls with
    the gradient of the masked positions set to zero, as the masked softmax of DIN/DIN.py:186-190 delivers it: scores
    and kept activations of every valid position, dq, dk and all parameter gradients bit for bit; skipped strips of
    the outputs untouched; lengths 0, 1, a multiple of 16, T and beyond T among them."""
    from recommendation_amd import ops
    c = _case(B, T, H, 7 * B + T + H)
    rng = np.random.default_rng(B + T)
    sl_np = rng.integers(0, T + 1, B).astype(np.int32)
    sl_np[:5] = [0, 1, 16, T, T + 9][:min(5, B)]
    valid = (np.arange(T)[None, :] < np.minimum(sl_np, T)[:, None]).reshape(-1)
    c['ds'] = c['ds'] * torch.as_tensor(valid, dtype=torch.float64)
    f = {n: v.to(torch.float32).to(DEV).contiguous() for n, v in c.items()}
    sl = torch.as_tensor(sl_np, device=DEV)
    computed = (np.arange(T)[None, :] // 16 < (np.minimum(sl_np, T)[:, None] + 15) // 16).reshape(-1)   # rows of run strips
    cm = torch.as_tensor(computed, device=DEV)
    z = lambda *s: torch.zeros(*s, device=DEV)
    outs = []
    for lens in (None, sl):
        fill = 0.0 if lens is None else 123.0
        d1, d2, score = z(B * T, 80) + fill, z(B * T, 40) + fill, z(B * T) + fill
        ops.din_attn_mlp_fwd(f['q'], f['k'], f['W1'], f['b1'], f['W2'], f['b2'], f['w3'], f['b3'], score, B=B, T=T, H=H,
                             d1=d1, d2=d2, sl=lens)
        g1, g2, dq = z(B * T, 80) + fill, z(B * T, 40) + fill, z(B, H)
        dk = f['dk0'].clone()
        ops.din_attn_mlp_bwd(f['q'], f['k'], f['W1'], f['W2'], f['w3'], d1, d2, f['ds'], g1, g2, dq, dk, B=B, T=T, H=H,
                             acc_k=True, sl=lens)
        w = dict(W1=z(4 * H, 80), b1=z(80), W2=z(80, 40), b2=z(40), w3=z(40), b3=z(1))
        ops.din_attn_mlp_wgrad(f['q'], f['k'], d1, d2, f['ds'], g1, g2, w['W1'], w['b1'], w['W2'], w['b2'], w['w3'],
                               w['b3'], B=B, T=T, H=H, ws=sess.ws, sl=lens)
        outs.append(dict(d1=d1, d2=d2, score=score, g1=g1, g2=g2, dq=dq, dk=dk, **{'d' + n: v for n, v in w.items()}))
    full, part = outs
    for n in ('d1', 'd2', 'score', 'g1', 'g2'):
        assert torch.equal(full[n][cm], part[n][cm]), n
        assert bool((part[n][~cm] == 123.0).all()), n + ' written in a skipped strip'
    for n in ('dq', 'dk', 'dW1', 'db1', 'dW2', 'db2', 'dw3', 'db3'):
        assert torch.equal(full[n], part[n]), n


def test_din_attn_mlp_rejects_unsupported_shapes(sess):
    from recommendation_amd import ops
    assert all(ops.din_attn_mlp_supported(h) for h in (16, 32, 64, 128))
    assert not ops.din_attn_mlp_supported(48) and not ops.din_attn_mlp_supported(256)
    assert not ops.din_attn_mlp_supported(32, 64, 40)
    c = _case(2, 4, 48, 1)
    f = {n: v.to(torch.float32).to(DEV).contiguous() for n, v in c.items()}
    with pytest.raises(Exception):
        ops.din_attn_mlp_fwd(f['q'], f['k'], f['W1'], f['b1'], f['W2'], f['b2'], f['w3'], f['b3'],
                             torch.zeros(8, device=DEV), B=2, T=4, H=48)

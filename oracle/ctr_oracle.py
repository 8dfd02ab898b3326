"""CPU restatement of the reference CTR graphs (TEST INFRASTRUCTURE ONLY).

This module is the *checker* for the HIP path: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it.  The product path never does.

PARITY UNPINNED: the reference (LiuJun523/recommendation) is TensorFlow-1.x graph
code, ships no tests, golden vectors or fixtures, and TensorFlow is not installable
here, so nothing produced by the reference itself pins this restatement.  It is
written from the reference's source text, op by op, and is cross-checked three ways
in tests/test_oracle.py: literal graph emulation vs closed forms, torch.autograd vs
hand-derived gradients, and float64 finite differences.

Everything is torch-CPU; `dtype` selects float64 (master) or float32 (shadow).

Reference files restated (file:line relative to the reference root):
  shared one-hot embedding block   DeepFM/DeepFM.py:42-53 (copies in DCN/NFM/AFM/PNN/ESMM)
  continuous-value embedding       AFM/AFM.py:52-57, PNN/PNN.py:55-59
  DeepFM                           DeepFM/DeepFM.py:55-100
  DCN                              DCN/DCN.py:55-101
  NFM                              NFM/NFM.py:52-93
  PNN                              PNN/PNN.py:60-131
  AFM                              AFM/AFM.py:59-97
  ESMM                             ESMM/ESMM.py:50-107
  DIN                              DIN/DIN.py:17-110,149-196
  standard lookup precedent        Wide_Deep/Wide_Deep.py:51, DIN/DIN.py:26
"""
from __future__ import annotations

import math
from collections import OrderedDict
from types import SimpleNamespace

import numpy as np
import torch

# ----------------------------------------------------------------------------
# Deterministic keep-mask generator shared (bit for bit) with the device code.
# 32-bit multiply/xorshift finalizer over (element index, seed, stream id); 24 random
# bits compared with floor(keep_prob * 2^24).
# ----------------------------------------------------------------------------


def keep_threshold(keep_prob: float) -> int:
    return int(math.floor(float(keep_prob) * float(1 << 24)))


def hash24(seed: int, stream: int, index: np.ndarray) -> np.ndarray:
    """The counter hash shared with the HIP kernels (csrc/common.h hash24), vectorised over index."""
    m64 = (1 << 64) - 1
    k = (int(seed) * 0xBF58476D1CE4E5B9 + int(stream) * 0x94D049BB133111EB) & m64
    k ^= k >> 29
    k0, k1 = np.uint32(k & 0xFFFFFFFF), np.uint32(k >> 32)
    with np.errstate(over="ignore"):
        i = np.asarray(index, dtype=np.uint64)
        lo = (i & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        hi = (i >> np.uint64(32)).astype(np.uint32)
        h = (lo ^ k0) * np.uint32(0x9E3779B1)
        h ^= h >> np.uint32(15)
        h = (h ^ k1 ^ (hi * np.uint32(0x85EBCA77))) * np.uint32(0x85EBCA6B)
        h ^= h >> np.uint32(13)
        h *= np.uint32(0xC2B2AE35)
        h ^= h >> np.uint32(16)
    return (h >> np.uint32(8)).astype(np.int64)


def keep_mask(seed: int, stream: int, n: int, keep_prob: float) -> np.ndarray:
    """0/1 float32 mask of n elements; element i kept iff hash24(i) < thr."""
    u = hash24(seed, stream, np.arange(n, dtype=np.uint64))
    return (u < keep_threshold(keep_prob)).astype(np.float32)


# ----------------------------------------------------------------------------
# TF-1.x primitives
# ----------------------------------------------------------------------------
def tf_one_hot(idx: torch.Tensor, depth: int) -> torch.Tensor:
    """tf.one_hot(idx, depth, dtype=int32): out-of-range index -> all-zero row."""
    ar = torch.arange(depth, dtype=torch.int64).view(1, depth)
    return (idx.view(-1, 1).to(torch.int64) == ar).to(torch.int64)


def embed_block_rmode_literal(tables, X_cate):
    """DeepFM/DeepFM.py:43-53 literally: one_hot -> embedding_lookup(W, one_hot)
    -> * float(one_hot) -> concat(axis=1).  Returns Z [B, sum(V), E]."""
    outs = []
    for f, W in enumerate(tables):
        V = W.shape[0]
        onehot = tf_one_hot(X_cate[:, f], V)                 # [B,V] in {0,1}
        embed = W[onehot]                                    # [B,V,E]: rows 0/1 of W
        value = onehot.to(W.dtype).view(-1, V, 1)
        outs.append(embed * value)
    return torch.cat(outs, dim=1)


def embed_block_rmode(tables, X_cate):
    """Closed form of the block above: Z[b, off_f + x, :] = W_f[1, :] if x in range."""
    B = X_cate.shape[0]
    E = tables[0].shape[1]
    outs = []
    for f, W in enumerate(tables):
        V = W.shape[0]
        onehot = tf_one_hot(X_cate[:, f], V).to(W.dtype)     # [B,V]
        outs.append(onehot.view(B, V, 1) * W[1].view(1, 1, E))
    return torch.cat(outs, dim=1)


class BagIds:
    """Multi-hot input in CSR form: bag (b,f) = ids[offsets[b*F+f] : offsets[b*F+f+1]].  Not a reference input:
    the reference feeds one id per field (DeepFM/DeepFM.py:43-53), which is the one-member case; the pooling is
    the published sum combiner of tf.nn.embedding_lookup_sparse."""

    def __init__(self, ids, offsets, F):
        self.ids = torch.as_tensor(np.asarray(ids).astype(np.int32).astype(np.int64)).reshape(-1)
        self.offsets = torch.as_tensor(np.asarray(offsets).astype(np.int64)).reshape(-1)
        self.F = int(F)
        self.B = (self.offsets.numel() - 1) // self.F
        counts = self.offsets[1:] - self.offsets[:-1]
        self.bag = torch.repeat_interleave(torch.arange(self.B * self.F), counts)     # bag of every member

    def __len__(self):
        return self.B


def embed_bags_smode(tables, bags):
    """Z[b,f,:] = sum over the members of bag (b,f) of W_f[id,:]; out-of-range members add nothing."""
    outs = []
    for f, W in enumerate(tables):
        V = W.shape[0]
        sel = (bags.bag % bags.F == f) & (bags.ids >= 0) & (bags.ids < V)
        z = torch.zeros(bags.B, W.shape[1], dtype=W.dtype)
        outs.append(z.index_add(0, bags.bag[sel] // bags.F, W[bags.ids[sel]]))
    return torch.stack(outs, dim=1)


def embed_block_smode(tables, X_cate):
    """Standard per-field lookup Z[b,f,:] = W_f[x[b,f],:] (Wide_Deep/Wide_Deep.py:51,
    DIN/DIN.py:26); an out-of-range id gives a zero row (the tf.one_hot convention)."""
    if isinstance(X_cate, BagIds):
        return embed_bags_smode(tables, X_cate)
    outs = []
    for f, W in enumerate(tables):
        V = W.shape[0]
        x = X_cate[:, f].to(torch.int64)
        ok = (x >= 0) & (x < V)
        rows = W[torch.where(ok, x, torch.zeros_like(x))]
        outs.append(rows * ok.to(W.dtype).view(-1, 1))
    return torch.stack(outs, dim=1)


def embed_cont(Wc, X_cont):
    """AFM/AFM.py:52-57: embed_w [C,E] * X_cont[B,C,1]."""
    return Wc.unsqueeze(0) * X_cont.unsqueeze(2)


def dropout(x, rate, keep):
    """tf.layers.dropout in training mode with an explicit 0/1 keep mask;
    keep=None is inference mode (identity)."""
    if keep is None or rate == 0.0:
        return x
    return x * keep.to(x.dtype).view(x.shape) * (1.0 / (1.0 - rate))


def log_loss(y, p, eps=1e-7):
    """tf.losses.log_loss: mean over the batch of -y log(p+eps) - (1-y) log(1-p+eps)."""
    return torch.mean(-y * torch.log(p + eps) - (1.0 - y) * torch.log(1.0 - p + eps))


def sigmoid_ce_with_logits(labels, logits):
    """max(x,0) - x*z + log(1+exp(-|x|))."""
    x, z = logits, labels
    return torch.clamp(x, min=0) - x * z + torch.log1p(torch.exp(-torch.abs(x)))


def exponential_decay(lr, step, decay_steps, decay_rate):
    """staircase=True; step is the value before the update."""
    return lr * (decay_rate ** (step // decay_steps))


def tf_round(x):
    return torch.round(x)  # round-half-to-even, as tf.round


# ----------------------------------------------------------------------------
# Model graphs.  P: OrderedDict name -> tensor (names follow the reference's
# variable scopes).  cfg: SimpleNamespace.  masks: dict name -> keep mask or {}.
# ----------------------------------------------------------------------------
def _tables(P, F):
    return [P['embedding/embed_cate_w_%d' % i] for i in range(F)]


def _embed(P, cfg, X_cate, literal=False):
    tabs = _tables(P, cfg.cate_num)
    if cfg.mode == 'R':
        return embed_block_rmode_literal(tabs, X_cate) if literal else embed_block_rmode(tabs, X_cate)
    return embed_block_smode(tabs, X_cate)


def _tower(x, P, prefix, n, rate, masks, mask_prefix):
    for i in range(n):
        x = torch.relu(x @ P['%s_w_%d' % (prefix, i)] + P['%s_b_%d' % (prefix, i)])
        x = dropout(x, rate, masks.get('%s_%d' % (mask_prefix, i)))
    return x


def deepfm_forward(P, cfg, X_cate, X_cont, masks=None, literal=False):
    masks = masks or {}
    Z = _embed(P, cfg, X_cate, literal)
    B = Z.shape[0]
    n = len(cfg.deep_layers)
    first = dropout(Z.sum(2), cfg.dropout_fm, masks.get('fm_first'))
    s = Z.sum(1)
    second = 0.5 * (s * s - (Z * Z).sum(1))
    second = dropout(second, cfg.dropout_fm, masks.get('fm_second'))
    x0 = torch.cat([Z.reshape(B, -1), X_cont], dim=1)
    deep = _tower(x0, P, 'deep_part/deep', n, cfg.dropout_deep, masks, 'deep')
    concat = torch.cat([first, second, deep], dim=1)
    logit = concat @ P['output_unit/output_w_%d' % (n - 1)] + P['output_unit/output_b_%d' % (n - 1)]
    y_out = torch.sigmoid(logit).reshape(-1)
    return SimpleNamespace(Z=Z, first=first, second=second, x0=x0, deep=deep, concat=concat,
                           logit=logit.reshape(-1), y_out=y_out)


def dcn_cross_literal(x0, ws, bs):
    """DCN/DCN.py:57-65 literally: tensordot(matmul(x0, xi^T)[B,D,D], w[D,1], 1) + b + xi."""
    c0 = x0.unsqueeze(2)
    xi = c0
    for w, b in zip(ws, bs):
        outer = torch.matmul(c0, xi.transpose(1, 2))          # [B,D,D]
        xi = torch.tensordot(outer, w, dims=1) + b + xi
    return xi.squeeze(2)


def dcn_cross(x0, ws, bs):
    """Same math in O(B*D): x_{l+1} = x0 * (x_l . w_l) + b_l + x_l."""
    xi = x0
    for w, b in zip(ws, bs):
        s = xi @ w                                            # [B,1]
        xi = x0 * s + b.view(1, -1) + xi
    return xi


def dcn_forward(P, cfg, X_cate, X_cont, masks=None, literal=False):
    masks = masks or {}
    Z = _embed(P, cfg, X_cate, literal)
    B = Z.shape[0]
    n = len(cfg.deep_layers)
    x0 = torch.cat([Z.reshape(B, -1), X_cont], dim=1)
    ws = [P['cross_part/cross_w_%d' % l] for l in range(cfg.cross_layer_num)]
    bs = [P['cross_part/cross_b_%d' % l] for l in range(cfg.cross_layer_num)]
    cross = dcn_cross_literal(x0, ws, bs) if literal else dcn_cross(x0, ws, bs)
    deep = _tower(x0, P, 'deep_part/deep', n, cfg.dropout, masks, 'deep')
    stack = torch.cat([cross, deep], dim=1)
    logit = stack @ P['stack_layer/stack_w_%d' % (n - 1)] + P['stack_layer/stack_b_%d' % (n - 1)]
    y_out = torch.sigmoid(logit).reshape(-1)
    return SimpleNamespace(Z=Z, x0=x0, cross=cross, deep=deep, logit=logit.reshape(-1), y_out=y_out)


def nfm_forward(P, cfg, X_cate, X_cont, masks=None, literal=False):
    masks = masks or {}
    Z = _embed(P, cfg, X_cate, literal)
    n = len(cfg.deep_layers)
    s = Z.sum(1)
    bi = 0.5 * (s * s - (Z * Z).sum(1))
    bi = dropout(bi, cfg.dropout_bi, masks.get('bi'))
    x0 = torch.cat([bi, X_cont], dim=1)
    deep = _tower(x0, P, 'deep/deep', n, cfg.dropout_deep, masks, 'deep')
    logit = deep @ P['output/output_w_%d' % (n - 1)] + P['output/output_b_%d' % (n - 1)]
    y_out = torch.sigmoid(logit).reshape(-1)
    return SimpleNamespace(Z=Z, bi=bi, x0=x0, deep=deep, logit=logit.reshape(-1), y_out=y_out)


def _embed_with_cont(P, cfg, X_cate, X_cont, literal):
    Z = _embed(P, cfg, X_cate, literal)
    Wc = P['embedding/embed_cont_w_%d' % (cfg.cate_num - 1)]
    return torch.cat([Z, embed_cont(Wc, X_cont)], dim=1)


def pnn_forward(P, cfg, X_cate, X_cont, masks=None, literal=False):
    masks = masks or {}
    Z = _embed_with_cont(P, cfg, X_cate, X_cont, literal)      # [B,N,E]
    lz_w = P['product/lz_w']                                   # [Pn,N,E]
    if literal:
        lz = torch.cat([(Z * lz_w[i]).sum(dim=(1, 2)).view(-1, 1) for i in range(cfg.product_size)], dim=1)
    else:
        lz = torch.einsum('bne,pne->bp', Z, lz_w)
    if cfg.inner_product:
        w = P['product/lp_inner_w']                            # [Pn,N]
        if literal:
            lp = torch.cat([torch.linalg.norm((Z * w[i].view(1, -1, 1)).sum(1)).view(1, 1)
                            for i in range(cfg.product_size)], dim=1)   # [1,Pn] batch-global norm
        elif getattr(cfg, 'inner_mode', 'batch_global') == 'per_sample':
            u = torch.einsum('bne,pn->bpe', Z, w)          # canonical IPNN option (not the reference's)
            lp = torch.sqrt((u * u).sum(dim=2))
        else:
            u = torch.einsum('bne,pn->pbe', Z, w)
            lp = torch.sqrt((u * u).sum(dim=(1, 2))).view(1, -1)
    else:
        s = Z.sum(1)
        w = P['product/lp_outer_w']                            # [Pn,E,E]
        lp = torch.einsum('be,bf,pef->bp', s, s, w)
    prod = torch.relu(lz + lp + P['product/product_bias'])
    prod = dropout(prod, cfg.dropout, masks.get('product'))
    x = _tower(prod, P, 'hidden/hidden', len(cfg.hidden_layers), cfg.dropout, masks, 'hidden')
    logit = x @ P['hidden/output_w'] + P['hidden/output_b']
    y_out = torch.sigmoid(logit).reshape(-1)
    return SimpleNamespace(Z=Z, lz=lz, lp=lp, prod=prod, hidden=x, logit=logit.reshape(-1), y_out=y_out)


def afm_forward(P, cfg, X_cate, X_cont, masks=None, literal=False):
    masks = masks or {}
    Z = _embed_with_cont(P, cfg, X_cate, X_cont, literal)      # [B,N,E]
    B, N, E = Z.shape
    ii, jj = torch.triu_indices(N, N, offset=1)                # lexicographic (i<j)
    pw = Z[:, ii, :] * Z[:, jj, :]                             # [B,Pn,E]
    att = torch.relu(pw.reshape(-1, E) @ P['attention_based_pooling/attention_w']
                     + P['attention_based_pooling/attention_b']).reshape(B, -1, cfg.attention_size)
    e = (att * P['attention_based_pooling/attention_h']).sum(2, keepdim=True)   # [B,Pn,1]
    if getattr(cfg, 'canonical_softmax', False):
        a = torch.softmax(e, dim=1)
    else:
        a = torch.softmax(e, dim=-1)                           # size-1 axis: all ones (AFM/AFM.py:79)
    a = dropout(a, cfg.dropout, masks.get('att'))
    x = (a * pw).sum(1)
    x = dropout(x, cfg.dropout, masks.get('out'))
    y_out = x.sum(1)
    return SimpleNamespace(Z=Z, pw=pw, a=a, x=x, logit=y_out, y_out=y_out)


def esmm_forward(P, cfg, X_cate, X_cont, masks=None, literal=False):
    masks = masks or {}
    Z = _embed(P, cfg, X_cate, literal)
    z = Z.sum(2)
    inp = torch.cat([z, X_cont], dim=1)
    n = len(cfg.deep_layers)
    ctr = _tower(inp, P, 'ctr_deep_part/ctr_deep', n, cfg.dropout, masks, 'ctr')
    ctr_out = ctr @ P['ctr_deep_part/ctr_output_w'] + P['ctr_deep_part/ctr_output_b']
    cvr = _tower(inp, P, 'cvr_deep_part/cvr_deep', n, cfg.dropout, masks, 'cvr')
    cvr_out = cvr @ P['cvr_deep_part/cvr_output_w'] + P['cvr_deep_part/cvr_output_b']
    y_out = (torch.sigmoid(ctr_out) * torch.sigmoid(cvr_out)).reshape(-1)
    return SimpleNamespace(Z=Z, inp=inp, ctr_logit=ctr_out.reshape(-1), cvr_logit=cvr_out.reshape(-1),
                           logit=y_out, y_out=y_out)


# ---- DIN (DIN/DIN.py) --------------------------------------------------------
_BN_EPS = 1e-3


def bn_infer(x, gamma, beta, mean, var):
    return gamma * (x - mean) / torch.sqrt(var + _BN_EPS) + beta


def dice(x, alpha):
    """DIN/DIN.py:149-166: BN(center=False, scale=False) in inference mode with
    moving stats 0/1, so x_p = sigmoid(x / sqrt(1 + 1e-3))."""
    xp = torch.sigmoid(x / math.sqrt(1.0 + _BN_EPS))
    return alpha * (1.0 - xp) * x + xp * x


def din_attention(P, q, K, sl):
    """DIN/DIN.py:169-196."""
    B, T, H = K.shape
    qt = q.unsqueeze(1).expand(B, T, H)
    din_all = torch.cat([qt, K, qt - K, qt * K], dim=-1)
    d1 = torch.sigmoid(din_all @ P['embedding/f1_att/kernel'] + P['embedding/f1_att/bias'])
    d2 = torch.sigmoid(d1 @ P['embedding/f2_att/kernel'] + P['embedding/f2_att/bias'])
    d3 = d2 @ P['embedding/f3_att/kernel'] + P['embedding/f3_att/bias']       # [B,T,1]
    out = d3.reshape(B, 1, T)
    mask = (torch.arange(T).view(1, 1, T) < sl.view(B, 1, 1))
    out = torch.where(mask, out, torch.full_like(out, float(-2 ** 32 + 1)))
    out = out / (H ** 0.5)
    out = torch.softmax(out, dim=-1)
    return torch.matmul(out, K)                                                # [B,1,H]


def din_embed(P, cate_list, ids):
    ic = cate_list[ids]
    return torch.cat([P['embedding/item_emb_w'][ids], P['embedding/cate_emb_w'][ic]], dim=-1)


def _din_fcn(P, u_emb, x_emb, tag):
    d = torch.cat([u_emb, x_emb], dim=-1)
    d = bn_infer(d, P['fcn/b1/gamma'], P['fcn/b1/beta'], P['fcn/b1/moving_mean'], P['fcn/b1/moving_variance'])
    d1 = dice(d @ P['fcn/f1/kernel'] + P['fcn/f1/bias'], P['fcn/alphadice_1_%s' % tag])
    d2 = dice(d1 @ P['fcn/f2/kernel'] + P['fcn/f2/bias'], P['fcn/alphadice_2_%s' % tag])
    return (d2 @ P['fcn/f3/kernel'] + P['fcn/f3/bias']).reshape(-1)


def din_forward(P, cfg, cate_list, i, hist_i, sl, j=None):
    H = cfg.hidden_units
    i = i.to(torch.int64)
    hist_i = hist_i.to(torch.int64)
    i_emb = din_embed(P, cate_list, i)
    i_b = P['embedding/item_b'][i]
    h_emb = din_embed(P, cate_list, hist_i)
    hist = din_attention(P, i_emb, h_emb, sl)
    hist = bn_infer(hist, P['embedding/batch_normalization/gamma'], P['embedding/batch_normalization/beta'],
                    P['embedding/batch_normalization/moving_mean'],
                    P['embedding/batch_normalization/moving_variance'])
    hist = hist.reshape(-1, H)
    u_emb = hist @ P['embedding/dense/kernel'] + P['embedding/dense/bias']
    d3_i = _din_fcn(P, u_emb, i_emb, 'i')
    logits = i_b + d3_i
    out = SimpleNamespace(i_emb=i_emb, h_emb=h_emb, u_emb=u_emb, logit=logits, y_out=torch.sigmoid(logits))
    if j is not None:
        j = j.to(torch.int64)
        j_emb = din_embed(P, cate_list, j)
        j_b = P['embedding/item_b'][j]
        d3_j = _din_fcn(P, u_emb, j_emb, 'j')
        x = i_b - j_b + d3_i - d3_j
        out.mf_auc = (x > 0).to(logits.dtype).mean()
        out.p_and_n = torch.stack([torch.sigmoid(i_b + d3_i), torch.sigmoid(j_b + d3_j)], dim=-1)
    return out


def din_logits_all(P, cfg, cate_list, i, hist_i, sl):
    """logits_all (DIN/DIN.py:73-89): every user row against every item through the FCN with the `_all` Dice
    alphas -> [B, item_count].  u_emb needs the target items i (the attention query, DIN/DIN.py:44)."""
    u_emb = din_forward(P, cfg, cate_list, i, hist_i, sl).u_emb
    B, I = u_emb.shape[0], P['embedding/item_emb_w'].shape[0]
    all_emb = torch.cat([P['embedding/item_emb_w'], P['embedding/cate_emb_w'][cate_list]], dim=1)
    u_all = u_emb.unsqueeze(1).expand(B, I, u_emb.shape[1])
    e_all = all_emb.unsqueeze(0).expand(B, I, all_emb.shape[1])
    d3 = _din_fcn(P, u_all, e_all, 'all').reshape(B, I)
    return torch.sigmoid(P['embedding/item_b'] + d3)


FORWARD = dict(DeepFM=deepfm_forward, DCN=dcn_forward, NFM=nfm_forward, PNN=pnn_forward,
               AFM=afm_forward, ESMM=esmm_forward)
# loss kind per model: 'log_loss' on y_out, or sigmoid CE applied to y_out (a probability)
LOSS_KIND = dict(DeepFM='log_loss', DCN='log_loss', NFM='log_loss', AFM='log_loss',
                 PNN='sce_on_prob', ESMM='sce_on_prob', DIN='sce_on_logit')
OPTIMIZER = dict(DeepFM='adam', DCN='adam', PNN='adam', NFM='adagrad', AFM='adagrad',
                 ESMM='adagrad', DIN='sgd')


def model_loss(model, out, y):
    kind = LOSS_KIND[model]
    if kind == 'log_loss':
        return log_loss(y, out.y_out)
    if kind == 'sce_on_prob':
        return sigmoid_ce_with_logits(y, out.y_out).mean()
    return sigmoid_ce_with_logits(y, out.logit).mean()


# ----------------------------------------------------------------------------
# Parameter construction (names/shapes as the reference's variable scopes)
# ----------------------------------------------------------------------------
def truncated_normal(rng, shape, std=0.01):
    """tf.truncated_normal_initializer(0, std): re-sample outside +-2 sigma."""
    x = rng.standard_normal(size=shape)
    bad = np.abs(x) > 2.0
    while bad.any():
        x[bad] = rng.standard_normal(size=int(bad.sum()))
        bad = np.abs(x) > 2.0
    return (x * std).astype(np.float64)


def n_rows(cfg):
    """Number of embedding 'field rows' N entering the interaction layer."""
    base = int(np.sum(cfg.cate_list)) if cfg.mode == 'R' else cfg.cate_num
    return base


def param_shapes(model, cfg):
    F, C, E = cfg.cate_num, cfg.cont_num, cfg.embed_size
    sh = OrderedDict()
    for i in range(F):
        sh['embedding/embed_cate_w_%d' % i] = (int(cfg.cate_list[i]), E)
    N = n_rows(cfg)
    if model in ('AFM', 'PNN'):
        sh['embedding/embed_cont_w_%d' % (F - 1)] = (C, E)
        N = N + C

    def tower(prefix, d_in, layers):
        for i, L in enumerate(layers):
            sh['%s_w_%d' % (prefix, i)] = (d_in, L)
            sh['%s_b_%d' % (prefix, i)] = (1, L)
            d_in = L
        return d_in

    if model == 'DeepFM':
        n = len(cfg.deep_layers)
        last = tower('deep_part/deep', N * E + C, cfg.deep_layers)
        sh['output_unit/output_w_%d' % (n - 1)] = (N + E + last, 1)
        sh['output_unit/output_b_%d' % (n - 1)] = (1,)
    elif model == 'DCN':
        n = len(cfg.deep_layers)
        D = N * E + C
        for l in range(cfg.cross_layer_num):
            sh['cross_part/cross_w_%d' % l] = (D, 1)
            sh['cross_part/cross_b_%d' % l] = (D, 1)
        last = tower('deep_part/deep', D, cfg.deep_layers)
        sh['stack_layer/stack_w_%d' % (n - 1)] = (D + last, 1)
        sh['stack_layer/stack_b_%d' % (n - 1)] = (1,)
    elif model == 'NFM':
        n = len(cfg.deep_layers)
        last = tower('deep/deep', E + C, cfg.deep_layers)
        sh['output/output_w_%d' % (n - 1)] = (last, 1)
        sh['output/output_b_%d' % (n - 1)] = (1,)
    elif model == 'PNN':
        Pn = cfg.product_size
        sh['product/lz_w'] = (Pn, N, E)
        if cfg.inner_product:
            sh['product/lp_inner_w'] = (Pn, N)
        else:
            sh['product/lp_outer_w'] = (Pn, E, E)
        sh['product/product_bias'] = (Pn,)
        last = tower('hidden/hidden', Pn, cfg.hidden_layers)
        sh['hidden/output_w'] = (last, 1)
        sh['hidden/output_b'] = (1,)
    elif model == 'AFM':
        A = cfg.attention_size
        sh['attention_based_pooling/attention_w'] = (E, A)
        sh['attention_based_pooling/attention_b'] = (1, A)
        sh['attention_based_pooling/attention_h'] = (1, A)
    elif model == 'ESMM':
        for t in ('ctr', 'cvr'):
            last = tower('%s_deep_part/%s_deep' % (t, t), N + C, cfg.deep_layers)
            sh['%s_deep_part/%s_output_w' % (t, t)] = (last, 1)
            sh['%s_deep_part/%s_output_b' % (t, t)] = (1,)
    else:
        raise ValueError(model)
    return sh


def init_params(model, cfg, seed=0, std=0.01):
    """Every weight AND bias is truncated_normal(0, 0.01) (DeepFM/DeepFM.py:30,47-48,69-70)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return OrderedDict((k, truncated_normal(rng, s, std)) for k, s in param_shapes(model, cfg).items())


def din_param_shapes(cfg, user_count, item_count, cate_count):
    H = cfg.hidden_units
    sh = OrderedDict()
    sh['embedding/item_emb_w'] = (item_count, H // 2)
    sh['embedding/item_b'] = (item_count,)
    sh['embedding/cate_emb_w'] = (cate_count, H // 2)
    for name, (i, o) in (('f1_att', (4 * H, 80)), ('f2_att', (80, 40)), ('f3_att', (40, 1))):
        sh['embedding/%s/kernel' % name] = (i, o)
        sh['embedding/%s/bias' % name] = (o,)
    for k in ('gamma', 'beta', 'moving_mean', 'moving_variance'):
        sh['embedding/batch_normalization/%s' % k] = (H,)
    sh['embedding/dense/kernel'] = (H, H)
    sh['embedding/dense/bias'] = (H,)
    for k in ('gamma', 'beta', 'moving_mean', 'moving_variance'):
        sh['fcn/b1/%s' % k] = (2 * H,)
    for name, (i, o) in (('f1', (2 * H, 80)), ('f2', (80, 40)), ('f3', (40, 1))):
        sh['fcn/%s/kernel' % name] = (i, o)
        sh['fcn/%s/bias' % name] = (o,)
    for tag in ('i', 'j', 'all'):
        sh['fcn/alphadice_1_%s' % tag] = (80,)
        sh['fcn/alphadice_2_%s' % tag] = (40,)
    return sh


DIN_NON_TRAINABLE = ('moving_mean', 'moving_variance')


def din_init_params(cfg, user_count, item_count, cate_count, seed=0):
    """tf.get_variable / tf.layers.dense defaults: Glorot-uniform kernels, zero biases,
    BN gamma=1 beta=0 mean=0 var=1, dice alphas 0, item_b 0 (DIN/DIN.py:18-21,150)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    P = OrderedDict()
    for k, s in din_param_shapes(cfg, user_count, item_count, cate_count).items():
        if k.endswith('gamma') or k.endswith('moving_variance'):
            P[k] = np.ones(s)
        elif (k.endswith('bias') or k.endswith('beta') or k.endswith('moving_mean')
              or 'alphadice' in k or k.endswith('item_b')):
            P[k] = np.zeros(s)
        else:
            lim = math.sqrt(6.0 / (s[0] + s[1]))
            P[k] = rng.uniform(-lim, lim, size=s)
    return P


# ----------------------------------------------------------------------------
# TF-1.x optimizers (dense-equivalent semantics)
# ----------------------------------------------------------------------------
class AdamTF1:
    """tf.train.AdamOptimizer: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); eps outside the sqrt.
    Sparse gradients are summed over duplicates and m, v decay for all rows, so the
    result equals dense Adam with a zero-filled gradient."""

    def __init__(self, params, b1=0.9, b2=0.999, eps=1e-8):
        self.b1, self.b2, self.eps, self.t = b1, b2, eps, 0
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}

    def apply(self, params, grads, lr):
        self.t += 1
        lr_t = lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        for k, g in grads.items():
            self.m[k] = self.b1 * self.m[k] + (1.0 - self.b1) * g
            self.v[k] = self.b2 * self.v[k] + (1.0 - self.b2) * g * g
            params[k] = params[k] - lr_t * self.m[k] / (torch.sqrt(self.v[k]) + self.eps)


class LazyAdamTF1(AdamTF1):
    """Row-wise ("lazy") Adam for embedding tables: the TF-1.x Adam formulas above (same bias-corrected
    step size lr_t of the GLOBAL update count t, eps outside the sqrt), applied only to the table rows a
    batch looks up (in range); every other row keeps its weight, m and v bit for bit.  This is the sparse
    update of tf.contrib.opt.LazyAdamOptimizer, NOT what the reference's tf.train.AdamOptimizer does
    (DeepFM/DeepFM.py:32,100 decays m, v of every row: AdamTF1) - it is the form that exists at Criteo scale,
    where a dense pass over the tables is 13 GB per step.  `rows[k]` (bool per row) marks the looked-up rows
    of table k; variables without an entry in `rows` get the dense update."""

    def apply(self, params, grads, lr, rows=None):
        self.t += 1
        lr_t = lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        for k, g in grads.items():
            m = self.b1 * self.m[k] + (1.0 - self.b1) * g
            v = self.b2 * self.v[k] + (1.0 - self.b2) * g * g
            w = params[k] - lr_t * m / (torch.sqrt(v) + self.eps)
            if rows is not None and k in rows:
                sel = rows[k].view(-1, *([1] * (g.dim() - 1)))
                m, v, w = torch.where(sel, m, self.m[k]), torch.where(sel, v, self.v[k]), torch.where(sel, w, params[k])
            self.m[k], self.v[k], params[k] = m, v, w


class AdagradTF1:
    """tf.train.AdagradOptimizer: accumulator starts at 0.1; no epsilon."""

    def __init__(self, params, init_acc=0.1):
        self.acc = {k: torch.full_like(v, init_acc) for k, v in params.items()}

    def apply(self, params, grads, lr):
        for k, g in grads.items():
            self.acc[k] = self.acc[k] + g * g
            params[k] = params[k] - lr * g / torch.sqrt(self.acc[k])


class SGD:
    def __init__(self, params):
        pass

    def apply(self, params, grads, lr):
        for k, g in grads.items():
            params[k] = params[k] - lr * g


class FtrlTF1:
    """tf.train.FtrlOptimizer(lr, l1) with its defaults lr_power=-0.5, initial accumulator 0.1, l2=0
    (Wide_Deep/Wide_Deep.py:107-108).  `rows[k]` (bool per row) restricts the update to the rows an
    embedding_lookup touched: the gradient of a lookup is sparse and TF applies it row-wise, so an
    untouched row keeps its weight even where that weight is not the closed form of (z, n)."""

    def __init__(self, params, l1=0.1, l2=0.0, init_acc=0.1):
        self.l1, self.l2 = l1, l2
        self.n = {k: torch.full_like(v, init_acc) for k, v in params.items()}
        self.z = {k: torch.zeros_like(v) for k, v in params.items()}

    def apply(self, params, grads, lr, rows=None):
        for k, g in grads.items():
            w, n, z = params[k], self.n[k], self.z[k]
            nn = n + g * g
            zz = z + g - (torch.sqrt(nn) - torch.sqrt(n)) / lr * w
            ww = torch.where(zz.abs() <= self.l1, torch.zeros_like(zz),
                             -(zz - torch.sign(zz) * self.l1) / (torch.sqrt(nn) / lr + 2.0 * self.l2))
            if rows is not None and k in rows:
                m = rows[k].view(-1, 1)
                ww, nn, zz = torch.where(m, ww, w), torch.where(m, nn, n), torch.where(m, zz, z)
            params[k], self.n[k], self.z[k] = ww, nn, zz


# ----------------------------------------------------------------------------
# Wide & Deep (Wide_Deep/Wide_Deep.py:34-118): standard lookups; wide = concat of E-wide rows of the
# cross and category tables -> one linear unit; deep = category embeddings + dense -> relu tower.
# ----------------------------------------------------------------------------
def wide_deep_param_shapes(cfg):
    E = cfg.embed_size
    sh = OrderedDict()
    for i, v in enumerate(cfg.cross_list):
        sh['wide_net/wide_cross_w_%d' % i] = (int(v), E)
        sh['wide_net/wide_cross_b_%d' % i] = (E,)           # declared, never used (Wide_Deep/Wide_Deep.py:49-50)
    for i, v in enumerate(cfg.cate_list):
        sh['wide_net/wide_cate_w_%d' % i] = (int(v), E)
    sh['wide_net/wide_output_w'] = ((len(cfg.cross_list) + len(cfg.cate_list)) * E, 1)
    sh['wide_net/wide_output_b'] = (1,)
    for i, v in enumerate(cfg.cate_list):
        sh['deep_net/deep_cate_w_%d' % i] = (int(v), E)
    d_in = len(cfg.cate_list) * E + cfg.cont_num
    for i, L in enumerate(cfg.hidden_layers):
        sh['deep_net/deep_hidden_w_%d' % i] = (d_in, L)
        sh['deep_net/deep_hidden_b_%d' % i] = (1, L)
        d_in = L
    sh['deep_net/deep_output_w'] = (d_in, 1)
    sh['deep_net/deep_output_b'] = (1,)
    return sh


def wide_deep_init_params(cfg, seed=0, std=0.01):
    rng = np.random.Generator(np.random.PCG64(seed))
    return OrderedDict((k, truncated_normal(rng, s, std)) for k, s in wide_deep_param_shapes(cfg).items())


def wide_deep_forward(P, cfg, X_cross, X_cate, X_cont, masks=None):
    masks = masks or {}
    wide = [P['wide_net/wide_cross_w_%d' % i] for i in range(len(cfg.cross_list))] + \
           [P['wide_net/wide_cate_w_%d' % i] for i in range(len(cfg.cate_list))]
    Zw = embed_block_smode(wide, torch.cat([X_cross, X_cate], dim=1))            # [B, Fw, E]
    wide_out = Zw.reshape(Zw.shape[0], -1) @ P['wide_net/wide_output_w'] + P['wide_net/wide_output_b']
    Zd = embed_block_smode([P['deep_net/deep_cate_w_%d' % i] for i in range(len(cfg.cate_list))], X_cate)
    x = torch.cat([Zd.reshape(Zd.shape[0], -1), X_cont], dim=1)
    n = len(cfg.hidden_layers)
    for i in range(n):
        x = torch.relu(x @ P['deep_net/deep_hidden_w_%d' % i] + P['deep_net/deep_hidden_b_%d' % i])
        x = dropout(x, cfg.dropout, masks.get('deep_%d' % i))
    deep_out = x @ P['deep_net/deep_output_w'] + P['deep_net/deep_output_b']
    logit = (wide_out + deep_out).reshape(-1)
    return SimpleNamespace(logit=logit, y_out=torch.sigmoid(logit), Zw=Zw, Zd=Zd)


class OracleWideDeep:
    """Wide & Deep graph + its two optimizers (FTRL with l1=0.1 on wide_net, Adagrad on deep_net;
    Wide_Deep/Wide_Deep.py:98-118).  The loss is sigmoid cross-entropy applied to y_out, which is already
    a probability (as written).  Both minimize() calls advance the one global_step, so it grows by two
    per step; the learning rate of a step is taken at its value before the step."""

    def __init__(self, cfg, params, dtype=torch.float64):
        self.cfg, self.dtype = cfg, dtype
        self.P = OrderedDict((k, torch.as_tensor(np.asarray(v), dtype=dtype).clone()) for k, v in params.items())
        self.wide = [k for k in self.P if k.startswith('wide_net/')]
        self.deep = [k for k in self.P if k.startswith('deep_net/')]
        self.ftrl = FtrlTF1({k: self.P[k] for k in self.wide})
        self.adagrad = AdagradTF1({k: self.P[k] for k in self.deep})
        self.global_step = 0

    def _inputs(self, X_cross, X_cate, X_cont, y):
        i64 = lambda a: torch.as_tensor(np.asarray(a).astype(np.int32).astype(np.int64))
        return (i64(X_cross), i64(X_cate), torch.as_tensor(np.asarray(X_cont).astype(np.float32)).to(self.dtype),
                torch.as_tensor(np.asarray(y).astype(np.float32)).to(self.dtype))

    def forward(self, X_cross, X_cate, X_cont):
        xr, xc, xf, _ = self._inputs(X_cross, X_cate, X_cont, np.zeros(len(X_cate)))
        with torch.no_grad():
            return wide_deep_forward(self.P, self.cfg, xr, xc, xf)

    def loss_and_grads(self, X_cross, X_cate, X_cont, y, masks=None):
        xr, xc, xf, yy = self._inputs(X_cross, X_cate, X_cont, y)
        P = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in self.P.items())
        out = wide_deep_forward(P, self.cfg, xr, xc, xf, masks)
        loss = sigmoid_ce_with_logits(yy, out.y_out).mean()
        grads = torch.autograd.grad(loss, list(P.values()), allow_unused=True)
        g = OrderedDict((k, torch.zeros_like(v) if gi is None else gi.detach()) for (k, v), gi in zip(P.items(), grads))
        return loss.detach(), g, out

    def _touched(self, X_cross, X_cate):
        """rows of every wide table that the batch looks up (in range)."""
        rows = {}
        xr = np.asarray(X_cross).astype(np.int64)
        xc = np.asarray(X_cate).astype(np.int64)
        for name, ids, cards in (('wide_net/wide_cross_w_%d', xr, self.cfg.cross_list),
                                 ('wide_net/wide_cate_w_%d', xc, self.cfg.cate_list)):
            for i, v in enumerate(cards):
                m = torch.zeros(int(v), dtype=torch.bool)
                col = ids[:, i]
                m[torch.as_tensor(col[(col >= 0) & (col < v)])] = True
                rows[name % i] = m
        return rows

    def run(self, X_cross, X_cate, X_cont, y, is_train=False, masks=None):
        if is_train:
            loss, g, _ = self.loss_and_grads(X_cross, X_cate, X_cont, y, masks)
            c = self.cfg
            lr = exponential_decay(c.learning_rate, self.global_step, c.decay_steps, c.decay_rate)
            pw = {k: self.P[k] for k in self.wide if not k.startswith('wide_net/wide_cross_b_')}
            self.ftrl.apply(pw, {k: g[k] for k in pw}, lr, rows=self._touched(X_cross, X_cate))
            pd_ = {k: self.P[k] for k in self.deep}
            self.adagrad.apply(pd_, {k: g[k] for k in pd_}, lr)
            self.P.update(pw)
            self.P.update(pd_)
            self.global_step += 2
            return np.float32(loss.item())
        out = self.forward(X_cross, X_cate, X_cont)
        return np.asarray(y, dtype=np.float32), tf_round(out.y_out).to(torch.float32).numpy()


# ----------------------------------------------------------------------------
# NCF (NCF/NCF.py:29-120): tf.layers.dense over one-hot ids = row lookup + bias, with relu; GMF
# product; three relu+dropout layers on the MLP embeddings; dense(1) WITH relu; sigmoid; and
# sigmoid cross-entropy applied to that probability.  Adam (dense kernels: every row decays).
# ----------------------------------------------------------------------------
def ncf_param_shapes(cfg):
    E = cfg.embed_size
    sh = OrderedDict()
    for n, v in (('user_embed_GMF', cfg.user_size), ('item_embed_GMF', cfg.item_size),
                 ('user_embed_MLP', cfg.user_size), ('item_embed_MLP', cfg.item_size)):
        sh['embedding/%s/kernel' % n] = (int(v), E)
        sh['embedding/%s/bias' % n] = (E,)
    for i, (a, b) in enumerate(((2 * E, 2 * E), (2 * E, E), (E, E // 2))):
        sh['MLP/MLP_layer%d/kernel' % (i + 1)] = (a, b)
        sh['MLP/MLP_layer%d/bias' % (i + 1)] = (b,)
    sh['concat/logits/kernel'] = (E + E // 2, 1)
    sh['concat/logits/bias'] = (1,)
    return sh


def ncf_init_params(cfg, seed=0, std=0.01):
    """kernels ~ truncated normal; tf.layers.dense biases start at zero."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return OrderedDict((k, truncated_normal(rng, s, std) if k.endswith('kernel') else np.zeros(s))
                       for k, s in ncf_param_shapes(cfg).items())


def _onehot_dense(P, name, ids, size):
    """relu(one_hot(ids, size) @ kernel + bias): an id outside [0, size) gives the all-zero one-hot row."""
    ok = (ids >= 0) & (ids < size)
    rows = P['embedding/%s/kernel' % name][torch.where(ok, ids, torch.zeros_like(ids))] * ok.to(P['embedding/%s/kernel' % name].dtype).view(-1, 1)
    return torch.relu(rows + P['embedding/%s/bias' % name])


def ncf_forward(P, cfg, X_user, X_item, masks=None):
    masks = masks or {}
    uG = _onehot_dense(P, 'user_embed_GMF', X_user, cfg.user_size)
    iG = _onehot_dense(P, 'item_embed_GMF', X_item, cfg.item_size)
    uM = _onehot_dense(P, 'user_embed_MLP', X_user, cfg.user_size)
    iM = _onehot_dense(P, 'item_embed_MLP', X_item, cfg.item_size)
    x = torch.cat([uM, iM], dim=1)
    for i in (1, 2, 3):
        x = torch.relu(x @ P['MLP/MLP_layer%d/kernel' % i] + P['MLP/MLP_layer%d/bias' % i])
        x = dropout(x, cfg.dropout, masks.get('mlp_%d' % i))
    concat = torch.cat([uG * iG, x], dim=1)
    logits = torch.relu(concat @ P['concat/logits/kernel'] + P['concat/logits/bias']).reshape(-1)
    return SimpleNamespace(logit=logits, y_out=torch.sigmoid(logits))


class OracleNCF:
    def __init__(self, cfg, params, dtype=torch.float64):
        self.cfg, self.dtype = cfg, dtype
        self.P = OrderedDict((k, torch.as_tensor(np.asarray(v), dtype=dtype).clone()) for k, v in params.items())
        self.opt = AdamTF1(self.P)
        self.global_step = 0

    def _ids(self, X):
        X = np.asarray(X).astype(np.int32).astype(np.int64)
        return torch.as_tensor(X[:, 0]), torch.as_tensor(X[:, 1])

    def forward(self, X):
        u, i = self._ids(X)
        with torch.no_grad():
            return ncf_forward(self.P, self.cfg, u, i)

    def loss_and_grads(self, X, y, masks=None):
        u, i = self._ids(X)
        yy = torch.as_tensor(np.asarray(y).astype(np.float32)).to(self.dtype)
        P = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in self.P.items())
        out = ncf_forward(P, self.cfg, u, i, masks)
        loss = sigmoid_ce_with_logits(yy, out.y_out).mean()
        grads = torch.autograd.grad(loss, list(P.values()), allow_unused=True)
        g = OrderedDict((k, torch.zeros_like(v) if gi is None else gi.detach()) for (k, v), gi in zip(P.items(), grads))
        return loss.detach(), g, out

    def run(self, X, y, is_train=False, masks=None):
        if is_train:
            loss, g, _ = self.loss_and_grads(X, y, masks)
            c = self.cfg
            self.opt.apply(self.P, g, exponential_decay(c.learning_rate, self.global_step, c.decay_steps, c.decay_rate))
            self.global_step += 1
            return np.float32(loss.item())
        out = self.forward(X)
        return np.asarray(y, dtype=np.float32), tf_round(out.y_out).to(torch.float32).numpy()


_OPT = dict(adam=AdamTF1, adagrad=AdagradTF1, sgd=SGD)


class OracleModel:
    """One of the six CSV-family graphs + its TF-1.x optimizer, on the CPU.

    run(X_cate, X_cont, y, is_train) mirrors <Model>.run (DeepFM/DeepFM.py:114-131):
    training returns the loss computed with the pre-update weights and applies one
    optimizer step; evaluation returns (actual, round(y_out))."""

    def __init__(self, model, cfg, params, dtype=torch.float64, literal=False):
        self.model, self.cfg, self.dtype, self.literal = model, cfg, dtype, literal
        self.P = OrderedDict((k, torch.as_tensor(np.asarray(v), dtype=dtype).clone()) for k, v in params.items())
        # cfg.adam_mode == 'lazy' (S semantics only; not a reference flag): row-wise Adam on the tables
        self.lazy = OPTIMIZER[model] == 'adam' and getattr(cfg, 'adam_mode', 'tf1') == 'lazy'
        if self.lazy and cfg.mode != 'S':
            raise ValueError("adam_mode 'lazy' is defined for the standard lookup (mode 'S')")
        self.opt = (LazyAdamTF1 if self.lazy else _OPT[OPTIMIZER[model]])(self.P)
        self.global_step = 0

    def touched_rows(self, X_cate):
        """name -> bool[V_f]: the rows of every table the batch looks up with an in-range id."""
        bags = self._bags(X_cate)
        xc = None if bags is not None else np.asarray(X_cate).astype(np.int32).astype(np.int64)
        rows = {}
        for f, v in enumerate(self.cfg.cate_list):
            m = torch.zeros(int(v), dtype=torch.bool)
            col = bags.ids[bags.bag % bags.F == f].numpy() if bags is not None else xc[:, f]
            m[torch.as_tensor(col[(col >= 0) & (col < v)])] = True
            rows['embedding/embed_cate_w_%d' % f] = m
        return rows

    def _bags(self, X_cate):
        if isinstance(X_cate, BagIds):
            return X_cate
        if isinstance(X_cate, tuple) and len(X_cate) == 2:
            return BagIds(X_cate[0], X_cate[1], len(self.cfg.cate_list))
        return None

    def _inputs(self, X_cate, X_cont, y):
        xc = self._bags(X_cate)
        if xc is None:
            xc = torch.as_tensor(np.asarray(X_cate).astype(np.int32).astype(np.int64))   # feed narrows to int32
        xf = torch.as_tensor(np.asarray(X_cont).astype(np.float32)).to(self.dtype)
        yy = torch.as_tensor(np.asarray(y).astype(np.float32)).to(self.dtype)
        return xc, xf, yy

    def forward(self, X_cate, X_cont, masks=None, P=None):
        xc, xf, _ = self._inputs(X_cate, X_cont, np.zeros(len(X_cont)))
        with torch.no_grad():
            return FORWARD[self.model](P or self.P, self.cfg, xc, xf, masks, self.literal)

    def loss_and_grads(self, X_cate, X_cont, y, masks=None):
        xc, xf, yy = self._inputs(X_cate, X_cont, y)
        P = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in self.P.items())
        out = FORWARD[self.model](P, self.cfg, xc, xf, masks, self.literal)
        loss = model_loss(self.model, out, yy)
        grads = torch.autograd.grad(loss, list(P.values()), allow_unused=True)
        g = OrderedDict()
        for (k, v), gi in zip(P.items(), grads):
            g[k] = torch.zeros_like(v) if gi is None else gi.detach()
        return loss.detach(), g, out

    def lr(self):
        c = self.cfg
        return exponential_decay(c.learning_rate, self.global_step, c.decay_steps, c.decay_rate)

    def run(self, X_cate, X_cont, y, is_train=False, masks=None):
        if is_train:
            loss, g, _ = self.loss_and_grads(X_cate, X_cont, y, masks)
            if self.lazy:
                self.opt.apply(self.P, g, self.lr(), rows=self.touched_rows(X_cate))
            else:
                self.opt.apply(self.P, g, self.lr())
            self.global_step += 1
            return np.float32(loss.item())
        out = self.forward(X_cate, X_cont)
        return np.asarray(y, dtype=np.float32), tf_round(out.y_out).to(torch.float32).numpy()


class OracleDIN:
    """DIN graph + SGD (DIN/DIN.py:5-131)."""

    def __init__(self, cfg, cate_list, params, dtype=torch.float64):
        self.cfg, self.dtype = cfg, dtype
        self.cate_list = torch.as_tensor(np.asarray(cate_list), dtype=torch.int64)
        self.P = OrderedDict((k, torch.as_tensor(np.asarray(v), dtype=dtype).clone()) for k, v in params.items())

    def _trainable(self, k):
        return not k.endswith(DIN_NON_TRAINABLE)

    def loss_and_grads(self, uij):
        _, i, y, hist_i, sl = uij
        P = OrderedDict((k, v.clone().requires_grad_(self._trainable(k))) for k, v in self.P.items())
        out = din_forward(P, self.cfg, self.cate_list, torch.as_tensor(np.asarray(i)),
                          torch.as_tensor(np.asarray(hist_i)), torch.as_tensor(np.asarray(sl)))
        yy = torch.as_tensor(np.asarray(y, dtype=np.float32)).to(self.dtype)
        loss = sigmoid_ce_with_logits(yy, out.logit).mean()
        names = [k for k in P if self._trainable(k)]
        grads = torch.autograd.grad(loss, [P[k] for k in names], allow_unused=True)
        g = OrderedDict((k, torch.zeros_like(P[k]) if gi is None else gi.detach()) for k, gi in zip(names, grads))
        return loss.detach(), g, out

    def train(self, uij, lr):
        loss, g, _ = self.loss_and_grads(uij)
        for k, gi in g.items():
            self.P[k] = self.P[k] - lr * gi
        return np.float32(loss.item())

    def eval(self, uij):
        _, i, j, hist_i, sl = uij
        with torch.no_grad():
            out = din_forward(self.P, self.cfg, self.cate_list, torch.as_tensor(np.asarray(i)),
                              torch.as_tensor(np.asarray(hist_i)), torch.as_tensor(np.asarray(sl)),
                              j=torch.as_tensor(np.asarray(j)))
        return np.float32(out.mf_auc.item()), out.p_and_n.to(torch.float32).numpy()

    def test(self, i, hist_i, sl):
        with torch.no_grad():
            return din_logits_all(self.P, self.cfg, self.cate_list, torch.as_tensor(np.asarray(i)),
                                  torch.as_tensor(np.asarray(hist_i)), torch.as_tensor(np.asarray(sl))).numpy()


def default_cfg(model, **kw):
    """Flag defaults of the reference main.py files (DeepFM/main.py:48-59 and siblings)."""
    c = dict(embed_size=8, batch_size=128, epoch=10, regular_rate=0.1, learning_rate=0.001,
             decay_steps=10000, decay_rate=0.99, mode='R')
    if model in ('DeepFM', 'NFM', 'ESMM', 'DCN'):
        c['deep_layers'] = [256, 128, 64, 32]
    if model == 'DeepFM':
        c.update(dropout_fm=0.5, dropout_deep=0.2)
    if model == 'NFM':
        c.update(dropout_bi=0.5, dropout_deep=0.2)
    if model == 'DCN':
        c.update(cross_layer_num=3, dropout=0.2)
    if model == 'AFM':
        c.update(attention_size=10, dropout=0.2)
    if model == 'PNN':
        c.update(inner_product=True, product_size=50, hidden_layers=[32, 64, 128], dropout=0.2)
    if model == 'ESMM':
        c.update(dropout=0.2)
    if model == 'Wide_Deep':
        c = dict(embed_size=8, hidden_layers=[32, 64, 128], batch_size=128, epoch=10, regular_rate=0.1,
                 learning_rate=0.001, dropout=0.2, decay_steps=10000, decay_rate=0.99)
    if model == 'NCF':
        c = dict(embed_size=16, batch_size=128, epoch=10, regular_rate=0.1, learning_rate=0.001, dropout=0.2,
                 decay_steps=10000, decay_rate=0.99)
    if model == 'DIN':
        c = dict(train_batch_size=32, test_batch_size=512, hidden_units=32, epoch=10, learning_rate=0.1)
    c.update(kw)
    return SimpleNamespace(**c)

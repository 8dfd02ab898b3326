"""Host-side building blocks shared by the model classes: flat dense parameters, the sharded-
ready embedding tables, the embedding/FM front end in both semantics, and the MLP tower.
All arithmetic happens in libctrhip kernels (ops.py); this file only owns buffers and order.
"""
from __future__ import annotations

import contextlib
import os
import math
from collections import OrderedDict

import numpy as np
import torch

from . import ops
from ._lib import OPT_ADAGRAD, OPT_ADAM, OPT_GRAD_ONLY, OPT_SGD

OPT_KIND = dict(adam=OPT_ADAM, adagrad=OPT_ADAGRAD, sgd=OPT_SGD)


def round_up(x, m):
    return (x + m - 1) // m * m


def truncated_normal(rng, shape, std=0.01):
    """Host initialiser with the distribution of tf.truncated_normal_initializer(0, std)
    (DeepFM/DeepFM.py:30): re-sample outside two standard deviations."""
    x = rng.standard_normal(size=shape)
    bad = np.abs(x) > 2.0
    while bad.any():
        x[bad] = rng.standard_normal(size=int(bad.sum()))
        bad = np.abs(x) > 2.0
    return (x * std).astype(np.float32)


class BufferCache(OrderedDict):
    """Per-shape scratch buffers of a model, least recently used first, at most `maxsize` unpinned shapes: a caller
    whose batch size (or DIN history length) changes all the time re-allocates instead of accumulating device memory.
    A shape that a captured graph points at is pinned (`pin`) and never evicted while the graph lives: a replay
    reads and writes those buffers without ever coming through lookup()."""

    def __init__(self, maxsize=8):
        super().__init__()
        self.maxsize = maxsize
        self.pinned = set()

    def lookup(self, key, make):
        if key in self:
            self.move_to_end(key)
            return self[key]
        free = [k for k in self if k not in self.pinned]
        while len(free) >= self.maxsize:
            del self[free.pop(0)]
        self[key] = make()
        return self[key]

    def pin(self, key):
        if key not in self:
            raise KeyError(key)
        self.pinned.add(key)

    def clear(self):
        super().clear()
        self.pinned.clear()


class DenseParams:
    """All dense variables of a model in ONE flat fp32 buffer (+ flat grad and optimizer
    slots), so the optimizer is a single launch and a data-parallel all-reduce is one call."""

    def __init__(self, shapes, device):
        self.shapes = OrderedDict(shapes)
        self.offsets = OrderedDict()
        off = 0
        for k, s in self.shapes.items():
            self.offsets[k] = off
            off += round_up(int(np.prod(s)), 4)          # keep every tensor 16-byte aligned
        self.size = off
        self.flat = torch.zeros(max(off, 4), dtype=torch.float32, device=device)
        self.grad = torch.zeros_like(self.flat)
        self.slot0 = None
        self.slot1 = None

    def _view(self, buf, k):
        s = self.shapes[k]
        n = int(np.prod(s))
        return buf[self.offsets[k]:self.offsets[k] + n].view(*s)

    def p(self, k):
        return self._view(self.flat, k)

    def g(self, k):
        return self._view(self.grad, k)

    def mat(self, k, buf=None):
        """2-D [rows, cols] view (vectors become [1, n])."""
        v = self._view(self.flat if buf is None else buf, k)
        return v if v.dim() == 2 else v.reshape(1, -1) if v.dim() == 1 else v.reshape(v.shape[0], -1)

    def stacked(self, names, buf=None):
        """[len(names), n] view over consecutively registered, equally sized variables
        (row stride = the 4-float-padded size)."""
        b = self.flat if buf is None else buf
        n = int(np.prod(self.shapes[names[0]]))
        stride = round_up(n, 4)
        off0 = self.offsets[names[0]]
        for i, k in enumerate(names):
            if self.offsets[k] != off0 + i * stride or int(np.prod(self.shapes[k])) != n:
                raise ValueError('variables are not stacked: %s' % (names,))
        return b[off0:off0 + len(names) * stride].view(len(names), stride)[:, :n]

    def g_with_bias_row(self, wn, bn):
        """[K+1, N] view of the gradient buffer covering dW (K rows) and, as its last row, db - or None when the
        two variables are not adjacent in the flat buffer.  A weight-gradient product whose input carries a column
        of ones after its K features writes both at once."""
        K, N = self.shapes[wn]
        if int(np.prod(self.shapes[bn])) != N or self.offsets[bn] != self.offsets[wn] + K * N:
            return None
        o = self.offsets[wn]
        return self.grad[o:o + (K + 1) * N].view(K + 1, N)

    def init_slots(self, kind):
        if kind == 'adam':
            self.slot0 = torch.zeros_like(self.flat)
            self.slot1 = torch.zeros_like(self.flat)
        elif kind == 'adagrad':
            self.slot0 = torch.full_like(self.flat, 0.1)   # tf.train.AdagradOptimizer initial accumulator
            self.slot1 = None
        elif kind == 'ftrl':
            self.slot0 = torch.full_like(self.flat, 0.1)   # tf.train.FtrlOptimizer initial accumulator
            self.slot1 = torch.zeros_like(self.flat)       # linear term

    def apply(self, opt):
        ops.dense_update(self.flat, self.grad, opt, self.slot0, self.slot1)


class EmbeddingTables:
    """All per-field tables concatenated into one [sum(V), E] buffer with a row-offset array
    (the layout the gather and the sorted backward index with one 32-bit global row id)."""

    def __init__(self, cate_list, E, device):
        self.cate_list = [int(v) for v in cate_list]
        self.F = len(self.cate_list)
        self.E = int(E)
        offs = np.zeros(self.F + 1, dtype=np.int64)
        offs[1:] = np.cumsum(self.cate_list)
        self.offsets_host = offs
        self.total_rows = int(offs[-1])
        self.row_offset = torch.as_tensor(offs, device=device)
        self.weight = torch.zeros(self.total_rows, self.E, dtype=torch.float32, device=device)
        self.slot0 = None
        self.slot1 = None
        self.grad = None        # dense gradient buffer of the TF-1.x-exact (non-lazy) Adam path

    def table(self, f, buf=None):
        b = self.weight if buf is None else buf
        return b[int(self.offsets_host[f]):int(self.offsets_host[f + 1])]

    def init_slots(self, kind, dense_adam, keep_grad=True):
        """dense_adam: every row moves each step (tf.train.AdamOptimizer); keep_grad: through a dense gradient buffer
        that get_grads() can show (False: row-wise pass + decay pass of the rows without a gradient, no buffer)."""
        self.touched = None
        if kind == 'adam' and dense_adam and not keep_grad:
            self.slot0 = torch.zeros_like(self.weight)
            self.slot1 = torch.zeros_like(self.weight)
            self.touched = torch.zeros(self.total_rows, dtype=torch.uint8, device=self.weight.device)
            return
        if kind == 'adam':
            if dense_adam or self.E > 16:
                self.slot0 = torch.zeros_like(self.weight)
                self.slot1 = torch.zeros_like(self.weight)
            else:
                # rows of up to 64 bytes: m and v side by side in one [rows, 2E] buffer, so the row-wise update
                # touches one 128-byte line for both (at E = 16) instead of two half-used ones; slot0 / slot1 are
                # its column blocks (checkpoints and get/set see [rows, E] tensors as before)
                self.slots = torch.zeros(self.total_rows, 2 * self.E, dtype=torch.float32, device=self.weight.device)
                self.slot0, self.slot1 = self.slots[:, :self.E], self.slots[:, self.E:]
            if dense_adam:
                self.grad = torch.zeros_like(self.weight)
        elif kind == 'adagrad':
            self.slot0 = torch.full_like(self.weight, 0.1)
        elif kind == 'ftrl':
            self.slot0 = torch.full_like(self.weight, 0.1)
            self.slot1 = torch.zeros_like(self.weight)


class Bags:
    """Multi-hot categorical input: bag (b, f) holds ids[offsets[b*F+f] : offsets[b*F+f+1]], all of field f, and is
    sum-pooled into one row of Z (the single id per field of DeepFM/DeepFM.py:43-53 is the one-member case).
    ids: int32 [nnz]; offsets: int32 [B*F+1], non-decreasing from 0 to nnz.  Both on the device."""

    def __init__(self, ids, offsets, B, F):
        if ids.dtype != torch.int32 or offsets.dtype != torch.int32 or ids.dim() != 1 or offsets.dim() != 1:
            raise TypeError('Bags: ids and offsets must be 1-D int32 tensors')
        if offsets.numel() != B * F + 1:
            raise ValueError('Bags: offsets must have B*F+1 = %d entries, got %d' % (B * F + 1, offsets.numel()))
        self.ids, self.offsets, self.shape = ids.contiguous(), offsets.contiguous(), (int(B), int(F))

    @classmethod
    def from_host(cls, ids, offsets, F, device):
        """Validates a host CSR (monotone, starting at 0, ending at len(ids)) and uploads it."""
        ids = np.ascontiguousarray(np.asarray(ids)).astype(np.int32).reshape(-1)
        off = np.ascontiguousarray(np.asarray(offsets)).astype(np.int64).reshape(-1)
        if off.size < 1 or (off.size - 1) % F or off[0] != 0 or off[-1] != ids.size or np.any(np.diff(off) < 0):
            raise ValueError('bag offsets must be non-decreasing, start at 0, end at len(ids) and hold B*F+1 entries')
        if ids.size > 2 ** 31 - 1:
            raise ValueError('too many bag members')
        return cls(torch.as_tensor(ids, device=device), torch.as_tensor(off.astype(np.int32), device=device),
                   (off.size - 1) // F, F)

    @property
    def nnz(self):
        return self.ids.numel()


class EmbedFront:
    """Embedding block + FM reductions in either semantics.

    mode 'R': the reference block exactly as written (DeepFM/DeepFM.py:43-53): Z is the dense
              [B, sum(V), E] tensor whose only non-zero rows are copies of W_f[1].
    mode 'S': standard per-field lookup Z[b,f,:] = W_f[x[b,f]] (Wide_Deep/Wide_Deep.py:51).
    cont_embed adds the value-scaled continuous rows of AFM/PNN (AFM/AFM.py:52-57)."""

    def __init__(self, tables, mode, cont_num, cont_embed):
        self.t = tables
        self.mode = mode
        self.C = int(cont_num)
        self.cont_embed = bool(cont_embed)
        self.Ncat = tables.total_rows if mode == 'R' else tables.F
        self.N = self.Ncat + (self.C if cont_embed else 0)
        self.E = tables.E
        self.oor_count = None     # optional device int32[1]: the standard lookup counts out-of-range ids here

    def forward(self, idx, x_cont, *, z_out, cont_tail=False, cont_w=None, first_out=None, s_out=None,
                second_out=None, drop_first=None, drop_second=None, oor_count=None, amax_out=None):
        """amax_out (optional largest-magnitude record the caller zeroed): raised to the largest magnitude of what
        this call writes to z_out (the standard lookup does it inside the gather; the other forms pay one pass)."""
        t, E = self.t, self.E
        want_fm = first_out is not None or s_out is not None or second_out is not None
        if amax_out is not None and (self.mode != 'S' or isinstance(idx, Bags)):
            self.forward(idx, x_cont, z_out=z_out, cont_tail=cont_tail, cont_w=cont_w, first_out=first_out, s_out=s_out,
                         second_out=second_out, drop_first=drop_first, drop_second=drop_second, oor_count=oor_count)
            width = self.N * E + (self.C if cont_tail else 0)
            ops.absmax(z_out[:, :width], amax_out)
            return
        if isinstance(idx, Bags):
            if self.mode != 'S':
                raise ValueError('multi-hot bags are defined for the standard lookup (mode S)')
            if z_out is None:
                raise ValueError('multi-hot bags need the Z buffer')
            ops.embed_bag_fwd(t.weight, t.row_offset, idx.ids, idx.offsets, x_cont,
                              cont_w if self.cont_embed else None, B=idx.shape[0], F=idx.shape[1], E=E, z_out=z_out,
                              cont_tail=cont_tail, first_out=first_out, s_out=s_out, second_out=second_out,
                              drop_first=drop_first, drop_second=drop_second,
                              oor_count=oor_count if oor_count is not None else self.oor_count)
            return
        if self.mode == 'S':
            ops.embed_fwd(t.weight, t.row_offset, idx, x_cont, cont_w if self.cont_embed else None, E=E,
                          z_out=z_out, cont_tail=cont_tail, first_out=first_out, s_out=s_out,
                          second_out=second_out, drop_first=drop_first, drop_second=drop_second,
                          oor_count=oor_count if oor_count is not None else self.oor_count, amax_out=amax_out)
            return
        if z_out is None:
            raise ValueError('R-mode needs the dense Z buffer')
        ops.embed_onehot_fwd(t.weight, t.row_offset, idx, x_cont, E=E, z_out=z_out,
                             cont_tail=cont_tail and not self.cont_embed)
        if self.cont_embed:
            empty_idx = idx[:, :0].contiguous()
            ops.embed_fwd(t.weight, t.row_offset[:1], empty_idx, x_cont, cont_w, E=E,
                          z_out=z_out[:, self.Ncat * E:], cont_tail=cont_tail)
        if want_fm:
            ops.fm_fwd(z_out, N=self.N, E=E, first_out=first_out, s_out=s_out, second_out=second_out,
                       drop_first=drop_first, drop_second=drop_second)

    def sort_pairs(self, idx, keys, src, ws):
        """The (key, src) pair sort of the table backward; it reads only the ids, so a caller may run it early, on
        another stream, and pass presorted=True to backward_tables."""
        t = self.t
        if isinstance(idx, Bags):
            ops.emb_bag_sort(idx.ids, idx.offsets, t.row_offset, B=idx.shape[0], F=idx.shape[1], N=self.N,
                             total_rows=t.total_rows, keys_out=keys, src_out=src, ws=ws)
            return
        ops.emb_sort(idx, t.row_offset, N=self.N, total_rows=t.total_rows, rmode=(self.mode == 'R'),
                     keys_out=keys, src_out=src, ws=ws)

    def backward_tables(self, idx, dz, keys, src, ws, opt, lazy, presorted=False, fm_g2=None):
        """dz: [B, >= N*E] gradient with respect to Z.  Sort (key, src), segment-reduce, update.
        fm_g2 (optional): dz lacks -g2 * Z of the FM second-order term; the update subtracts it per table row
        (single-id standard lookup only: a row of Z is the table row)."""
        t = self.t
        B, F = idx.shape
        n_items = idx.nnz if isinstance(idx, Bags) else B * F     # one pair per bag member
        if n_items == 0:
            return
        if not presorted:
            self.sort_pairs(idx, keys, src, ws)
        if fm_g2 is not None and (self.mode != 'S' or isinstance(idx, Bags) or self.cont_embed):
            raise ValueError('fm_g2 is defined for the single-id standard lookup without embedded continuous rows')
        if opt.kind == OPT_ADAM and not lazy and t.grad is None:
            # every row moves, as tf.train.AdamOptimizer moves it, without a dense gradient buffer: the rows with a
            # gradient by the row-wise pass, all others by the decay pass
            ops.emb_sparse_apply(keys, src, n_items, N=self.N, E=self.E, total_rows=t.total_rows, dz=dz,
                                 opt=opt, tables=t.weight, slot0=t.slot0, slot1=t.slot1, ws=ws, fm_g2=fm_g2)
            ops.adam_untouched_rows(keys, n_items, total_rows=t.total_rows, E=self.E, opt=opt, tables=t.weight,
                                    slot0=t.slot0, slot1=t.slot1, touched=t.touched)
        elif opt.kind == OPT_ADAM and not lazy:
            t.grad.zero_()
            ops.emb_sparse_apply(keys, src, n_items, N=self.N, E=self.E, total_rows=t.total_rows, dz=dz,
                                 opt=ops.make_opt(OPT_GRAD_ONLY), slot0=t.grad, ws=ws,
                                 **(dict(tables=t.weight, fm_g2=fm_g2) if fm_g2 is not None else {}))
            ops.dense_update(t.weight, t.grad, opt, t.slot0, t.slot1)
        else:
            ops.emb_sparse_apply(keys, src, n_items, N=self.N, E=self.E, total_rows=t.total_rows, dz=dz,
                                 opt=opt, tables=t.weight, slot0=t.slot0, slot1=t.slot1, ws=ws, fm_g2=fm_g2)

    def backward_cont_w(self, dz, x_cont, g_cont_w, ws):
        """d cont_w[c,:] = sum_b dz[b, Ncat + c, :] * x_cont[b,c]."""
        E = self.E
        ops.colsum(dz[:, self.Ncat * E:(self.Ncat + self.C) * E], g_cont_w.reshape(-1), w=x_cont, w_group=E, ws=ws)


def ones_padded(B, L, device):
    """A [B, L] activation buffer inside [B, round_up(L + 1, 4)] storage whose column L holds ones: returns
    (view [B, L], view [B, L + 1]).  Kernels write the first L columns only (they take the row pitch), so the ones
    stay; the weight-gradient product of the layer this buffer feeds uses the wider view and gets the bias gradient
    as one more output row instead of a separate pass over the gradient."""
    buf = torch.zeros(B, round_up(L + 1, 4), dtype=torch.float32, device=device)
    buf[:, L] = 1.0
    return buf[:, :L], buf[:, :L + 1]


class Tower:
    """relu(x W_i + b_i) -> dropout chain on the fp32 MFMA GEMM, with the relu/dropout backward
    fused into the next GEMM's epilogue (DeepFM/DeepFM.py:65-80)."""

    def __init__(self, dense, prefix, n_layers):
        self.dense = dense
        self.names = [('%s_w_%d' % (prefix, i), '%s_b_%d' % (prefix, i)) for i in range(n_layers)]
        self.widths = [dense.shapes[w][1] for w, _ in self.names]
        self.d_in = dense.shapes[self.names[0][0]][0]
        self._chain = None
        self._bwd_chain = None
        self._tail_wgrad = None
        self.am = None          # largest magnitudes of the products' operands (enable_amax)
        self._x_amax = None
        # chain backward + its two weight gradients in one launch (ops.tail_bwd_fused): 15 us less kernel time at
        # B = 65536 but 3 % more step time, because the weight-gradient half then sits on the main stream ahead of the
        # activation-gradient GEMMs instead of beside the embedding backward on the side stream - off by default
        self.fuse_tail = os.environ.get('CTR_FUSE_TAIL', '0') == '1'

    def enable_amax(self, device):
        """Keeps the largest magnitude of every operand of the tower's wide products in one device word each, so that
        those products can run as the two-term fp16 split (ops.gemm, a_amax / b_amax): the weights' records are
        refreshed by begin_step(), a layer's output and gradient words by the product that writes them, the input's
        word by whoever fills the input (x_amax of forward()).  Words: [0, n) weights, [n, 2n) layer outputs,
        [2n, 3n) gradients with respect to the layer outputs, 3n the tower's input."""
        n = len(self.names)
        self.am = torch.zeros((3 * n + 1) * ops.AMAX_FLOATS, dtype=torch.float32, device=device)
        offs = [self.dense.offsets[w] for w, _ in self.names]
        cnts = [int(np.prod(self.dense.shapes[w])) for w, _ in self.names]
        self._am_offs = torch.tensor(offs, dtype=torch.int64, device=device)
        self._am_cnts = torch.tensor(cnts, dtype=torch.int64, device=device)
        self._have_g = set()

    @property
    def x_amax(self):
        """The input's word, for the producer of x to raise (None while enable_amax() has not been called)."""
        n = len(self.names)
        return None if self.am is None else self.am[3 * n * ops.AMAX_FLOATS:(3 * n + 1) * ops.AMAX_FLOATS]

    def begin_step(self):
        """Start of a forward pass: every word back to zero, the weights' words recomputed (one small launch)."""
        if self.am is None:
            return
        self.am.zero_()
        ops.absmax_tensors(self.dense.flat, self._am_offs, self._am_cnts, self.am)
        self._have_g = set()

    def g_last_amax(self):
        """The record of the gradient with respect to the last layer's output (None without enable_amax()), for a
        producer that can raise it while it writes that gradient (unit_backward)."""
        return None if self.am is None else self._slot('g', len(self.names) - 1)

    def _slot(self, kind, i):
        n = len(self.names)
        k = (dict(w=0, a=n, g=2 * n)[kind] + i) * ops.AMAX_FLOATS
        return self.am[k:k + ops.AMAX_FLOATS]

    def forward(self, x, acts, drops, ws, x_amax=None):
        """x: [B, >= d_in]; acts[i]: output buffer of layer i ([B, L_i] views).  x_amax: the word holding max |x|
        (self.x_amax, raised by the producer of x after begin_step()), or None."""
        h = x[:, :self.d_in]
        n = len(self.names)
        chain = self._chain_len(acts)
        am = self.am is not None
        self._x_amax = x_amax if am else None
        for i, (wn, bn) in enumerate(self.names[:n - chain]):
            kw = {}
            if am:
                kw['amax_out'] = self._slot('a', i)
                a_in = x_amax if i == 0 else self._slot('a', i - 1)
                if a_in is not None:
                    kw.update(a_amax=a_in, b_amax=self._slot('w', i))
            ops.gemm(h, self.dense.p(wn), acts[i], bias=self.dense.p(bn), relu=True, dropout=drops[i], ws=ws, **kw)
            h = acts[i]
        if chain:
            k = n - chain
            ops.mlp_chain_fwd(h, [self.dense.p(w) for w, _ in self.names[k:]],
                              [self.dense.p(b).reshape(-1) for _, b in self.names[k:]], drops[k:], acts[k:])

    def _chain_len(self, acts):
        """How many of the last (narrow) layers go through one chained launch: the longest covered tail."""
        if self._chain is None:
            self._chain = 0
            n = len(self.names)
            for c in (3, 2):
                k = n - c
                if k < 1 or not ops.mlp_chain_supported(self.widths[k - 1], self.widths[k:]):
                    continue
                if all(a.stride(0) % 4 == 0 and a.data_ptr() % 16 == 0 for a in acts[k - 1:n - 1]):
                    self._chain = c
                    break
        return self._chain

    def backward(self, x, acts, g_last, gbufs, rates, dx, ws, accumulate_dx=False, g_last_masked=False, side=None,
                 ext=None, dx_fm=None, g_last_has_amax=False):
        """g_last: gradient w.r.t. acts[-1] (post relu+dropout); it is masked in place unless the caller
        already did (g_last_masked; g_last_has_amax: and raised g_last_amax() to its largest magnitude).
        gbufs[i]: scratch [B, L_i] for i < n-1.  dx (optional): [B, >= d_in] receives dL/dx
        (added to its content when accumulate_dx).  ext (optional): per layer, the layer's input as a [B, K+1] view
        whose last column is ones (ones_padded), or None.  dx_fm (optional): the fm_bwd argument of ops.gemm for the
        product that writes dx (the FM terms of dZ added in its epilogue instead of by a pass over dx)."""
        n = len(self.names)
        scale = [1.0 / (1.0 - r) if r else 1.0 for r in rates]
        if not g_last_masked:
            ops.relu_mask_apply(g_last, acts[n - 1], scale[n - 1])
        g = g_last
        am = self.am is not None
        if am and g_last_masked and g_last_has_amax:
            self._have_g.add(n - 1)
        n_fwd_gemm = n - self._chain_len(acts)          # layers whose output word the forward pass raised

        def in_amax(i):                                  # word of layer i's input, or None
            if not am:
                return None
            return self._x_amax if i == 0 else (self._slot('a', i - 1) if i - 1 < n_fwd_gemm else None)
        # the last two activation gradients in one chained launch when the widths are covered
        chained = 0
        tail_done = False
        if n >= 3 and self._bwd_chain_ok(acts, gbufs):
            if self.fuse_tail:
                # ... and both layers' weight / bias gradients in the same pass over the activations
                (wa, ba), (wb, bb) = self.names[n - 2], self.names[n - 1]
                ops.tail_bwd_fused(g_last, self.dense.p(wb), acts[n - 2], scale[n - 2], self.dense.p(wa), acts[n - 3],
                                   scale[n - 3], gbufs[n - 3], self.dense.g(wa), self.dense.g(ba).reshape(-1),
                                   self.dense.g(wb), self.dense.g(bb).reshape(-1), ws=ws)
                tail_done = True
            else:
                ops.mlp_chain_bwd(g_last, self.dense.p(self.names[n - 1][0]), acts[n - 2], scale[n - 2], gbufs[n - 2],
                                  self.dense.p(self.names[n - 2][0]), acts[n - 3], scale[n - 3], gbufs[n - 3],
                                  g_out_amax=self._slot('g', n - 3) if am else None)
                if am:
                    self._have_g.add(n - 3)
            chained = 2
        # activation gradients first (they are what the rest of the step waits for) ...
        gs = [None] * n
        gs[n - 1] = g_last
        top = n - 1 - chained
        if am and top not in self._have_g:
            # the gradient that enters the first wide product comes from a kernel that does not track its maximum
            # (output unit, fused tail): one pass over it
            ops.absmax(g_last if top == n - 1 else gbufs[top], self._slot('g', top))
            self._have_g.add(top)
        for i in range(n - 1, -1, -1):
            wn = self.names[i][0]
            kw = {}
            if am and i in self._have_g:
                kw.update(a_amax=self._slot('g', i), b_amax=self._slot('w', i))
            if i > 0:
                if chained:
                    chained -= 1
                else:
                    if am:
                        kw['amax_out'] = self._slot('g', i - 1)
                        self._have_g.add(i - 1)
                    ops.gemm(gs[i], self.dense.p(wn), gbufs[i - 1], trans_b=True, mask_src=acts[i - 1],
                             mask_scale=scale[i - 1], ws=ws, **kw)
                gs[i - 1] = gbufs[i - 1]
            elif dx is not None:
                ops.gemm(gs[0], self.dense.p(wn), dx[:, :self.d_in], trans_b=True, accumulate=accumulate_dx,
                         fm_bwd=dx_fm, ws=ws, **kw)
        # ... then the weight gradients: dW = h_in^T g, db.  With `side` (a Session) they go to its side stream
        # and overlap whatever the caller launches next; the caller joins before the dense optimizer.
        ctx = side.side_stream() if side is not None else contextlib.nullcontext()
        wws = side.ws_side if side is not None else ws
        with ctx:
            first = n - 1
            if tail_done:
                first = n - 3
            elif n >= 3 and self._tail_wgrad_ok(acts, gs):
                # the last two (narrow) layers' weight and bias gradients in one pass over their operands
                (wa, ba), (wb, bb) = self.names[n - 2], self.names[n - 1]
                ops.tail_wgrad2(acts[n - 3], gs[n - 2], self.dense.g(wa), self.dense.g(ba).reshape(-1),
                                acts[n - 2], gs[n - 1], self.dense.g(wb), self.dense.g(bb).reshape(-1), ws=wws)
                first = n - 3
            for i in range(first, -1, -1):
                wn, bn = self.names[i]
                gext = self.dense.g_with_bias_row(wn, bn) if ext is not None and ext[i] is not None else None
                K = self.dense.shapes[wn][0]
                kw = {}
                if am and i in self._have_g and in_amax(i) is not None:
                    kw = dict(x_amax=in_amax(i), g_amax=self._slot('g', i))
                if gext is not None and (K + 128) // 128 == (K + 127) // 128:
                    # db falls out of the product as its last row (only where that row fits the last 128-row tile)
                    if kw:
                        kw['x_amax_floor'] = 1.0                   # the column of ones
                    ops.dense_wgrad(ext[i], gs[i], gext, None, ws=wws, **kw)
                    continue
                h_in = x[:, :self.d_in] if i == 0 else acts[i - 1]
                ops.dense_wgrad(h_in, gs[i], self.dense.g(wn), self.dense.g(bn).reshape(-1), ws=wws, **kw)

    def _tail_wgrad_ok(self, acts, gs):
        if self._tail_wgrad is None:
            n = len(self.names)
            ok = ops.tail_wgrad2_supported(self.widths[n - 3], self.widths[n - 2], self.widths[n - 2], self.widths[n - 1])
            for t in (acts[n - 3], gs[n - 2], acts[n - 2]):
                ok = ok and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0
            self._tail_wgrad = bool(ok)
        return self._tail_wgrad

    def _bwd_chain_ok(self, acts, gbufs):
        if self._bwd_chain is None:
            n = len(self.names)
            ok = ops.mlp_chain_bwd_supported(self.widths[n - 1], self.widths[n - 2], self.widths[n - 3])
            for t in (acts[n - 2], acts[n - 3], gbufs[n - 2], gbufs[n - 3]):
                ok = ok and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0
            self._bwd_chain = bool(ok)
        return self._bwd_chain


def unit_backward(dense, wn, bn, x, dx, dl, ws, mask_from=-1, rate=0.0, keep=None, g_amax=None):
    """Backward of an output unit y = x w + b: dx = dl w^T, dw, db.  Columns mask_from.. of x are the output of
    a tower's last relu (+ dropout at `rate`) layer; returns True when their mask was applied to dx here (one
    fused pass), False when the caller (Tower.backward) still has to.  keep (optional): (cols_a, drop_a, cols_b,
    drop_b), the keep masks of two dropouts for dx's leading column blocks (ops.unit_bwd).  g_amax (optional,
    Tower.g_last_amax()): the record of the tower's incoming gradient; the fused pass raises it (a True return then
    also means Tower.backward can be told g_last_has_amax)."""
    B, K = x.shape
    w = dense.p(wn).reshape(-1)
    if ops.unit_bwd_supported(K):
        scale = 1.0 / (1.0 - rate) if rate else 1.0
        ops.unit_bwd(dl, w, x, dx, dense.g(wn).reshape(-1), dense.g(bn).reshape(-1), mask_from=mask_from,
                     mask_scale=scale, keep=keep, masked_amax=g_amax if mask_from >= 0 else None, ws=ws)
        return mask_from >= 0
    ops.rowdot_bwd_x(dl, w, dx)
    if keep is not None:
        ops.dropout_apply2(dx, *keep)
    ops.dense_wgrad(x, dl.view(B, 1), dense.g(wn), dense.g(bn), ws=ws)
    return False


def adam_lr_t(lr, t, b1=0.9, b2=0.999):
    """tf.train.AdamOptimizer's bias-corrected step size for update number t (1-based)."""
    return lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)

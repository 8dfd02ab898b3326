"""Thin operator layer: torch tensors in, libctrhip calls out.

torch is only the carrier of device memory and of the HIP stream; every operator hands raw
device pointers to the C ABI.  Shapes, dtypes and devices are validated here, on the host,
before a kernel is launched.
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib
from ._lib import Dropout, Epilogue, FmBwd, Opt, check

KEEP_ONE = 1 << 24


def lib():
    return _lib.load()


class KernelTimer:
    """Optional HIP-event timer around operator calls (events are recorded on the stream the kernels
    run on).  Used by bench.py to attribute the timed region to kernels.  Events come from a
    pre-allocated pool and `only` restricts the timed operators, so the host cost stays small."""

    def __init__(self, only=None, pool=4096):
        self.only = set(only) if only else None
        self.pool = [torch.cuda.Event(enable_timing=True) for _ in range(pool)]
        self.used = 0
        self.records = []          # (name, start_event, end_event)

    def pair(self):
        if self.used + 2 > len(self.pool):
            return None
        a, b = self.pool[self.used], self.pool[self.used + 1]
        self.used += 2
        return a, b

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, a, b in self.records:
            ms = a.elapsed_time(b)
            tot, cnt = out.get(name, (0.0, 0))
            out[name] = (tot + ms, cnt + 1)
        return {k: dict(total_ms=v[0], calls=v[1], avg_us=1e3 * v[0] / v[1]) for k, v in out.items()}


_TIMER = None


def set_timer(timer):
    global _TIMER
    _TIMER = timer


def _timed(fn):
    import functools

    @functools.wraps(fn)
    def wrap(*a, **k):
        if _TIMER is None:
            return fn(*a, **k)
        name = fn.__name__
        if name == 'gemm':                      # one bucket per GEMM shape
            A, Cout = a[0], a[2]
            ta = bool(k.get('trans_a', False))
            name = 'gemm_%dx%dx%d_%s%s' % (Cout.shape[0], Cout.shape[1], A.shape[0] if ta else A.shape[1],
                                            'T' if ta else 'N', 'T' if k.get('trans_b', False) else 'N')
        elif name == 'dense_wgrad':             # and per weight-gradient shape (M x N x K)
            name = 'wgrad_%dx%dx%d' % (a[0].shape[1], a[1].shape[1], a[0].shape[0])
        if _TIMER.only is not None and name not in _TIMER.only and fn.__name__ not in _TIMER.only:
            return fn(*a, **k)
        ev = _TIMER.pair()
        if ev is None:
            return fn(*a, **k)
        e0, e1 = ev
        e0.record()
        r = fn(*a, **k)
        e1.record()
        _TIMER.records.append((name, e0, e1))
        return r
    return wrap


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t, dtype=None, name='tensor'):
    if t is None:
        return None
    if not t.is_cuda:
        raise ValueError('%s must live on the GPU (libctrhip has no CPU path)' % name)
    if dtype is not None and t.dtype != dtype:
        raise TypeError('%s: expected %s, got %s' % (name, dtype, t.dtype))
    return C.c_void_p(t.data_ptr())


def _rows(t, name, dtype=torch.float32):
    """(pointer, leading dimension) of a 2-D view whose last dim is contiguous."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError('%s must be 2-D with a contiguous last dimension' % name)
    return _ptr(t, dtype, name), int(t.stride(0)) if t.shape[0] > 1 else max(int(t.stride(0)), t.shape[1])


def make_dropout(rate, seed, stream_id, seed_add=None):
    """Dropout descriptor for tf.layers.dropout(rate) in training mode; None when rate == 0.
    seed_add: optional device int64[1] whose value is added to the seed when the kernel runs."""
    if rate is None or rate <= 0.0:
        return None
    keep = 1.0 - float(rate)
    return Dropout(seed=int(seed), stream_id=int(stream_id),
                   keep_threshold=int(math.floor(keep * KEEP_ONE)), scale=1.0 / keep,
                   seed_add=None if seed_add is None else seed_add.data_ptr())


def _dp(d):
    return C.byref(d) if d is not None else None


class Workspace:
    """Grow-only device scratch buffer handed to the kernels that need one.  A buffer that has been outgrown
    is kept, not freed: a captured graph may still hold its address (a step captured at a small batch is
    replayed after a larger batch made the workspace grow).  Sizes at least double, so what is kept is less
    than the live buffer."""

    def __init__(self, device):
        self.device = device
        self.buf = None
        self.retired = []

    def get(self, nbytes):
        nbytes = int(nbytes)
        if self.buf is None or self.buf.numel() < nbytes:
            if self.buf is not None:
                self.retired.append(self.buf)
                nbytes = max(nbytes, 2 * self.buf.numel())
            self.buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=self.device)
        return C.c_void_p(self.buf.data_ptr()), self.buf.numel()


# ------------------------------------------------------------------ embedding front end
@_timed
def embed_fwd(tables, row_offset, idx, x_cont=None, cont_w=None, *, E, z_out=None, cont_tail=False,
              first_out=None, s_out=None, second_out=None, drop_first=None, drop_second=None,
              oor_count=None, amax_out=None):
    """amax_out (optional largest-magnitude record, AMAX_FLOATS floats zeroed by the caller): raised to the largest
    magnitude written to z_out."""
    B, F = idx.shape
    C_ = 0 if x_cont is None else x_cont.shape[1]
    if idx.dtype != torch.int32 or not idx.is_contiguous():
        raise TypeError('idx must be contiguous int32 [B,F]')
    if row_offset.dtype != torch.int64 or row_offset.numel() != F + 1:
        raise ValueError('row_offset must be int64 [F+1]')
    if tables.dim() != 2 or tables.shape[1] != E or not tables.is_contiguous():
        raise ValueError('tables must be contiguous [rows, E]')
    if x_cont is not None and (not x_cont.is_contiguous() or x_cont.shape[0] != B):
        raise ValueError('x_cont must be contiguous [B,C]')
    if cont_w is not None and (tuple(cont_w.shape) != (C_, E) or not cont_w.is_contiguous()):
        raise ValueError('cont_w must be contiguous [C,E]')
    N = F + (C_ if cont_w is not None else 0)
    zp, zld = (None, 0)
    if z_out is not None:
        zp, zld = _rows(z_out, 'z_out')
        if z_out.shape[0] != B or z_out.shape[1] < N * E + (C_ if cont_tail else 0):
            raise ValueError('z_out too small')
    fp, fld = (None, 0) if first_out is None else _rows(first_out, 'first_out')
    sp, sld = (None, 0) if second_out is None else _rows(second_out, 'second_out')
    if first_out is not None and (first_out.shape[0] != B or first_out.shape[1] < N):
        raise ValueError('first_out too small')
    if second_out is not None and (second_out.shape[0] != B or second_out.shape[1] < E):
        raise ValueError('second_out too small')
    if s_out is not None and (tuple(s_out.shape) != (B, E) or not s_out.is_contiguous()):
        raise ValueError('s_out must be contiguous [B,E]')
    check(lib().ctr_embed_fwd_amax(_ptr(tables, torch.float32, 'tables'), _ptr(row_offset), _ptr(idx),
                                   _ptr(x_cont, torch.float32, 'x_cont'), _ptr(cont_w, torch.float32, 'cont_w'),
                                   B, F, C_, E, zp, zld, int(bool(cont_tail)), fp, fld,
                                   _ptr(s_out, torch.float32, 's_out'), sp, sld, _dp(drop_first), _dp(drop_second),
                                   _ptr(oor_count, torch.int32, 'oor_count'), _amax_rec(amax_out, 'amax_out'),
                                   _stream()), 'ctr_embed_fwd')


@_timed
def embed_bag_fwd(tables, row_offset, idx, bag_offsets, x_cont=None, cont_w=None, *, B, F, E, z_out, cont_tail=False,
                  first_out=None, s_out=None, second_out=None, drop_first=None, drop_second=None, oor_count=None):
    """Multi-hot front end: bag (b,f) = idx[bag_offsets[b*F+f] : bag_offsets[b*F+f+1]], sum-pooled into Z[b,f,:];
    everything else as embed_fwd.  The CSR must be well formed (non-decreasing, ending at idx.numel())."""
    nnz = idx.numel()
    C_ = 0 if x_cont is None else x_cont.shape[1]
    if idx.dtype != torch.int32 or idx.dim() != 1 or not idx.is_contiguous():
        raise TypeError('idx must be contiguous int32 [nnz]')
    if bag_offsets.dtype != torch.int32 or bag_offsets.numel() != B * F + 1 or not bag_offsets.is_contiguous():
        raise ValueError('bag_offsets must be contiguous int32 [B*F+1]')
    if row_offset.dtype != torch.int64 or row_offset.numel() != F + 1:
        raise ValueError('row_offset must be int64 [F+1]')
    if tables.dim() != 2 or tables.shape[1] != E or not tables.is_contiguous():
        raise ValueError('tables must be contiguous [rows, E]')
    if x_cont is not None and (not x_cont.is_contiguous() or x_cont.shape[0] != B):
        raise ValueError('x_cont must be contiguous [B,C]')
    if cont_w is not None and (tuple(cont_w.shape) != (C_, E) or not cont_w.is_contiguous()):
        raise ValueError('cont_w must be contiguous [C,E]')
    N = F + (C_ if cont_w is not None else 0)
    zp, zld = _rows(z_out, 'z_out')
    if z_out.shape[0] != B or z_out.shape[1] < N * E + (C_ if cont_tail else 0):
        raise ValueError('z_out too small')
    fp, fld = (None, 0) if first_out is None else _rows(first_out, 'first_out')
    sp, sld = (None, 0) if second_out is None else _rows(second_out, 'second_out')
    if first_out is not None and (first_out.shape[0] != B or first_out.shape[1] < N):
        raise ValueError('first_out too small')
    if second_out is not None and (second_out.shape[0] != B or second_out.shape[1] < E):
        raise ValueError('second_out too small')
    if s_out is not None and (tuple(s_out.shape) != (B, E) or not s_out.is_contiguous()):
        raise ValueError('s_out must be contiguous [B,E]')
    check(lib().ctr_embed_bag_fwd(_ptr(tables, torch.float32, 'tables'), _ptr(row_offset), _ptr(idx),
                                  _ptr(bag_offsets), nnz, _ptr(x_cont, torch.float32, 'x_cont'),
                                  _ptr(cont_w, torch.float32, 'cont_w'), B, F, C_, E, zp, zld, int(bool(cont_tail)),
                                  fp, fld, _ptr(s_out, torch.float32, 's_out'), sp, sld, _dp(drop_first),
                                  _dp(drop_second), _ptr(oor_count, torch.int32, 'oor_count'), _stream()),
          'ctr_embed_bag_fwd')


@_timed
def embed_assemble_fwd(blocks, fbase, fstride, x_cont=None, cont_w=None, *, F, E, z_out, cont_tail=False,
                       first_out=None, s_out=None, second_out=None, drop_first=None, drop_second=None):
    """Z from per-owner row blocks (sharded tables) + continuous rows + FM sums, one pass (see embed_fwd)."""
    B = z_out.shape[0]
    C_ = 0 if x_cont is None else x_cont.shape[1]
    if fbase.numel() != F or fstride.numel() != F or fbase.dtype != torch.int64 or fstride.dtype != torch.int64:
        raise ValueError('fbase / fstride must be int64 [F]')
    N = F + (C_ if cont_w is not None else 0)
    zp, zld = _rows(z_out, 'z_out')
    if z_out.shape[1] < N * E + (C_ if cont_tail else 0):
        raise ValueError('z_out too small')
    fp, fld = (None, 0) if first_out is None else _rows(first_out, 'first_out')
    sp, sld = (None, 0) if second_out is None else _rows(second_out, 'second_out')
    if s_out is not None and (tuple(s_out.shape) != (B, E) or not s_out.is_contiguous()):
        raise ValueError('s_out must be contiguous [B,E]')
    check(lib().ctr_embed_assemble_fwd(_ptr(blocks, torch.float32), _ptr(fbase, torch.int64), _ptr(fstride, torch.int64),
                                       _ptr(x_cont, torch.float32), _ptr(cont_w, torch.float32), B, int(F), C_, E,
                                       zp, zld, int(bool(cont_tail)), fp, fld, _ptr(s_out, torch.float32), sp, sld,
                                       _dp(drop_first), _dp(drop_second), _stream()), 'ctr_embed_assemble_fwd')


@_timed
def fm_fwd(z, *, N, E, first_out=None, s_out=None, second_out=None, drop_first=None, drop_second=None):
    B = z.shape[0]
    zp, zld = _rows(z, 'z')
    if z.shape[1] < N * E:
        raise ValueError('z narrower than N*E')
    fp, fld = (None, 0) if first_out is None else _rows(first_out, 'first_out')
    sp, sld = (None, 0) if second_out is None else _rows(second_out, 'second_out')
    if first_out is not None and (first_out.shape[0] != B or first_out.shape[1] < N):
        raise ValueError('first_out too small')
    if second_out is not None and (second_out.shape[0] != B or second_out.shape[1] < E):
        raise ValueError('second_out too small')
    if s_out is not None and (tuple(s_out.shape) != (B, E) or not s_out.is_contiguous()):
        raise ValueError('s_out must be contiguous [B,E]')
    check(lib().ctr_fm_fwd(zp, zld, B, N, E, fp, fld, _ptr(s_out, torch.float32), sp, sld, _dp(drop_first),
                           _dp(drop_second), _stream()), 'ctr_fm_fwd')


@_timed
def relu_mask_apply(g, act, scale=1.0):
    rows, cols = g.shape
    gp, gld = _rows(g, 'g')
    ap, ald = _rows(act, 'act')
    if act.shape[0] != rows or act.shape[1] < cols:
        raise ValueError('relu_mask_apply shape mismatch')
    check(lib().ctr_relu_mask_apply(gp, gld, ap, ald, rows, cols, float(scale), _stream()), 'ctr_relu_mask_apply')


@_timed
def embed_onehot_fwd(tables, row_offset, idx, x_cont, *, E, z_out, cont_tail=False, pos_out=None):
    B, F = idx.shape
    C_ = 0 if x_cont is None else x_cont.shape[1]
    if idx.dtype != torch.int32 or not idx.is_contiguous():
        raise TypeError('idx must be contiguous int32 [B,F]')
    zp, zld = _rows(z_out, 'z_out')
    if pos_out is not None and (pos_out.dtype != torch.int32 or tuple(pos_out.shape) != (B, F)):
        raise ValueError('pos_out must be int32 [B,F]')
    check(lib().ctr_embed_onehot_fwd(_ptr(tables, torch.float32), _ptr(row_offset, torch.int64), _ptr(idx),
                                     _ptr(x_cont, torch.float32), B, F, C_, E, zp, zld,
                                     int(bool(cont_tail)), _ptr(pos_out), _stream()), 'ctr_embed_onehot_fwd')


@_timed
def embed_bwd_dz(dz, *, B, N, E, gz=None, g_first=None, g_second=None, s=None, z=None, g_sum=None,
                 drop_first=None, drop_second=None, blocks_out=None, fbase=None, fstride=None, F=0):
    dzp, dzld = _rows(dz, 'dz')
    gzp, gzld = (None, 0) if gz is None else _rows(gz, 'gz')
    g1p, g1ld = (None, 0) if g_first is None else _rows(g_first, 'g_first')
    g2p, g2ld = (None, 0) if g_second is None else _rows(g_second, 'g_second')
    zp, zld = (None, 0) if z is None else _rows(z, 'z')
    for t, w, nm in ((dz, N * E, 'dz'), (gz, N * E, 'gz'), (g_first, N, 'g_first'), (g_second, E, 'g_second'),
                     (z, N * E, 'z')):
        if t is not None and (t.shape[0] != B or t.shape[1] < w):
            raise ValueError('%s too small' % nm)
    if g_sum is not None and (tuple(g_sum.shape) != (B, E) or not g_sum.is_contiguous()):
        raise ValueError('g_sum must be contiguous [B,E]')
    if blocks_out is not None:     # rows of the sharded fields go straight into the exchange buffer
        check(lib().ctr_embed_bwd_dz_blocks(gzp, gzld, g1p, g1ld, g2p, g2ld, _ptr(s, torch.float32), zp, zld,
                                            _ptr(g_sum, torch.float32), B, N, E, _dp(drop_first), _dp(drop_second),
                                            dzp, dzld, _ptr(blocks_out, torch.float32), _ptr(fbase, torch.int64),
                                            _ptr(fstride, torch.int64), int(F), _stream()), 'ctr_embed_bwd_dz_blocks')
        return
    check(lib().ctr_embed_bwd_dz(gzp, gzld, g1p, g1ld, g2p, g2ld, _ptr(s, torch.float32), zp, zld,
                                 _ptr(g_sum, torch.float32), B, N, E,
                                 _dp(drop_first), _dp(drop_second), dzp, dzld, _stream()), 'ctr_embed_bwd_dz')


# ------------------------------------------------------------------ sort + sparse update
@_timed
def emb_sort(idx, row_offset, *, N, total_rows, rmode, keys_out, src_out, ws):
    B, F = idx.shape
    n = B * F
    if keys_out.numel() < n or src_out.numel() < n:
        raise ValueError('sort outputs too small')
    wp, wn = ws.get(lib().ctr_emb_sort_workspace_bytes_fields(B, F, int(total_rows), int(bool(rmode))))
    check(lib().ctr_emb_sort(_ptr(idx, torch.int32), _ptr(row_offset, torch.int64), B, F, N, int(total_rows),
                             int(bool(rmode)), _ptr(keys_out), _ptr(src_out), wp, wn, _stream()), 'ctr_emb_sort')


@_timed
def emb_bag_sort(idx, bag_offsets, row_offset, *, B, F, N, total_rows, keys_out, src_out, ws):
    """(key, src) pairs of multi-hot bags, sorted by key: one pair per member, src = the bag's row of dZ."""
    nnz = idx.numel()
    if bag_offsets.dtype != torch.int32 or bag_offsets.numel() != B * F + 1:
        raise ValueError('bag_offsets must be int32 [B*F+1]')
    if keys_out.numel() < nnz or src_out.numel() < nnz:
        raise ValueError('sort outputs too small')
    wp, wn = ws.get(lib().ctr_emb_sort_workspace_bytes(max(nnz, 1)))
    check(lib().ctr_emb_bag_sort(_ptr(idx, torch.int32), _ptr(bag_offsets, torch.int32), nnz,
                                 _ptr(row_offset, torch.int64), B, F, N, int(total_rows), _ptr(keys_out),
                                 _ptr(src_out), wp, wn, _stream()), 'ctr_emb_bag_sort')


@_timed
def sort_pairs(keys, vals, key_bits, ws):
    n = keys.numel()
    ko = torch.empty_like(keys)
    vo = torch.empty_like(vals)
    wp, wn = ws.get(lib().ctr_emb_sort_workspace_bytes(n))
    check(lib().ctr_sort_pairs(_ptr(keys), _ptr(vals), n, int(key_bits), _ptr(ko), _ptr(vo), wp, wn, _stream()),
          'ctr_sort_pairs')
    return ko, vo


def unit_bwd_supported(K):
    return bool(lib().ctr_unit_bwd_supported(int(K)))


@_timed
def unit_bwd(dl, w, x, dx, dw, db, *, mask_from=-1, mask_scale=1.0, keep=None, masked_amax=None, ws):
    """Backward of the output unit y = x w + b: dx = dl w^T (relu/dropout-masked from column mask_from on),
    dw = x^T dl, db = sum dl, one pass over x.  keep (optional): (cols_a, drop_a, cols_b, drop_b) - the keep masks
    of two dropouts on dx's first cols_a columns and the cols_b after them (as dropout_apply2 afterwards).
    masked_amax (optional): a record (AMAX_FLOATS floats) that receives max |dx| over the masked columns."""
    B, K = x.shape
    if dl.numel() != B or w.numel() != K or dw.numel() != K or tuple(dx.shape) != (B, K):
        raise ValueError('unit_bwd: shapes disagree')
    xp, ldx = _rows(x, 'x')
    dp, ldd = _rows(dx, 'dx')
    wp, wn = ws.get(lib().ctr_unit_bwd_workspace_bytes(B, K))
    ca, da, cb, db_ = keep if keep is not None else (0, None, 0, None)
    check(lib().ctr_unit_bwd_keep(B, K, _ptr(dl, torch.float32), _ptr(w, torch.float32), xp, ldx, dp, ldd, int(mask_from),
                                  float(mask_scale), int(ca), C.byref(da) if da is not None else None, int(cb),
                                  C.byref(db_) if db_ is not None else None, _amax_rec(masked_amax, 'masked_amax'),
                                  _ptr(dw, torch.float32), _ptr(db, torch.float32), wp, wn, _stream()),
          'ctr_unit_bwd_keep')


def mlp_chain_supported(K0, widths):
    n3 = widths[2] if len(widths) > 2 else 0
    return len(widths) in (2, 3) and bool(lib().ctr_mlp_chain_supported(int(K0), int(widths[0]), int(widths[1]), int(n3)))


@_timed
def mlp_chain_fwd(x, Ws, bs, drops, outs):
    """outs[i] = dropout_i(relu(outs[i-1] Ws[i] + bs[i])) for two or three layers, outs[-1] = x."""
    M, K0 = x.shape
    xp, ldx = _rows(x, 'x')
    n = len(Ws)
    args = []
    for i in range(3):
        if i < n:
            W, b, o = Ws[i], bs[i], outs[i]
            kin = K0 if i == 0 else Ws[i - 1].shape[1]
            if tuple(W.shape) != (kin, o.shape[1]) or not W.is_contiguous() or b.numel() != W.shape[1] or o.shape[0] != M:
                raise ValueError('mlp_chain_fwd: layer %d shapes disagree' % i)
            op, old = _rows(o, 'out')
            args += [int(W.shape[1]), _ptr(W, torch.float32), _ptr(b, torch.float32), _dp(drops[i]), op, old]
        else:
            args += [0, None, None, None, None, 0]
    check(lib().ctr_mlp_chain_fwd(M, K0, xp, ldx, *args, _stream()), 'ctr_mlp_chain_fwd')


def mlp_chain_bwd_supported(w_in, w_mid, w_out):
    return bool(lib().ctr_mlp_chain_bwd_supported(int(w_in), int(w_mid), int(w_out)))


@_timed
def mlp_chain_bwd(g_in, Wb, h_mid, scale_mid, g_mid, Wa, h_out, scale_out, g_out, g_out_amax=None):
    """g_mid = (g_in Wb^T) masked by h_mid, g_out = (g_mid Wa^T) masked by h_out (relu/dropout masks).
    g_out_amax: optional largest-magnitude record of g_out (zeroed by the caller)."""
    M, n_in = g_in.shape
    mid, out = Wb.shape[0], Wa.shape[0]
    if Wb.shape[1] != n_in or Wa.shape[1] != mid or not Wb.is_contiguous() or not Wa.is_contiguous():
        raise ValueError('mlp_chain_bwd: weight shapes disagree')
    for t, wdt in ((h_mid, mid), (g_mid, mid), (h_out, out), (g_out, out)):
        if tuple(t.shape) != (M, wdt):
            raise ValueError('mlp_chain_bwd: activation shapes disagree')
    gp, ldgi = _rows(g_in, 'g_in')
    hm, ldhm = _rows(h_mid, 'h_mid')
    gm, ldgm = _rows(g_mid, 'g_mid')
    ho, ldho = _rows(h_out, 'h_out')
    go, ldgo = _rows(g_out, 'g_out')
    check(lib().ctr_mlp_chain_bwd(M, n_in, gp, ldgi, mid, _ptr(Wb, torch.float32), hm, ldhm, float(scale_mid), gm, ldgm,
                                  out, _ptr(Wa, torch.float32), ho, ldho, float(scale_out), go, ldgo,
                                  _amax_rec(g_out_amax, 'g_out_amax'), _stream()), 'ctr_mlp_chain_bwd')


@_timed
def tail_bwd_fused(g_in, Wb, h_mid, scale_mid, Wa, h_out, scale_out, g_out, dWa, dba, dWb, dbb, *, ws):
    """mlp_chain_bwd (without writing the mid gradient) plus both layers' weight / bias gradients in one pass:
    dWb = h_mid^T g_in, dbb = colsum(g_in), dWa = h_out^T g_mid, dba = colsum(g_mid)."""
    M, n_in = g_in.shape
    mid, out = Wb.shape[0], Wa.shape[0]
    if Wb.shape[1] != n_in or Wa.shape[1] != mid or not Wb.is_contiguous() or not Wa.is_contiguous():
        raise ValueError('tail_bwd_fused: weight shapes disagree')
    for t, wdt in ((h_mid, mid), (h_out, out), (g_out, out)):
        if tuple(t.shape) != (M, wdt):
            raise ValueError('tail_bwd_fused: activation shapes disagree')
    if dWa.numel() != out * mid or dWb.numel() != mid * n_in or dba.numel() != mid or dbb.numel() != n_in \
            or not dWa.is_contiguous() or not dWb.is_contiguous():
        raise ValueError('tail_bwd_fused: gradient shapes disagree')
    gp, ldgi = _rows(g_in, 'g_in')
    hm, ldhm = _rows(h_mid, 'h_mid')
    ho, ldho = _rows(h_out, 'h_out')
    go, ldgo = _rows(g_out, 'g_out')
    wp, wn = ws.get(lib().ctr_tail_bwd_fused_workspace_bytes(M, n_in, mid, out))
    check(lib().ctr_tail_bwd_fused(M, n_in, gp, ldgi, mid, _ptr(Wb, torch.float32), hm, ldhm, float(scale_mid), out,
                                   _ptr(Wa, torch.float32), ho, ldho, float(scale_out), go, ldgo,
                                   _ptr(dWa, torch.float32), _ptr(dba, torch.float32), _ptr(dWb, torch.float32),
                                   _ptr(dbb, torch.float32), wp, wn, _stream()), 'ctr_tail_bwd_fused')


@_timed
def auc_logloss(score, label, ws):
    """(auc, logloss) device scalars of float32 scores / labels (AUC with average ranks for ties)."""
    n = score.numel()
    if label.numel() != n or not score.is_contiguous() or not label.is_contiguous():
        raise ValueError('auc_logloss wants contiguous score and label of equal length')
    out = torch.empty(2, dtype=torch.float32, device=score.device)
    wp, wn = ws.get(lib().ctr_auc_workspace_bytes(n))
    check(lib().ctr_auc(_ptr(score, torch.float32), _ptr(label, torch.float32), n, C.c_void_p(out.data_ptr()),
                        C.c_void_p(out.data_ptr() + 4), wp, wn, _stream()), 'ctr_auc')
    return out[0], out[1]


@_timed
def shard_route(idx, row_offset, local_offset, W, dest_out, lkey_out, oor_count=None):
    """oor_count (optional device int32[1]): ids outside their table are added to it."""
    B, F = idx.shape
    if idx.dtype != torch.int32 or local_offset.dtype != torch.int32 or tuple(local_offset.shape) != (W, F):
        raise TypeError('shard_route wants int32 idx [B,F] and int32 local_offset [W,F]')
    if dest_out.numel() < B * F or lkey_out.numel() < B * F or dest_out.dtype != torch.int32 or lkey_out.dtype != torch.int32:
        raise ValueError('shard_route outputs must be int32 with B*F elements')
    check(lib().ctr_shard_route(_ptr(idx), _ptr(row_offset, torch.int64), _ptr(local_offset), B, F, int(W),
                                _ptr(dest_out), _ptr(lkey_out), _ptr(oor_count), _stream()), 'ctr_shard_route')


def shard_bucket(dest_sorted, perm, lkey, n, W, cap, slot_key, slot_perm, overflow, z_zero=None, F=1, E=1):
    """Fixed-capacity send buckets (cap slots per destination rank) from lookups sorted by destination.
    z_zero (optional): the [B, >= F*E] activation buffer the rows are scattered into later; the row of a lookup that
    does not fit its bucket is zeroed there."""
    for t in (dest_sorted, perm, lkey, slot_key, slot_perm, overflow):
        if t.dtype != torch.int32:
            raise TypeError('shard_bucket moves int32 words')
    if slot_key.numel() < W * cap or slot_perm.numel() < W * cap or dest_sorted.numel() < n or perm.numel() < n:
        raise ValueError('shard_bucket: buffers too small')
    zp, zld = (None, 0) if z_zero is None else _rows(z_zero, 'z_zero')
    check(lib().ctr_shard_bucket(_ptr(dest_sorted), _ptr(perm), _ptr(lkey), int(n), int(W), int(cap), _ptr(slot_key),
                                 _ptr(slot_perm), _ptr(overflow), zp, zld, int(F), int(E), _stream()),
          'ctr_shard_bucket')


def _words(t):
    if t.element_size() != 4:
        raise TypeError('rows_gather / rows_scatter move 32-bit words')
    return t


@_timed
def rows_gather(src, perm, n, out, *, F, E):
    """out[s, :E] = src[b, f*E:(f+1)*E] with (b, f) = divmod(perm[s], F)."""
    sp, ld = _rows(_words(src), 'src', dtype=None)
    if perm.dtype != torch.int32 or perm.numel() < n or out.numel() < n * E or not out.is_contiguous():
        raise ValueError('rows_gather: perm must be int32 [n], out contiguous with n*E words')
    check(lib().ctr_rows_gather(sp, ld, _ptr(perm), int(n), int(F), int(E), _ptr(_words(out)), _stream()),
          'ctr_rows_gather')


@_timed
def rows_scatter(rows, perm, n, dst, *, F, E):
    """dst[b, f*E:(f+1)*E] = rows[s, :E] with (b, f) = divmod(perm[s], F)."""
    dp, ld = _rows(_words(dst), 'dst', dtype=None)
    if perm.dtype != torch.int32 or perm.numel() < n or rows.numel() < n * E or not rows.is_contiguous():
        raise ValueError('rows_scatter: perm must be int32 [n], rows contiguous with n*E words')
    check(lib().ctr_rows_scatter(_ptr(_words(rows)), _ptr(perm), int(n), int(F), int(E), dp, ld, _stream()),
          'ctr_rows_scatter')


def set_step_words(seed_add, seed_add_value, lr_dev, lr_value):
    """Stream-ordered write of the device words make_dropout(seed_add=) / make_opt(lr_dev=) point at."""
    check(lib().ctr_set_step_words(None if seed_add is None else seed_add.data_ptr(), int(seed_add_value),
                                   None if lr_dev is None else lr_dev.data_ptr(), float(lr_value), _stream()),
          'ctr_set_step_words')


def make_opt(kind, lr=0.0, beta1=0.9, beta2=0.999, eps=1e-8, l1=0.0, l2=0.0, lr_dev=None):
    """lr_dev: optional device float32[1] read by the kernels instead of lr."""
    return Opt(kind=int(kind), lr=float(lr), beta1=float(beta1), beta2=float(beta2), eps=float(eps),
               l1=float(l1), l2=float(l2), lr_dev=None if lr_dev is None else lr_dev.data_ptr())


def emb_sparse_apply_fm_supported(E, g2):
    """Whether emb_sparse_apply takes fm_g2 = g2 (rows of at most 32 floats, 8-byte aligned rows of g2)."""
    return (E <= 32 and E % 4 == 0 and g2.dim() == 2 and g2.shape[1] == E and g2.stride(1) == 1 and g2.stride(0) % 2 == 0
            and g2.data_ptr() % 8 == 0 and g2.dtype == torch.float32)


@_timed
def emb_sparse_apply(keys, src, n_items, *, N, E, total_rows, dz, opt, ws, tables=None, slot0=None, slot1=None,
                     src_scale=None, fm_g2=None):
    """slot0 / slot1: [total_rows, E], contiguous or both column blocks of one wider row-major buffer (the same row
    pitch: Adam's m | v side by side, engine.EmbeddingTables).
    fm_g2 (optional [B, E] view): dz was formed without the FM second-order term's product with the looked-up row
    (gemm fm_bwd with z None); a run's gradient is then sum(dz) - sum(fm_g2[sample]) * tables[key]
    (ctr_emb_sparse_apply_fm; see emb_sparse_apply_fm_supported)."""
    dzp, dzld = _rows(dz, 'dz')
    if tables is not None and (tables.numel() < total_rows * E or not tables.is_contiguous()):
        raise ValueError('tables must be contiguous [total_rows, E]')
    slot_ld = E
    for t, nm in ((slot0, 'slot0'), (slot1, 'slot1')):
        if t is None:
            continue
        if t.dim() != 2 or t.shape[0] < total_rows or t.shape[1] != E or t.stride(1) != 1 or t.stride(0) % 4:
            raise ValueError('%s must be [total_rows, E] with unit column stride and a row pitch of whole float4s' % nm)
        if t is slot0:
            slot_ld = t.stride(0)
        elif slot0 is not None and t.stride(0) != slot_ld:
            raise ValueError('slot0 and slot1 must share one row pitch')
    if dz.shape[1] < N * E:
        raise ValueError('dz narrower than N*E')
    if fm_g2 is not None:
        if src_scale is not None or tables is None or not emb_sparse_apply_fm_supported(E, fm_g2):
            raise ValueError('fm_g2 needs tables, no src_scale, E <= 32 and an 8-byte aligned [B, E] view with an even row pitch')
        g2p, g2ld = _rows(fm_g2, 'fm_g2')
        wp, wn = ws.get(lib().ctr_emb_sparse_apply_fm_workspace_bytes(int(n_items), E))
        check(lib().ctr_emb_sparse_apply_fm(_ptr(keys), _ptr(src), int(n_items), N, E, int(total_rows), dzp, dzld, g2p, g2ld,
                                            _ptr(tables, torch.float32), _ptr(slot0, torch.float32),
                                            _ptr(slot1, torch.float32), int(slot_ld), C.byref(opt), wp, wn, _stream()),
              'ctr_emb_sparse_apply_fm')
        return
    wp, wn = ws.get(lib().ctr_emb_sparse_apply_workspace_bytes(int(n_items), E))
    check(lib().ctr_emb_sparse_apply_ld(_ptr(keys), _ptr(src), int(n_items), N, E, int(total_rows), dzp, dzld,
                                        _ptr(src_scale, torch.float32), _ptr(tables, torch.float32),
                                        _ptr(slot0, torch.float32), _ptr(slot1, torch.float32), int(slot_ld),
                                        C.byref(opt), wp, wn, _stream()), 'ctr_emb_sparse_apply')


@_timed
def adam_untouched_rows(keys, n_items, *, total_rows, E, opt, tables, slot0, slot1, touched):
    """tf.train.AdamOptimizer's step on the table rows that received no gradient this step (m, v decay and the row
    still moves); the rows the sorted `keys` name were updated by emb_sparse_apply and are skipped.  touched: uint8
    [total_rows]; bit 0 is clear on entry and on return, bit 1 marks the rows that have optimizer state (set here on
    the rows `keys` names; rows whose byte is 0 must have m = v = 0 and are not read - fill the map with 2 after
    writing m or v by other means)."""
    for t in (tables, slot0, slot1):
        if not t.is_contiguous() or t.numel() < total_rows * E:
            raise ValueError('adam_untouched_rows takes contiguous [total_rows, E] tensors')
    if touched.dtype != torch.uint8 or touched.numel() < total_rows:
        raise ValueError('touched must be uint8 [total_rows]')
    kp, tp = _ptr(keys), C.c_void_p(touched.data_ptr())
    check(lib().ctr_mark_rows(kp, int(n_items), int(total_rows), tp, 1, _stream()), 'ctr_mark_rows')
    check(lib().ctr_adam_untouched_rows(_ptr(tables, torch.float32), _ptr(slot0, torch.float32), _ptr(slot1, torch.float32),
                                        int(total_rows), E, tp, C.byref(opt), _stream()), 'ctr_adam_untouched_rows')


# ------------------------------------------------------------------ dense pieces
GEMM_MODES = {'fp32': 0, 'split': 1, 'bf16': 2}


def set_gemm_mode(name):
    """Arithmetic of the tiled GEMM, process-wide: 'split' (default: fp32 result from six bf16 MFMAs per product),
    'fp32' (the fp32 MFMA) or 'bf16' (operands rounded to bf16, one MFMA per product, fp32 accumulate - reduced
    precision, opt-in).  Returns the previous mode's name."""
    prev = get_gemm_mode()
    check(lib().ctr_set_gemm_mode(GEMM_MODES[name]), 'ctr_set_gemm_mode')
    return prev


def get_gemm_mode():
    m = int(lib().ctr_get_gemm_mode())
    return [k for k, v in GEMM_MODES.items() if v == m][0]


@_timed
def gemm(A, B_, Cout, *, trans_a=False, trans_b=False, bias=None, relu=False, sigmoid=False, accumulate=False,
         dropout=None, mask_src=None, mask_scale=1.0, fm_bwd=None, ws=None, a_amax=None, b_amax=None,
         a_amax_floor=0.0, b_amax_floor=0.0, amax_out=None):
    """Cout[M,N] = epilogue(op(A) @ op(B_)); all three are 2-D views with contiguous last dim.
    fm_bwd (optional): dict(N, E, g_first, g_second, s, z) - the FM terms of embed_bwd_dz are added to the first N*E
    columns of the product on its way out (ctr_fm_bwd_t; g_first / g_second already carry their keep masks); z None
    leaves the term with the looked-up row to emb_sparse_apply(fm_g2=...).
    a_amax / b_amax (optional largest-magnitude records of A and B, AMAX_FLOATS floats each): with both, the product runs as
    the two-term fp16 split; amax_out: raised to max |Cout| (zero it first)."""
    M, N = Cout.shape
    ap, lda = _rows(A, 'A')
    bp, ldb = _rows(B_, 'B')
    cp, ldc = _rows(Cout, 'C')
    K = A.shape[0] if trans_a else A.shape[1]
    am = A.shape[1] if trans_a else A.shape[0]
    bk, bn = (B_.shape[1], B_.shape[0]) if trans_b else (B_.shape[0], B_.shape[1])
    if am != M or bk != K or bn != N:
        raise ValueError('gemm shape mismatch: op(A)=%dx%d op(B)=%dx%d C=%dx%d' % (am, K, bk, bn, M, N))
    if bias is not None and bias.numel() != N:
        raise ValueError('bias must have N elements')
    mp, mld = (None, 0)
    if mask_src is not None:
        mp, mld = _rows(mask_src, 'mask_src')
        if mask_src.shape[0] != M or mask_src.shape[1] < N:
            raise ValueError('mask_src shape mismatch')
    epi = Epilogue(bias=None if bias is None else bias.data_ptr(), relu=2 if sigmoid else int(relu), accumulate=int(accumulate),
                   dropout=C.pointer(dropout) if dropout is not None else None,
                   mask_src=None if mask_src is None else mask_src.data_ptr(), mask_ld=mld,
                   mask_scale=float(mask_scale),
                   a_amax=None if a_amax is None else _amax_rec(a_amax, 'a_amax').value,
                   b_amax=None if b_amax is None else _amax_rec(b_amax, 'b_amax').value,
                   a_amax_floor=float(a_amax_floor), b_amax_floor=float(b_amax_floor),
                   c_amax_out=None if amax_out is None else _amax_rec(amax_out, 'amax_out').value)
    if fm_bwd is not None:
        g1, g2 = fm_bwd.get('g_first'), fm_bwd.get('g_second')
        g1p, g1ld = _rows(g1, 'g_first') if g1 is not None else (None, 0)
        g2p, g2ld = _rows(g2, 'g_second') if g2 is not None else (None, 0)
        zp, zld = _rows(fm_bwd['z'], 'z') if g2 is not None and fm_bwd.get('z') is not None else (None, 0)
        fm = FmBwd(g_first=g1p, g_first_ld=g1ld, g_second=g2p, g_second_ld=g2ld,
                   s=_ptr(fm_bwd['s'], torch.float32) if g2 is not None else None, z=zp, z_ld=zld,
                   n_rows=int(fm_bwd['N']), E=int(fm_bwd['E']))
        epi.fm_bwd = C.pointer(fm)
    need = lib().ctr_gemm_f32_workspace_bytes(M, N, K)
    wp, wn = (None, 0)
    if need:
        wp, wn = ws.get(need)
    check(lib().ctr_gemm_f32(int(trans_a), int(trans_b), M, N, K, ap, lda, bp, ldb, cp, ldc, C.byref(epi), wp, wn,
                             _stream()), 'ctr_gemm_f32')


@_timed
def dense_wgrad(x, g, dW, db=None, *, ws, x_amax=None, g_amax=None, x_amax_floor=0.0):
    """dW[M,N] = x^T g (x: [K,M] view, g: [K,N] view), db[N] = column sums of g.  x_amax / g_amax: as a_amax / b_amax
    of gemm()."""
    K, M = x.shape
    N = g.shape[1]
    xp, lda = _rows(x, 'x')
    gp, ldb = _rows(g, 'g')
    if g.shape[0] != K or dW.numel() != M * N or not dW.is_contiguous() or (db is not None and db.numel() != N):
        raise ValueError('dense_wgrad shape mismatch')
    wp, wn = ws.get(lib().ctr_dense_wgrad_workspace_bytes(M, N, K))
    check(lib().ctr_dense_wgrad_scaled(M, N, K, xp, lda, gp, ldb, _ptr(dW, torch.float32), _ptr(db, torch.float32),
                                       _amax_rec(x_amax, 'x_amax'), float(x_amax_floor), _amax_rec(g_amax, 'g_amax'),
                                       wp, wn, _stream()), 'ctr_dense_wgrad')


AMAX_FLOATS = 1024      # a largest-magnitude record (CTR_AMAX_FLOATS): its value is record.max()


def _amax_rec(t, name='amax'):
    if t is None:
        return None
    if t.numel() < AMAX_FLOATS or not t.is_contiguous():
        raise ValueError('%s must be a contiguous record of %d floats' % (name, AMAX_FLOATS))
    return _ptr(t, torch.float32, name)


def absmax(x, out):
    """Raises the record `out` to max |x| over a 2-D view (zero `out` first: it is only ever raised)."""
    xp, ld = _rows(x, 'x')
    _amax_rec(out)
    check(lib().ctr_absmax(xp, ld, x.shape[0], x.shape[1], _ptr(out, torch.float32), _stream()), 'ctr_absmax')


def absmax_tensors(flat, offsets, counts, out):
    """Record i of `out` (AMAX_FLOATS floats each) is raised to max |flat[offsets[i] : offsets[i] + counts[i]]|;
    offsets / counts: device int64."""
    n = offsets.numel()
    if counts.numel() != n or out.numel() < n * AMAX_FLOATS:
        raise ValueError('absmax_tensors: offsets, counts and out disagree')
    check(lib().ctr_absmax_tensors(_ptr(flat, torch.float32), _ptr(offsets, torch.int64), _ptr(counts, torch.int64), n,
                                   _ptr(out, torch.float32), _stream()), 'ctr_absmax_tensors')


def set_gemm_f16_split(on):
    """Two-term fp16 split for products that bring both operands' maxima (default on); returns the previous setting."""
    return bool(lib().ctr_set_gemm_f16_split(int(bool(on))))


def tail_wgrad2_supported(ka, na, kb, nb):
    return bool(lib().ctr_tail_wgrad2_supported(int(ka), int(na), int(kb), int(nb)))


@_timed
def tail_wgrad2(xa, ga, dWa, dba, xb, gb, dWb, dbb, *, ws):
    """Weight / bias gradients of two narrow layers in one launch: dWa = xa^T ga, dba = colsum(ga), dWb = xb^T gb,
    dbb = colsum(gb).  xa, ga, xb: 16-byte aligned row views; gb may be any column slice."""
    M, ka = xa.shape
    na, kb, nb = ga.shape[1], xb.shape[1], gb.shape[1]
    if ga.shape[0] != M or xb.shape[0] != M or gb.shape[0] != M or dWa.numel() != ka * na or dWb.numel() != kb * nb \
            or dba.numel() != na or dbb.numel() != nb or not dWa.is_contiguous() or not dWb.is_contiguous():
        raise ValueError('tail_wgrad2 shape mismatch')
    xap, lxa = _rows(xa, 'xa')
    gap, lga = _rows(ga, 'ga')
    xbp, lxb = _rows(xb, 'xb')
    gbp, lgb = _rows(gb, 'gb')
    wp, wn = ws.get(lib().ctr_tail_wgrad2_workspace_bytes(M, ka, na, kb, nb))
    check(lib().ctr_tail_wgrad2(M, ka, xap, lxa, na, gap, lga, _ptr(dWa, torch.float32), _ptr(dba, torch.float32), kb, xbp,
                                lxb, nb, gbp, lgb, _ptr(dWb, torch.float32), _ptr(dbb, torch.float32), wp, wn, _stream()),
          'ctr_tail_wgrad2')


@_timed
def colsum(X, out, *, w=None, w_group=1, accumulate=False, ws):
    B, K = X.shape
    xp, ld = _rows(X, 'X')
    wp_, wld = (None, 0) if w is None else _rows(w, 'w')
    if out.numel() < K:
        raise ValueError('colsum output too small')
    if w is not None and (w.shape[0] != B or w.shape[1] * w_group < K):
        raise ValueError('colsum weight shape mismatch')
    wsp, wsn = ws.get(lib().ctr_colsum_workspace_bytes(B, K))
    check(lib().ctr_colsum(xp, ld, wp_, wld, int(w_group), B, K, _ptr(out, torch.float32), int(accumulate), wsp, wsn,
                           _stream()), 'ctr_colsum')


@_timed
def rowdot_fwd(X, w, bias, out):
    B, K = X.shape
    xp, ld = _rows(X, 'X')
    if w.numel() != K or out.numel() < B:
        raise ValueError('rowdot shape mismatch')
    check(lib().ctr_rowdot_fwd(xp, ld, _ptr(w, torch.float32), _ptr(bias, torch.float32), B, K,
                               _ptr(out, torch.float32), _stream()), 'ctr_rowdot_fwd')


@_timed
def rowdot_bwd_x(g, w, dX, accumulate=False):
    B, K = dX.shape
    dp, ld = _rows(dX, 'dX')
    if w.numel() != K or g.numel() < B:
        raise ValueError('rowdot_bwd shape mismatch')
    check(lib().ctr_rowdot_bwd_x(_ptr(g, torch.float32), _ptr(w, torch.float32), B, K, dp, ld, int(accumulate),
                                 _stream()), 'ctr_rowdot_bwd_x')


@_timed
def loss_head(kind, x, y, *, y_out=None, loss_out=None, dx=None, ws=None, grad_scale=1.0):
    B = x.numel()
    if y is not None and y.numel() != B:
        raise ValueError('loss_head: label count mismatch')
    wp, wn = (None, 0)
    if loss_out is not None:
        wp, wn = ws.get(lib().ctr_loss_head_workspace_bytes(B))
    check(lib().ctr_loss_head(int(kind), _ptr(x, torch.float32), _ptr(y, torch.float32), B, float(grad_scale),
                              _ptr(y_out, torch.float32), _ptr(loss_out, torch.float32), _ptr(dx, torch.float32),
                              wp, wn, _stream()), 'ctr_loss_head')


@_timed
def esmm_head(ctr_logit, cvr_logit, y, *, y_out=None, loss_out=None, dctr=None, dcvr=None, ws=None,
              grad_scale=1.0):
    B = ctr_logit.numel()
    if cvr_logit.numel() != B or (y is not None and y.numel() != B):
        raise ValueError('esmm_head size mismatch')
    wp, wn = (None, 0)
    if loss_out is not None:
        wp, wn = ws.get(lib().ctr_loss_head_workspace_bytes(B))
    f32 = torch.float32
    check(lib().ctr_esmm_head(_ptr(ctr_logit, f32), _ptr(cvr_logit, f32), _ptr(y, f32), B, float(grad_scale),
                              _ptr(y_out, f32),
                              _ptr(loss_out, f32), _ptr(dctr, f32), _ptr(dcvr, f32), wp, wn, _stream()),
          'ctr_esmm_head')


@_timed
def dense_update(param, grad, opt, slot0=None, slot1=None):
    n = param.numel()
    if grad.numel() != n or (slot0 is not None and slot0.numel() != n) or (slot1 is not None and slot1.numel() != n):
        raise ValueError('dense_update size mismatch')
    check(lib().ctr_dense_update(_ptr(param, torch.float32), _ptr(grad, torch.float32), _ptr(slot0, torch.float32),
                                 _ptr(slot1, torch.float32), n, C.byref(opt), _stream()), 'ctr_dense_update')


@_timed
def cross_fwd(x0, w, b, xL, aux, ws):
    """w, b: [L, >= D] views (row stride = wld)."""
    B, D = xL.shape
    L = w.shape[0]
    xp, xld = _rows(x0, 'x0')
    op, old = _rows(xL, 'xL')
    wp_, wld = _rows(w, 'w')
    bp_, bld = _rows(b, 'b')
    if x0.shape[0] != B or x0.shape[1] < D or w.shape[1] < D or tuple(b.shape) != tuple(w.shape) or bld != wld:
        raise ValueError('cross_fwd shape mismatch')
    if aux is not None and aux.numel() < B * 2 * L:
        raise ValueError('cross aux too small')
    wp, wn = ws.get(lib().ctr_cross_fwd_workspace_bytes(D, L))
    check(lib().ctr_cross_fwd(xp, xld, wp_, bp_, wld, B, D, L, op, old,
                              _ptr(aux, torch.float32), wp, wn, _stream()), 'ctr_cross_fwd')


@_timed
def cross_bwd(x0, w, b, aux, g, dx0, dw, db, ws, accumulate=False):
    B, D = g.shape
    L = w.shape[0]
    xp, xld = _rows(x0, 'x0')
    gp, gld = _rows(g, 'g')
    dp, dld = _rows(dx0, 'dx0')
    wp_, wld = _rows(w, 'w')
    bp_, bld = _rows(b, 'b')
    dwp, dwld = _rows(dw, 'dw')
    dbp, dbld = _rows(db, 'db')
    if dx0.shape[0] != B or dx0.shape[1] < D or not (wld == bld == dwld == dbld) or dw.shape[0] != L:
        raise ValueError('cross_bwd shape mismatch')
    wp, wn = ws.get(lib().ctr_cross_bwd_workspace_bytes(B, D, L))
    check(lib().ctr_cross_bwd(xp, xld, wp_, bp_, wld, _ptr(aux, torch.float32),
                              gp, gld, B, D, L, dp, dld, int(accumulate), dwp, dbp, wp, wn, _stream()),
          'ctr_cross_bwd')


@_timed
def copy2d(src, dst, accumulate=False):
    rows, cols = src.shape
    sp, sld = _rows(src, 'src')
    dp, dld = _rows(dst, 'dst')
    if dst.shape[0] != rows or dst.shape[1] < cols:
        raise ValueError('copy2d shape mismatch')
    check(lib().ctr_copy2d(sp, sld, dp, dld, rows, cols, int(accumulate), _stream()), 'ctr_copy2d')


@_timed
def mul2d(a, b, out):
    rows, cols = out.shape
    ap, ald = _rows(a, 'a')
    bp, bld = _rows(b, 'b')
    op, old = _rows(out, 'out')
    if tuple(a.shape) != (rows, cols) or tuple(b.shape) != (rows, cols):
        raise ValueError('mul2d shape mismatch')
    check(lib().ctr_mul2d(ap, ald, bp, bld, op, old, rows, cols, _stream()), 'ctr_mul2d')


@_timed
def dropout_apply(x, drop):
    if drop is None:
        return
    rows, cols = x.shape
    xp, ld = _rows(x, 'x')
    check(lib().ctr_dropout_apply(xp, ld, rows, cols, C.byref(drop), _stream()), 'ctr_dropout_apply')


@_timed
def dropout_apply2(x, cols_a, drop_a, cols_b, drop_b):
    """x[:, :cols_a] under drop_a and x[:, cols_a:cols_a + cols_b] under drop_b in one launch."""
    if drop_a is None and drop_b is None:
        return
    xp, ld = _rows(x, 'x')
    check(lib().ctr_dropout_apply2(xp, ld, x.shape[0], int(cols_a), C.byref(drop_a) if drop_a is not None else None,
                                   int(cols_b), C.byref(drop_b) if drop_b is not None else None, _stream()),
          'ctr_dropout_apply2')


# ------------------------------------------------------------------ PNN / AFM interaction pieces
@_timed
def pnn_inner_fwd(z, w, usq, *, N, E):
    B = z.shape[0]
    P = w.shape[0]
    zp, zld = _rows(z, 'z')
    if tuple(w.shape) != (P, N) or not w.is_contiguous() or tuple(usq.shape) != (B, P) or not usq.is_contiguous():
        raise ValueError('pnn_inner_fwd shape mismatch')
    check(lib().ctr_pnn_inner_fwd(zp, zld, _ptr(w, torch.float32), B, N, E, P, _ptr(usq, torch.float32), _stream()),
          'ctr_pnn_inner_fwd')


@_timed
def pnn_inner_bwd(z, w, coef, dz, dw, *, N, E, ws):
    B = z.shape[0]
    P = w.shape[0]
    zp, zld = _rows(z, 'z')
    dp, dld = _rows(dz, 'dz')
    coef_ld = 0 if coef.dim() == 1 else int(coef.stride(0))
    if coef.dim() == 2 and (coef.shape[0] != B or coef.shape[1] < P):
        raise ValueError('coef must be [P] or [B,P]')
    if dw.numel() != P * N or not dw.is_contiguous():
        raise ValueError('dw must be contiguous [P,N]')
    wp, wn = ws.get(lib().ctr_pnn_inner_bwd_workspace_bytes(B, N, P))
    check(lib().ctr_pnn_inner_bwd(zp, zld, _ptr(w, torch.float32), _ptr(coef, torch.float32), coef_ld, B, N, E, P,
                                  dp, dld, _ptr(dw, torch.float32), wp, wn, _stream()), 'ctr_pnn_inner_bwd')


@_timed
def outer_fwd(s, ss):
    B, E = s.shape
    if tuple(ss.shape) != (B, E * E) or not ss.is_contiguous() or not s.is_contiguous():
        raise ValueError('outer_fwd shape mismatch')
    check(lib().ctr_outer_fwd(_ptr(s, torch.float32), B, E, _ptr(ss, torch.float32), _stream()), 'ctr_outer_fwd')


@_timed
def outer_bwd(dss, s, ds):
    B, E = s.shape
    if tuple(dss.shape) != (B, E * E) or not dss.is_contiguous() or tuple(ds.shape) != (B, E) or not ds.is_contiguous():
        raise ValueError('outer_bwd shape mismatch')
    check(lib().ctr_outer_bwd(_ptr(dss, torch.float32), _ptr(s, torch.float32), B, E, _ptr(ds, torch.float32),
                              _stream()), 'ctr_outer_bwd')


@_timed
def bias_act(x, bias1=None, bias2=None, relu=False, dropout=None):
    rows, cols = x.shape
    xp, ld = _rows(x, 'x')
    for b in (bias1, bias2):
        if b is not None and b.numel() != cols:
            raise ValueError('bias must have one value per column')
    check(lib().ctr_bias_act(xp, ld, rows, cols, _ptr(bias1, torch.float32), _ptr(bias2, torch.float32), int(relu),
                             _dp(dropout), _stream()), 'ctr_bias_act')


@_timed
def vec_op(kind, a, b, out):
    n = out.numel()
    if a.numel() != n or (b is not None and b.numel() != n):
        raise ValueError('vec_op size mismatch')
    check(lib().ctr_vec_op(int(kind), _ptr(a, torch.float32), _ptr(b, torch.float32), _ptr(out, torch.float32), n,
                           _stream()), 'ctr_vec_op')


f32 = torch.float32


@_timed
def afm_pair_fwd(z, x_out, *, N, E, pair_drop=None):
    B = z.shape[0]
    zp, zld = _rows(z, 'z')
    if tuple(x_out.shape) != (B, E) or not x_out.is_contiguous() or z.shape[1] < N * E:
        raise ValueError('afm_pair_fwd shape mismatch')
    check(lib().ctr_afm_pair_fwd(zp, zld, B, N, E, _dp(pair_drop), _ptr(x_out, torch.float32), _stream()),
          'ctr_afm_pair_fwd')


@_timed
def afm_pair_bwd(z, dx, dz, *, N, E, pair_drop=None):
    B = z.shape[0]
    zp, zld = _rows(z, 'z')
    dp, dld = _rows(dz, 'dz')
    if tuple(dx.shape) != (B, E) or not dx.is_contiguous() or dz.shape[1] < N * E:
        raise ValueError('afm_pair_bwd shape mismatch')
    check(lib().ctr_afm_pair_bwd(zp, zld, _ptr(dx, torch.float32), B, N, E, _dp(pair_drop), dp, dld, _stream()),
          'ctr_afm_pair_bwd')


def _afm_attn_check(z, W, b, h, N, E):
    A = W.shape[1]
    if tuple(W.shape) != (E, A) or b.numel() != A or h.numel() != A or z.shape[1] < N * E:
        raise ValueError('afm_attn: W must be [E,A], b and h [A], z at least N*E wide')
    for t in (W, b, h):
        if not t.is_contiguous():
            raise ValueError('afm_attn: parameters must be contiguous')
    return A


@_timed
def afm_attn_fwd(z, W, b, h, x_out, *, N, E, pair_drop=None):
    B = z.shape[0]
    A = _afm_attn_check(z, W, b, h, N, E)
    zp, zld = _rows(z, 'z')
    if tuple(x_out.shape) != (B, E) or not x_out.is_contiguous():
        raise ValueError('afm_attn_fwd: x_out must be contiguous [B,E]')
    check(lib().ctr_afm_attn_fwd(zp, zld, B, N, E, A, _ptr(W, f32), _ptr(b, f32), _ptr(h, f32), _dp(pair_drop),
                                 _ptr(x_out, f32), _stream()), 'ctr_afm_attn_fwd')


@_timed
def afm_attn_bwd(z, dx, W, b, h, dz, dW, db, dh, *, N, E, ws, pair_drop=None):
    B = z.shape[0]
    A = _afm_attn_check(z, W, b, h, N, E)
    zp, zld = _rows(z, 'z')
    dp, dld = _rows(dz, 'dz')
    if tuple(dx.shape) != (B, E) or not dx.is_contiguous() or dz.shape[1] < N * E:
        raise ValueError('afm_attn_bwd shape mismatch')
    if dW.numel() != E * A or db.numel() != A or dh.numel() != A:
        raise ValueError('afm_attn_bwd: gradient buffers must match W, b, h')
    for t in (dW, db, dh):
        if not t.is_contiguous():
            raise ValueError('afm_attn_bwd: gradient buffers must be contiguous')
    wp, wn = ws.get(lib().ctr_afm_attn_bwd_workspace_bytes(B, E, A))
    check(lib().ctr_afm_attn_bwd(zp, zld, _ptr(dx, f32), B, N, E, A, _ptr(W, f32), _ptr(b, f32), _ptr(h, f32),
                                 _dp(pair_drop), dp, dld, _ptr(dW, f32), _ptr(db, f32), _ptr(dh, f32), wp, wn,
                                 _stream()), 'ctr_afm_attn_bwd')


# ------------------------------------------------------------------ DIN pieces
@_timed
def din_build_idx(ids, cate_list, item_count, out):
    n = ids.numel()
    if ids.dtype != torch.int32 or cate_list.dtype != torch.int32 or out.dtype != torch.int32 or out.numel() < 2 * n:
        raise TypeError('din_build_idx wants int32 ids / cate_list / out[n,2]')
    if cate_list.numel() < item_count:
        raise ValueError('cate_list shorter than item_count')
    check(lib().ctr_din_build_idx(_ptr(ids), _ptr(cate_list), n, int(item_count), _ptr(out), _stream()),
          'ctr_din_build_idx')


@_timed
def din_concat_fwd(q, k, out, *, B, T, H):
    if q.numel() != B * H or k.numel() != B * T * H or out.numel() != B * T * 4 * H:
        raise ValueError('din_concat_fwd size mismatch')
    check(lib().ctr_din_concat_fwd(_ptr(q, f32), _ptr(k, f32), B, T, H, _ptr(out, f32), _stream()), 'ctr_din_concat_fwd')


@_timed
def din_concat_bwd(d_all, q, k, dq, dk, *, B, T, H, acc_q=False, acc_k=False):
    if d_all.numel() != B * T * 4 * H or dq.numel() != B * H or dk.numel() != B * T * H:
        raise ValueError('din_concat_bwd size mismatch')
    check(lib().ctr_din_concat_bwd(_ptr(d_all, f32), _ptr(q, f32), _ptr(k, f32), B, T, H, _ptr(dq, f32), int(acc_q),
                                   _ptr(dk, f32), int(acc_k), _stream()), 'ctr_din_concat_bwd')


def din_attn_mlp_supported(H, n1=80, n2=40):
    return bool(lib().ctr_din_attn_mlp_supported(int(H), int(n1), int(n2)))


def _sl_ptr(sl, B):
    if sl is None:
        return None
    if sl.dtype != torch.int32 or sl.numel() != B or not sl.is_contiguous():
        raise ValueError('sl must be a contiguous int32 [B] tensor')
    return C.c_void_p(sl.data_ptr())


@_timed
def din_attn_mlp_fwd(q, k, W1, b1, W2, b2, w3, b3, score, *, B, T, H, d1=None, d2=None, sl=None):
    """score[b,t] of the DIN attention network from q [B,H] and k [B*T,H]; d1 / d2 keep the sigmoid outputs.
    sl (sequence lengths, optional, the same for _bwd and _wgrad): strips of 16 positions from a sample's length on are
    not computed - the masked softmax never reads their scores and their gradients are zeros."""
    if q.numel() != B * H or k.numel() != B * T * H or score.numel() != B * T or tuple(W1.shape) != (4 * H, 80) \
            or tuple(W2.shape) != (80, 40) or w3.numel() != 40 or b1.numel() != 80 or b2.numel() != 40 or b3.numel() != 1:
        raise ValueError('din_attn_mlp_fwd size mismatch')
    for t_, n in ((d1, 80), (d2, 40)):
        if t_ is not None and (t_.numel() != B * T * n or not t_.is_contiguous()):
            raise ValueError('din_attn_mlp_fwd: d1 / d2 must be contiguous [B*T, 80] / [B*T, 40]')
    for t_ in (q, k, W1, W2):
        if not t_.is_contiguous():
            raise ValueError('din_attn_mlp_fwd: contiguous inputs required')
    check(lib().ctr_din_attn_mlp_fwd_sl(_ptr(q, f32), _ptr(k, f32), _sl_ptr(sl, B), B, T, H, _ptr(W1, f32), _ptr(b1, f32),
                                        _ptr(W2, f32), _ptr(b2, f32), _ptr(w3, f32), _ptr(b3, f32), _ptr(d1, f32),
                                        _ptr(d2, f32), _ptr(score, f32), _stream()), 'ctr_din_attn_mlp_fwd')


@_timed
def din_attn_mlp_bwd(q, k, W1, W2, w3, d1, d2, dscore, g1, g2, dq, dk, *, B, T, H, acc_k=False, sl=None):
    if q.numel() != B * H or k.numel() != B * T * H or dscore.numel() != B * T or d1.numel() != B * T * 80 \
            or d2.numel() != B * T * 40 or g1.numel() != B * T * 80 or g2.numel() != B * T * 40 or dq.numel() != B * H \
            or dk.numel() != B * T * H or tuple(W1.shape) != (4 * H, 80) or tuple(W2.shape) != (80, 40) or w3.numel() != 40:
        raise ValueError('din_attn_mlp_bwd size mismatch')
    for t_ in (q, k, W1, W2, d1, d2, g1, g2, dq, dk):
        if not t_.is_contiguous():
            raise ValueError('din_attn_mlp_bwd: contiguous buffers required')
    check(lib().ctr_din_attn_mlp_bwd_sl(_ptr(q, f32), _ptr(k, f32), _sl_ptr(sl, B), B, T, H, _ptr(W1, f32), _ptr(W2, f32),
                                        _ptr(w3, f32), _ptr(d1, f32), _ptr(d2, f32), _ptr(dscore, f32), _ptr(g1, f32),
                                        _ptr(g2, f32), _ptr(dq, f32), _ptr(dk, f32), int(acc_k), _stream()),
          'ctr_din_attn_mlp_bwd')


@_timed
def din_attn_mlp_wgrad(q, k, d1, d2, dscore, g1, g2, dW1, db1, dW2, db2, dw3, db3, *, B, T, H, ws, sl=None):
    if q.numel() != B * H or k.numel() != B * T * H or dscore.numel() != B * T or d1.numel() != B * T * 80 \
            or d2.numel() != B * T * 40 or g1.numel() != B * T * 80 or g2.numel() != B * T * 40 \
            or tuple(dW1.shape) != (4 * H, 80) or tuple(dW2.shape) != (80, 40) or dw3.numel() != 40 or db1.numel() != 80 \
            or db2.numel() != 40 or db3.numel() != 1:
        raise ValueError('din_attn_mlp_wgrad size mismatch')
    for t_ in (q, k, d1, d2, g1, g2, dW1, dW2, dw3):
        if not t_.is_contiguous():
            raise ValueError('din_attn_mlp_wgrad: contiguous buffers required')
    wp, wn = ws.get(lib().ctr_din_attn_mlp_wgrad_workspace_bytes(B, H))
    check(lib().ctr_din_attn_mlp_wgrad_sl(_ptr(q, f32), _ptr(k, f32), _sl_ptr(sl, B), B, T, H, _ptr(d1, f32), _ptr(d2, f32),
                                          _ptr(dscore, f32), _ptr(g1, f32), _ptr(g2, f32), _ptr(dW1, f32),
                                          _ptr(db1, f32), _ptr(dW2, f32), _ptr(db2, f32), _ptr(dw3, f32), _ptr(db3, f32),
                                          wp, wn, _stream()), 'ctr_din_attn_mlp_wgrad')


@_timed
def din_attn_pool_fwd(score, sl, K, att, out, *, B, T, H):
    if score.numel() != B * T or sl.numel() != B or sl.dtype != torch.int32 or K.numel() != B * T * H \
            or att.numel() != B * T or out.numel() != B * H:
        raise ValueError('din_attn_pool_fwd size mismatch')
    check(lib().ctr_din_attn_pool_fwd(_ptr(score, f32), _ptr(sl), _ptr(K, f32), B, T, H, _ptr(att, f32),
                                      _ptr(out, f32), _stream()), 'ctr_din_attn_pool_fwd')


@_timed
def din_attn_pool_bwd(dout, att, sl, K, dscore, dK, *, B, T, H, acc_k=False):
    if dout.numel() != B * H or att.numel() != B * T or dscore.numel() != B * T or dK.numel() != B * T * H:
        raise ValueError('din_attn_pool_bwd size mismatch')
    check(lib().ctr_din_attn_pool_bwd(_ptr(dout, f32), _ptr(att, f32), _ptr(sl), _ptr(K, f32), B, T, H,
                                      _ptr(dscore, f32), _ptr(dK, f32), int(acc_k), _stream()), 'ctr_din_attn_pool_bwd')


@_timed
def bn_infer(x, y, *, gamma=None, beta=None, mean=None, var=None, eps=1e-3, xhat=None):
    rows, cols = x.shape
    xp, xld = _rows(x, 'x')
    yp, yld = _rows(y, 'y')
    hp, hld = (None, 0) if xhat is None else _rows(xhat, 'xhat')
    for v in (gamma, beta, mean, var):
        if v is not None and v.numel() != cols:
            raise ValueError('bn_infer: per-column vectors must have %d entries' % cols)
    check(lib().ctr_bn_infer(xp, xld, rows, cols, _ptr(gamma, f32), _ptr(beta, f32), _ptr(mean, f32), _ptr(var, f32),
                             float(eps), yp, yld, hp, hld, _stream()), 'ctr_bn_infer')


@_timed
def dice_fwd(x, alpha, y, eps=1e-3):
    rows, cols = x.shape
    xp, xld = _rows(x, 'x')
    yp, yld = _rows(y, 'y')
    if alpha.numel() != cols:
        raise ValueError('dice alpha size mismatch')
    check(lib().ctr_dice_fwd(xp, xld, rows, cols, _ptr(alpha, f32), float(eps), yp, yld, _stream()), 'ctr_dice_fwd')


@_timed
def dice_bwd(x, alpha, dy, dx, dalpha_src=None, eps=1e-3):
    rows, cols = x.shape
    xp, xld = _rows(x, 'x')
    gp, gld = _rows(dy, 'dy')
    dp, dld = _rows(dx, 'dx')
    ap, ald = (None, 0) if dalpha_src is None else _rows(dalpha_src, 'dalpha_src')
    check(lib().ctr_dice_bwd(xp, xld, rows, cols, _ptr(alpha, f32), float(eps), gp, gld, dp, dld, ap, ald, _stream()),
          'ctr_dice_bwd')


@_timed
def bn_infer_bwd(dy, xhat, dx, *, gamma, var=None, eps=1e-3, dgamma, dbeta=None):
    """Backward of bn_infer in one launch: dx = gamma * dy * rsqrt(var + eps), dgamma = sum_r dy * xhat,
    dbeta = sum_r dy.  For batch-sized row counts (a block per 16 columns walks all rows)."""
    rows, cols = dy.shape
    gp, gld = _rows(dy, 'dy')
    hp, hld = _rows(xhat, 'xhat')
    dp, dld = _rows(dx, 'dx')
    for v in (gamma, var, dgamma, dbeta):
        if v is not None and v.numel() != cols:
            raise ValueError('bn_infer_bwd: per-column vectors must have %d entries' % cols)
    if tuple(xhat.shape) != (rows, cols) or tuple(dx.shape) != (rows, cols):
        raise ValueError('bn_infer_bwd shape mismatch')
    check(lib().ctr_bn_infer_bwd(gp, gld, hp, hld, rows, cols, _ptr(gamma, f32), _ptr(var, f32), float(eps), dp, dld,
                                 _ptr(dgamma, f32), _ptr(dbeta, f32), _stream()), 'ctr_bn_infer_bwd')


@_timed
def dice_bwd_sum(x, alpha, dy, dx, dalpha, eps=1e-3):
    """dice_bwd and the column sum that is the gradient of alpha, in one launch (batch-sized row counts)."""
    rows, cols = x.shape
    xp, xld = _rows(x, 'x')
    gp, gld = _rows(dy, 'dy')
    dp, dld = _rows(dx, 'dx')
    if alpha.numel() != cols or dalpha.numel() != cols or tuple(dy.shape) != (rows, cols) or tuple(dx.shape) != (rows, cols):
        raise ValueError('dice_bwd_sum shape mismatch')
    check(lib().ctr_dice_bwd_sum(xp, xld, rows, cols, _ptr(alpha, f32), float(eps), gp, gld, dp, dld, _ptr(dalpha, f32),
                                 _stream()), 'ctr_dice_bwd_sum')


@_timed
def dsigmoid_apply(g, y):
    rows, cols = g.shape
    gp, gld = _rows(g, 'g')
    yp, yld = _rows(y, 'y')
    if y.shape[0] != rows or y.shape[1] < cols:
        raise ValueError('dsigmoid_apply shape mismatch')
    check(lib().ctr_dsigmoid_apply(gp, gld, yp, yld, rows, cols, _stream()), 'ctr_dsigmoid_apply')


@_timed
def mean_positive_diff(a, b, out):
    n = a.numel()
    if b.numel() != n:
        raise ValueError('mean_positive_diff size mismatch')
    check(lib().ctr_mean_positive_diff(_ptr(a, f32), _ptr(b, f32), n, _ptr(out, f32), _stream()),
          'ctr_mean_positive_diff')

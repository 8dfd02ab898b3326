"""Multi-GPU layout of the hot path: one process per GPU, embedding tables sharded table-wise,
dense towers data-parallel.

The reference is single-process (no collective anywhere), so this is new design; only the
single-device numerics are inherited.  Layout:

  * field f belongs to exactly one rank; ranks own CONTIGUOUS field ranges (balanced by field
    count), so a rank's rows land in a contiguous column block of every sample's Z;
  * forward:  all-to-all of ids (each rank sends, per owner, its batch's ids of that owner's
    fields) -> the owner gathers rows for all W*B_loc samples with the ordinary gather kernel
    -> all-to-all of rows back -> strided copies assemble Z / the deep input;
  * backward: all-to-all of dZ column blocks to the owners -> the ordinary sort + segment-reduce
    + fused optimizer on the owner's shard;
  * dense gradients: ONE all-reduce over the flat gradient buffer; the loss gradient is
    pre-scaled by 1/W so the sum over ranks is the global-batch mean.

Every split size is static (B_loc * fields-per-owner): no count exchange, no host sync.
Collectives go through torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box).

RowShardedFront is the alternative for tables too large or too skewed for one owner: row r of
field f lives on rank r mod W (Zipf-hot rows spread over the ranks).  Its message sizes depend on
the batch, so each step exchanges W counts and reads them on the host before the payload
all-to-all-v.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops as _ops
from .engine import EmbeddingTables
from ._lib import OPT_ADAM, OPT_GRAD_ONLY


def field_ranges(F, W):
    """W contiguous [a, b) field ranges covering [0, F), sizes differing by at most one."""
    base, rem = divmod(F, W)
    out, a = [], 0
    for j in range(W):
        n = base + (1 if j < rem else 0)
        out.append((a, a + n))
        a += n
    return out


class TorchComm:
    """Collectives straight on device tensors (RCCL when the group's backend is nccl)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def all_to_all(self, out, inp, out_splits, in_splits):
        self.dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits,
                                    group=self.group)

    def all_reduce_sum(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)


class HostStagedComm:
    """Same collectives staged through host memory over a gloo group.  For rehearsing the multi-rank
    path when several ranks share one GPU (RCCL cannot place two ranks on one device); never the
    transport of a real multi-GPU run."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def all_to_all(self, out, inp, out_splits, in_splits):
        o = torch.empty(out.shape, dtype=out.dtype)
        i = inp.detach().cpu()
        outs, ins = list(o.split(out_splits)), list(i.split(in_splits))
        outs[self.rank].copy_(ins[self.rank])
        reqs = []
        for j in range(self.world):
            if j != self.rank:
                if ins[j].numel():
                    reqs.append(self.dist.isend(ins[j].contiguous(), j, group=self.group))
                if outs[j].numel():
                    reqs.append(self.dist.irecv(outs[j], j, group=self.group))
        for r in reqs:
            r.wait()
        out.copy_(o)

    def all_reduce_sum(self, t):
        c = t.detach().cpu()
        self.dist.all_reduce(c, group=self.group)
        t.copy_(c)


class ShardedFront:
    """Drop-in for engine.EmbedFront (S semantics) when the tables are sharded over ranks."""

    def __init__(self, cate_list, E, cont_num, cont_embed, comm, device, backend=_ops):
        self.comm, self.ops = comm, backend
        self.W, self.rank = comm.world, comm.rank
        self.F = len(cate_list)
        self.E = int(E)
        self.C = int(cont_num)
        self.cont_embed = bool(cont_embed)
        self.mode = 'S'
        self.Ncat = self.F
        self.N = self.F + (self.C if cont_embed else 0)
        self.ranges = field_ranges(self.F, self.W)
        self.a, self.b = self.ranges[self.rank]
        self.nf_own = self.b - self.a
        own = list(cate_list[self.a:self.b]) or [1]          # a rank without fields keeps a 1-row dummy
        self.t = EmbeddingTables(own, E, device)
        self.device = device
        self._buf = {}

    def owner_of(self, f):
        for j, (a, b) in enumerate(self.ranges):
            if a <= f < b:
                return j
        raise IndexError(f)

    def _buffers(self, B):
        if B not in self._buf:
            E, W = self.E, self.W
            nf = [b - a for a, b in self.ranges]
            dev = self.device
            self._buf[B] = dict(
                recv_idx=torch.zeros(W * B * self.nf_own, dtype=torch.int32, device=dev),
                rows=torch.zeros(W * B, max(self.nf_own, 1) * E, dtype=torch.float32, device=dev),
                blocks=torch.zeros(B * self.F * E, dtype=torch.float32, device=dev),
                grows=torch.zeros(W * B, max(self.nf_own, 1) * E, dtype=torch.float32, device=dev),
                keys=torch.zeros(max(W * B * self.nf_own, 1), dtype=torch.int32, device=dev),
                src=torch.zeros(max(W * B * self.nf_own, 1), dtype=torch.int32, device=dev),
                idx_in=[B * n for n in nf], idx_out=[B * self.nf_own] * W,
                row_in=[B * self.nf_own * E] * W, row_out=[B * n * E for n in nf])
            # where row n of sample b sits in `blocks` (one [B, nf_j*E] block per owner j, owner after owner)
            fbase, fstride, off = [], [], 0
            for a, b in self.ranges:
                for n in range(a, b):
                    fbase.append(off + (n - a) * E)
                    fstride.append((b - a) * E)
                off += B * (b - a) * E
            self._buf[B]['fbase'] = torch.tensor(fbase, dtype=torch.int64, device=dev)
            self._buf[B]['fstride'] = torch.tensor(fstride, dtype=torch.int64, device=dev)
        return self._buf[B]

    def _block_views(self, flat, B):
        """Per-owner [B, nf_j*E] views of a flat buffer laid out owner after owner."""
        views, off = [], 0
        for a, b in self.ranges:
            n = B * (b - a) * self.E
            views.append(flat[off:off + n].view(B, (b - a) * self.E) if b > a else None)
            off += n
        return views

    def forward(self, idx, x_cont, *, z_out, cont_tail=False, cont_w=None, first_out=None, s_out=None,
                second_out=None, drop_first=None, drop_second=None, oor_count=None):
        if z_out is None:
            raise ValueError('the sharded front end assembles Z in z_out; it cannot be None')
        ops, E, W = self.ops, self.E, self.W
        B = idx.shape[0]
        bf = self._buffers(B)
        send_idx = torch.cat([idx[:, a:b].reshape(-1) for a, b in self.ranges])
        self.comm.all_to_all(bf['recv_idx'], send_idx, bf['idx_out'], bf['idx_in'])
        if self.nf_own:
            ops.embed_fwd(self.t.weight, self.t.row_offset, bf['recv_idx'].view(W * B, self.nf_own), E=E,
                          z_out=bf['rows'], oor_count=oor_count)
        self.comm.all_to_all(bf['blocks'], bf['rows'].view(-1)[:W * B * self.nf_own * E], bf['row_out'],
                             bf['row_in'])
        # one pass: owner blocks -> Z rows, continuous rows / tail, FM sums, dropout
        ops.embed_assemble_fwd(bf['blocks'], bf['fbase'], bf['fstride'], x_cont if (self.cont_embed or cont_tail) else None,
                               cont_w if self.cont_embed else None, F=self.F, E=E, z_out=z_out, cont_tail=cont_tail,
                               first_out=first_out, s_out=s_out, second_out=second_out, drop_first=drop_first,
                               drop_second=drop_second)

    def dz_route(self, B):
        """Keyword arguments for ops.embed_bwd_dz: dZ of the sharded fields goes straight into the buffer the
        gradient exchange sends (then call backward_tables(..., dz_in_blocks=True))."""
        bf = self._buffers(B)
        return dict(blocks_out=bf['blocks'], fbase=bf['fbase'], fstride=bf['fstride'], F=self.F)

    def backward_tables(self, idx, dz, keys, src, ws, opt, lazy, dz_in_blocks=False):
        ops, E, W = self.ops, self.E, self.W
        B = idx.shape[0]
        bf = self._buffers(B)
        if not dz_in_blocks:
            for (a, b), blk in zip(self.ranges, self._block_views(bf['blocks'], B)):
                if blk is not None:
                    ops.copy2d(dz[:, a * E:b * E], blk)
        self.comm.all_to_all(bf['grows'].view(-1)[:W * B * self.nf_own * E], bf['blocks'], bf['row_in'],
                             bf['row_out'])
        if not self.nf_own:
            return
        t, n_items = self.t, W * B * self.nf_own
        ops.emb_sort(bf['recv_idx'].view(W * B, self.nf_own), t.row_offset, N=self.nf_own,
                     total_rows=t.total_rows, rmode=False, keys_out=bf['keys'], src_out=bf['src'], ws=ws)
        if opt.kind == OPT_ADAM and not lazy:
            t.grad.zero_()
            ops.emb_sparse_apply(bf['keys'], bf['src'], n_items, N=self.nf_own, E=E, total_rows=t.total_rows,
                                 dz=bf['grows'], opt=ops.make_opt(OPT_GRAD_ONLY), slot0=t.grad, ws=ws)
            ops.dense_update(t.weight, t.grad, opt, t.slot0, t.slot1)
        else:
            ops.emb_sparse_apply(bf['keys'], bf['src'], n_items, N=self.nf_own, E=E, total_rows=t.total_rows,
                                 dz=bf['grows'], opt=opt, tables=t.weight, slot0=t.slot0, slot1=t.slot1, ws=ws)

    def backward_cont_w(self, dz, x_cont, g_cont_w, ws):
        E = self.E
        self.ops.colsum(dz[:, self.Ncat * E:(self.Ncat + self.C) * E], g_cont_w.reshape(-1), w=x_cont,
                        w_group=E, ws=ws)


def row_shard_layout(cate_list, W):
    """Per rank j: the number of rows of each field it owns (rows r with r % W == j) and the first
    local row of each field inside its concatenated shard."""
    cards = [[(int(v) - j + W - 1) // W if int(v) > j else 0 for v in cate_list] for j in range(W)]
    offs = [np.concatenate([[0], np.cumsum(c)]).astype(np.int64) for c in cards]
    return cards, offs


class RowShardedFront:
    """Drop-in for engine.EmbedFront (S semantics) with every table sharded by row over the ranks.

    forward : route each lookup (dest = id mod W, key = row inside the owner's shard) -> stable sort
              by dest -> all-to-all-v of keys -> the owner gathers rows with the ordinary gather kernel
              -> all-to-all-v of rows back -> scatter into Z in lookup order.
    backward: gather dZ rows in the same send order -> all-to-all-v to the owners -> the ordinary
              sort + segment-reduce + fused optimizer on the owner's shard."""

    def __init__(self, cate_list, E, cont_num, cont_embed, comm, device, backend=_ops):
        self.comm, self.ops = comm, backend
        self.W, self.rank = comm.world, comm.rank
        self.cate_list = [int(v) for v in cate_list]
        self.F = len(self.cate_list)
        self.E = int(E)
        self.C = int(cont_num)
        self.cont_embed = bool(cont_embed)
        self.mode = 'S'
        self.Ncat = self.F
        self.N = self.F + (self.C if cont_embed else 0)
        self.cards_loc, self.offs_loc = row_shard_layout(self.cate_list, self.W)
        # one spare row keeps the shard non-empty on ranks that own nothing
        self.t = EmbeddingTables(list(self.cards_loc[self.rank]) + [1], E, device)
        self.total_local = int(self.offs_loc[self.rank][-1])
        full = np.concatenate([[0], np.cumsum(self.cate_list)]).astype(np.int64)
        self.full_offset = torch.as_tensor(full, device=device)
        self.local_offset = torch.as_tensor(np.stack([o[:-1] for o in self.offs_loc]).astype(np.int32), device=device)
        self.shard_offset = torch.as_tensor(np.array([0, max(self.total_local, 1)], dtype=np.int64), device=device)
        self.device = device
        self.key_bits = max(1, int(self.W - 1).bit_length())
        self._buf = {}
        self._cap = 0
        self._last = None

    def local_rows(self, f):
        """Global row ids of field f held by this rank, in local order."""
        return np.arange(self.rank, self.cate_list[f], self.W)

    def _buffers(self, B):
        if B not in self._buf:
            n, E, dev = B * self.F, self.E, self.device
            i32 = lambda k: torch.zeros(max(k, 1), dtype=torch.int32, device=dev)
            self._buf[B] = dict(dest=i32(n), lkey=i32(n), send_keys=i32(n),
                                arange=torch.arange(max(n, 1), dtype=torch.int32, device=dev),
                                back=torch.zeros(max(n, 1), E, dtype=torch.float32, device=dev))
        return self._buf[B]

    def _grow(self, n_recv):
        if n_recv > self._cap:
            cap = max(int(n_recv * 1.25) + 64, 1024)
            dev = self.device
            self._recv_keys = torch.zeros(cap, dtype=torch.int32, device=dev)
            self._rows = torch.zeros(cap, self.E, dtype=torch.float32, device=dev)
            self._skeys = torch.zeros(cap, dtype=torch.int32, device=dev)
            self._ssrc = torch.zeros(cap, dtype=torch.int32, device=dev)
            self._cap = cap

    def forward(self, idx, x_cont, *, z_out, cont_tail=False, cont_w=None, first_out=None, s_out=None,
                second_out=None, drop_first=None, drop_second=None, oor_count=None, ws=None):
        if z_out is None:
            raise ValueError('the sharded front end assembles Z in z_out; it cannot be None')
        ops, E, W, F = self.ops, self.E, self.W, self.F
        B = idx.shape[0]
        n = B * F
        bf = self._buffers(B)
        ws = ws if ws is not None else self._ws()
        ops.shard_route(idx, self.full_offset, self.local_offset, W, bf['dest'], bf['lkey'])
        sdest, perm = ops.sort_pairs(bf['dest'][:n], bf['arange'][:n], self.key_bits, ws)
        ops.rows_gather(bf['lkey'][:n].view(B, F), perm, n, bf['send_keys'], F=F, E=1)
        send_counts = torch.bincount(sdest, minlength=W).to(torch.int64)
        recv_counts = torch.zeros_like(send_counts)
        self.comm.all_to_all(recv_counts, send_counts, [1] * W, [1] * W)
        sc = [int(v) for v in send_counts.tolist()]           # host reads the W counts: one sync per step
        rc = [int(v) for v in recv_counts.tolist()]
        n_recv = sum(rc)
        self._grow(n_recv)
        self.comm.all_to_all(self._recv_keys[:n_recv], bf['send_keys'][:n], rc, sc)
        if n_recv:
            ops.embed_fwd(self.t.weight, self.shard_offset, self._recv_keys[:n_recv].view(n_recv, 1), E=E,
                          z_out=self._rows[:n_recv], oor_count=oor_count)
        self.comm.all_to_all(bf['back'].view(-1)[:n * E], self._rows.view(-1)[:n_recv * E],
                             [c * E for c in sc], [c * E for c in rc])
        ops.rows_scatter(bf['back'], perm, n, z_out, F=F, E=E)
        self._last = (perm, sc, rc, n_recv)
        if self.cont_embed:
            ops.embed_fwd(self.t.weight, self.t.row_offset[:1], idx[:, :0].contiguous(), x_cont, cont_w, E=E,
                          z_out=z_out[:, F * E:], cont_tail=cont_tail)
        elif cont_tail:
            ops.copy2d(x_cont, z_out[:, F * E:F * E + self.C])
        if first_out is not None or s_out is not None or second_out is not None:
            ops.fm_fwd(z_out, N=self.N, E=E, first_out=first_out, s_out=s_out, second_out=second_out,
                       drop_first=drop_first, drop_second=drop_second)

    def _ws(self):
        if not hasattr(self, '_own_ws'):
            self._own_ws = self.ops.Workspace(self.device)
        return self._own_ws

    def backward_tables(self, idx, dz, keys, src, ws, opt, lazy):
        ops, E, F = self.ops, self.E, self.F
        B = idx.shape[0]
        n = B * F
        bf = self._buffers(B)
        perm, sc, rc, n_recv = self._last
        ops.rows_gather(dz, perm, n, bf['back'], F=F, E=E)
        self.comm.all_to_all(self._rows.view(-1)[:n_recv * E], bf['back'].view(-1)[:n * E],
                             [c * E for c in rc], [c * E for c in sc])
        if not n_recv:
            return
        t, total = self.t, max(self.total_local, 1)
        ops.emb_sort(self._recv_keys[:n_recv].view(n_recv, 1), self.shard_offset, N=1, total_rows=total, rmode=False,
                     keys_out=self._skeys, src_out=self._ssrc, ws=ws)
        if opt.kind == OPT_ADAM and not lazy:
            t.grad.zero_()
            ops.emb_sparse_apply(self._skeys, self._ssrc, n_recv, N=1, E=E, total_rows=total, dz=self._rows[:n_recv],
                                 opt=ops.make_opt(OPT_GRAD_ONLY), slot0=t.grad, ws=ws)
            ops.dense_update(t.weight, t.grad, opt, t.slot0, t.slot1)
        else:
            ops.emb_sparse_apply(self._skeys, self._ssrc, n_recv, N=1, E=E, total_rows=total, dz=self._rows[:n_recv],
                                 opt=opt, tables=t.weight, slot0=t.slot0, slot1=t.slot1, ws=ws)

    def backward_cont_w(self, dz, x_cont, g_cont_w, ws):
        E = self.E
        self.ops.colsum(dz[:, self.Ncat * E:(self.Ncat + self.C) * E], g_cont_w.reshape(-1), w=x_cont,
                        w_group=E, ws=ws)

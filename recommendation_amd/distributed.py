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
"""
from __future__ import annotations

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
        for (a, b), blk in zip(self.ranges, self._block_views(bf['blocks'], B)):
            if blk is not None:
                ops.copy2d(blk, z_out[:, a * E:b * E])
        if self.cont_embed:
            ops.embed_fwd(self.t.weight, self.t.row_offset[:1], idx[:, :0].contiguous(), x_cont, cont_w, E=E,
                          z_out=z_out[:, self.F * E:], cont_tail=cont_tail)
        elif cont_tail:
            ops.copy2d(x_cont, z_out[:, self.F * E:self.F * E + self.C])
        if first_out is not None or s_out is not None or second_out is not None:
            ops.fm_fwd(z_out, N=self.N, E=E, first_out=first_out, s_out=s_out, second_out=second_out,
                       drop_first=drop_first, drop_second=drop_second)

    def backward_tables(self, idx, dz, keys, src, ws, opt, lazy):
        ops, E, W = self.ops, self.E, self.W
        B = idx.shape[0]
        bf = self._buffers(B)
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

"""Multi-GPU layout of the hot path: one process per GPU, embedding tables sharded table-wise,
dense towers data-parallel.

The reference is single-process (no collective anywhere), so this is new design; only the
single-device numerics are inherited.  Layout:

  * field f belongs to exactly one rank; fields are dealt to ranks by table ROWS (largest table first, to the
    least loaded rank: pack_fields), so the big tables of a Criteo-shape model end up on different ranks;
  * forward:  all-to-all of ids (each rank sends, per owner, its batch's ids of that owner's
    fields) -> the owner gathers rows for all W*B_loc samples with the ordinary gather kernel
    -> all-to-all of rows back -> strided copies assemble Z / the deep input;
  * backward: all-to-all of dZ column blocks to the owners -> the ordinary sort + segment-reduce
    + fused optimizer on the owner's shard;
  * dense gradients: ONE all-reduce over the flat gradient buffer; the loss gradient is
    pre-scaled by 1/W so the sum over ranks is the global-batch mean.

Every split size is static (B_loc * fields-per-owner): no count exchange, no host sync.
Collectives go through torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box).

RowShardedFront is the layout for tables too large or too skewed for one owner: row r of field f
lives on rank (r + f) mod W (Zipf-hot rows spread over the ranks; the rotation by f keeps the hottest
row of every field off one rank).  Its traffic depends on the batch, so every rank sends each peer a
FIXED-capacity bucket (B*F/W lookups times a slack factor, unfilled slots are zero rows): all splits are
static here too - no count exchange and no host synchronisation; a bucket that overflows is counted on the
device and reported at the step's end.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops as _ops
from .engine import EmbeddingTables
from ._lib import OPT_ADAM, OPT_GRAD_ONLY


def pack_fields(cate_list, W):
    """Fields dealt to W ranks by table rows: largest table first, each to the rank with the fewest rows so far
    (ties: lowest rank).  Returns W ascending field lists.  With the Criteo-1TB cardinalities and W = 8 the four
    tables above 100 M rows go to four different ranks (balancing by field COUNT put three of them on one)."""
    load = [0] * W
    out = [[] for _ in range(W)]
    for f in sorted(range(len(cate_list)), key=lambda i: (-int(cate_list[i]), i)):
        j = min(range(W), key=lambda r: (load[r], r))
        out[j].append(f)
        load[j] += int(cate_list[f])
    return [sorted(fs) for fs in out]


class TorchComm:
    """Collectives straight on device tensors (RCCL when the group's backend is nccl)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def all_to_all(self, out, inp, out_splits=None, in_splits=None):
        """splits None: W equal parts (static sizes; nothing for the host to read)."""
        self.dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits,
                                    group=self.group)

    def all_reduce_sum(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)


class CtrComm:
    """The same collectives through the library's own entry points (ctr_comm_init / ctr_all2all(v) /
    ctr_allreduce_sum_f32 over RCCL): what a host that binds only the C header uses.  The 128-byte id that rank 0
    creates travels to the others through `exchange` (a callable: bytes-or-None on rank 0 -> bytes on every rank;
    default: a broadcast over an initialised torch.distributed group, which then carries nothing else)."""

    def __init__(self, world=None, rank=None, exchange=None):
        import ctypes as C
        from . import _lib
        self._C, self.lib = C, _lib.load()
        if world is None or rank is None:
            import torch.distributed as dist
            world, rank = dist.get_world_size(), dist.get_rank()
        self.world, self.rank = int(world), int(rank)
        ident = C.create_string_buffer(_lib.COMM_ID_BYTES)
        if self.rank == 0:
            self._check(self.lib.ctr_comm_unique_id(ident))
        if self.world > 1 or exchange is not None:
            raw = (exchange or self._broadcast)(ident.raw if self.rank == 0 else None)
            ident = C.create_string_buffer(bytes(raw), _lib.COMM_ID_BYTES)
        self.handle = C.c_void_p()
        self._check(self.lib.ctr_comm_init(ident, self.world, self.rank, C.byref(self.handle)))

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(self.lib.ctr_last_error().decode())

    @staticmethod
    def _broadcast(raw):
        import torch.distributed as dist
        box = [raw]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def _stream(self):
        return self._C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def all_to_all(self, out, inp, out_splits=None, in_splits=None):
        """Rows of `inp` to the ranks, rows of `out` from them (split sizes in rows; None: W equal parts)."""
        C = self._C
        if not (out.is_contiguous() and inp.is_contiguous()):
            raise ValueError('all_to_all needs contiguous tensors')
        row_o = out.element_size() * (out.numel() // max(out.shape[0], 1))
        row_i = inp.element_size() * (inp.numel() // max(inp.shape[0], 1))
        if out_splits is None and in_splits is None and out.shape[0] % self.world == 0 and inp.shape[0] % self.world == 0 \
                and out.shape[0] * row_o == inp.shape[0] * row_i:
            self._check(self.lib.ctr_all2all(self.handle, C.c_void_p(inp.data_ptr()), C.c_void_p(out.data_ptr()),
                                             inp.shape[0] // self.world * row_i, self._stream()))
            return
        out_splits = [out.shape[0] // self.world] * self.world if out_splits is None else list(out_splits)
        in_splits = [inp.shape[0] // self.world] * self.world if in_splits is None else list(in_splits)
        arr = lambda v: (C.c_int64 * self.world)(*[int(x) for x in v])
        offs = lambda v: [sum(v[:k]) for k in range(self.world)]
        self._check(self.lib.ctr_all2allv(self.handle, C.c_void_p(inp.data_ptr()), arr([o * row_i for o in offs(in_splits)]),
                                          arr([n * row_i for n in in_splits]), C.c_void_p(out.data_ptr()),
                                          arr([o * row_o for o in offs(out_splits)]), arr([n * row_o for n in out_splits]),
                                          self._stream()))

    def all_reduce_sum(self, t):
        if t.dtype == torch.int32:                 # the small event counters (overflowed buckets, bad ids): exact in float32
            f = t.to(torch.float32)
            self.all_reduce_sum(f)
            t.copy_(f)
            return
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise TypeError('all_reduce_sum takes a contiguous float32 (or int32 counter) tensor')
        self._check(self.lib.ctr_allreduce_sum_f32(self.handle, self._C.c_void_p(t.data_ptr()), t.numel(), self._stream()))

    def close(self):
        """Waits for the device, then destroys the communicator (every rank, at the same point)."""
        if getattr(self, 'handle', None):
            torch.cuda.synchronize()
            self.lib.ctr_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:            # interpreter shutdown: the runtime may be gone already
            pass


class DirectAllReduce:
    """Sum of a flat float32 buffer over the ranks as a reduce-scatter and an all-gather, each ONE all-to-all of equal
    pieces (every pair of GPUs exchanges 1/W of the buffer directly, twice) instead of the collective library's ring:
    for the towers' few megabytes of gradients on point-to-point links, 2 (W-1)/W of the buffer per rank in two
    latency-bound steps rather than 2 (W-1) ring steps.  The W pieces of a chunk are added in rank order by
    ops.colsum, so the sum does not depend on the topology.  Works over any of the transports above."""

    def __init__(self, comm, n, device):
        self.comm, self.n, W = comm, int(n), comm.world
        self.chunk = (-(-self.n // W) + 3) // 4 * 4
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=device)
        self.send, self.recv, self.red = z(W, self.chunk), z(W, self.chunk), z(self.chunk)

    def __call__(self, t, ws):
        if t.numel() != self.n or t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError('DirectAllReduce was sized for a contiguous float32 buffer of %d elements' % self.n)
        W = self.comm.world
        self.send.view(-1)[:self.n].copy_(t.view(-1))              # the padding behind it stays zero
        self.comm.all_to_all(self.recv, self.send)                 # row r: chunk `rank` as rank r holds it
        self.ops_colsum(self.recv, self.red, ws)
        self.send.copy_(self.red.unsqueeze(0).expand(W, self.chunk))
        self.comm.all_to_all(self.recv, self.send)                 # row r: the finished chunk r
        t.view(-1).copy_(self.recv.view(-1)[:self.n])

    @staticmethod
    def ops_colsum(x, out, ws):
        _ops.colsum(x, out, ws=ws)


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

    def all_to_all(self, out, inp, out_splits=None, in_splits=None):
        o = torch.empty(out.shape, dtype=out.dtype)
        i = inp.detach().cpu()
        if out_splits is None:
            out_splits = [out.shape[0] // self.world] * self.world
        if in_splits is None:
            in_splits = [inp.shape[0] // self.world] * self.world
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


class _SortBesideExchange:
    """The owner's pair sort needs only the ids that arrived in the forward pass, and the gradient exchange leaves the
    compute units idle: on a GPU the sort runs on a stream of its own while the exchange is under way (the two are
    joined before the row update).  With any other backend (the CPU stand-ins of the plan tests) the exchange comes
    first, then the sort, on the one stream there is."""
    _aux = None
    _aux_ws = None

    def _sort_beside(self, sort_fn, exchange_fn, ws):
        if self.ops is not _ops or torch.device(self.device).type != 'cuda':
            exchange_fn()
            sort_fn(ws)
            return
        cur = torch.cuda.current_stream()
        if self._aux is None:
            self._aux = torch.cuda.Stream(self.device)
            self._aux_ws = _ops.Workspace(self.device)
        self._aux.wait_stream(cur)                   # after the forward's id exchange and the last update's reads
        with torch.cuda.stream(self._aux):
            sort_fn(self._aux_ws)
        exchange_fn()
        cur.wait_stream(self._aux)


class ShardedFront(_SortBesideExchange):
    """Drop-in for engine.EmbedFront (S semantics) when whole tables are dealt to ranks (pack_fields)."""

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
        self.fields_of = pack_fields(cate_list, self.W)       # owner rank -> ascending field list
        self.own_fields = self.fields_of[self.rank]
        self.nf_own = len(self.own_fields)
        own = [int(cate_list[f]) for f in self.own_fields] or [1]   # a rank without fields keeps a 1-row dummy
        self.t = EmbeddingTables(own, E, device)
        self.device = device
        self.oor_count = None     # optional device int32[1]: out-of-range ids are counted here, on the OWNER's rank
        self._buf = {}

    def owner_of(self, f):
        for j, fs in enumerate(self.fields_of):
            if f in fs:
                return j
        raise IndexError(f)

    def _buffers(self, B):
        if B not in self._buf:
            E, W, F = self.E, self.W, self.F
            nf = [len(fs) for fs in self.fields_of]
            dev = self.device
            bf = dict(
                send_idx=torch.zeros(max(B * F, 1), dtype=torch.int32, device=dev),
                recv_idx=torch.zeros(max(W * B * self.nf_own, 1), dtype=torch.int32, device=dev),
                rows=torch.zeros(W * B, max(self.nf_own, 1) * E, dtype=torch.float32, device=dev),
                blocks=torch.zeros(max(B * F * E, 1), dtype=torch.float32, device=dev),
                grows=torch.zeros(W * B, max(self.nf_own, 1) * E, dtype=torch.float32, device=dev),
                keys=torch.zeros(max(W * B * self.nf_own, 1), dtype=torch.int32, device=dev),
                src=torch.zeros(max(W * B * self.nf_own, 1), dtype=torch.int32, device=dev),
                idx_in=[B * n for n in nf], idx_out=[B * self.nf_own] * W,
                row_in=[B * self.nf_own * E] * W, row_out=[B * n * E for n in nf])
            # send order: owner after owner, inside an owner sample after sample, inside a sample the owner's fields;
            # send_perm[s] = b*F + f of the lookup in send slot s (one array serves ids and E-wide rows alike).
            # fbase / fstride: where row f of sample b sits in `blocks` (one [B, nf_j*E] block per owner j)
            perm = np.empty(B * F, dtype=np.int64)
            fbase, fstride = [0] * F, [0] * F
            pos, off = 0, 0
            b_col = np.arange(B, dtype=np.int64)[:, None] * F
            for fs in self.fields_of:
                k = len(fs)
                if k:
                    perm[pos:pos + B * k] = (b_col + np.asarray(fs, dtype=np.int64)[None, :]).reshape(-1)
                    for q, f in enumerate(fs):
                        fbase[f] = off + q * E
                        fstride[f] = k * E
                pos += B * k
                off += B * k * E
            bf['send_perm'] = torch.as_tensor(perm.astype(np.int32), device=dev)
            bf['fbase'] = torch.tensor(fbase, dtype=torch.int64, device=dev)
            bf['fstride'] = torch.tensor(fstride, dtype=torch.int64, device=dev)
            self._buf[B] = bf
        return self._buf[B]

    def forward(self, idx, x_cont, *, z_out, cont_tail=False, cont_w=None, first_out=None, s_out=None,
                second_out=None, drop_first=None, drop_second=None, oor_count=None):
        if z_out is None:
            raise ValueError('the sharded front end assembles Z in z_out; it cannot be None')
        ops, E, W, F = self.ops, self.E, self.W, self.F
        B = idx.shape[0]
        bf = self._buffers(B)
        ops.rows_gather(idx, bf['send_perm'], B * F, bf['send_idx'], F=F, E=1)        # ids in send order
        self.comm.all_to_all(bf['recv_idx'][:W * B * self.nf_own], bf['send_idx'][:B * F], bf['idx_out'], bf['idx_in'])
        if self.nf_own:
            ops.embed_fwd(self.t.weight, self.t.row_offset, bf['recv_idx'][:W * B * self.nf_own].view(W * B, self.nf_own),
                          E=E, z_out=bf['rows'], oor_count=oor_count if oor_count is not None else self.oor_count)
        self.comm.all_to_all(bf['blocks'][:B * F * E], bf['rows'].view(-1)[:W * B * self.nf_own * E], bf['row_out'],
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
        ops, E, W, F = self.ops, self.E, self.W, self.F
        B = idx.shape[0]
        bf = self._buffers(B)
        if not dz_in_blocks:
            ops.rows_gather(dz, bf['send_perm'], B * F, bf['blocks'], F=F, E=E)     # dZ rows in send order
        t, n_items = self.t, W * B * self.nf_own

        def exchange():
            self.comm.all_to_all(bf['grows'].view(-1)[:W * B * self.nf_own * E], bf['blocks'][:B * F * E], bf['row_in'],
                                 bf['row_out'])

        def sort(sws):
            if self.nf_own:
                ops.emb_sort(bf['recv_idx'][:n_items].view(W * B, self.nf_own), t.row_offset, N=self.nf_own,
                             total_rows=t.total_rows, rmode=False, keys_out=bf['keys'], src_out=bf['src'], ws=sws)

        self._sort_beside(sort, exchange, ws)
        if not self.nf_own:
            return
        if opt.kind == OPT_ADAM and not lazy and t.grad is None:
            # every row moves (tf.train.AdamOptimizer) without a dense gradient buffer, as in engine.EmbedFront
            ops.emb_sparse_apply(bf['keys'], bf['src'], n_items, N=self.nf_own, E=E, total_rows=t.total_rows,
                                 dz=bf['grows'], opt=opt, tables=t.weight, slot0=t.slot0, slot1=t.slot1, ws=ws)
            ops.adam_untouched_rows(bf['keys'], n_items, total_rows=t.total_rows, E=E, opt=opt, tables=t.weight,
                                    slot0=t.slot0, slot1=t.slot1, touched=t.touched)
        elif opt.kind == OPT_ADAM and not lazy:
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


def row_owner_start(f, rank, W):
    """Smallest row r >= 0 of field f that rank `rank` owns: (r + f) mod W == rank.  Its rows are
    r = start, start + W, ... at local rows r // W = 0, 1, ..."""
    return (rank - f) % W


def row_shard_layout(cate_list, W):
    """Per rank j: the number of rows of each field it owns (rows r with (r + f) % W == j) and the first
    local row of each field inside its concatenated shard."""
    cards = []
    for j in range(W):
        row = []
        for f, v in enumerate(cate_list):
            s = row_owner_start(f, j, W)
            row.append((int(v) - s + W - 1) // W if int(v) > s else 0)
        cards.append(row)
    offs = [np.concatenate([[0], np.cumsum(c)]).astype(np.int64) for c in cards]
    return cards, offs


class RowShardedFront(_SortBesideExchange):
    """Drop-in for engine.EmbedFront (S semantics) with every table sharded by row over the ranks.

    forward : route each lookup (dest = (id + f) mod W, key = row inside the owner's shard) -> stable sort
              by dest -> fixed-capacity buckets (cap slots per peer) -> all-to-all of keys (equal splits) ->
              the owner gathers rows with the ordinary gather kernel (empty slots: zero rows) -> all-to-all
              of rows back -> scatter into Z in lookup order.
    backward: gather dZ rows into the same slots -> all-to-all to the owners -> the ordinary
              sort + segment-reduce + fused optimizer on the owner's shard.
    Nothing on this path depends on the batch's ids on the host: no count exchange, no synchronisation,
    no allocation after the first step at a batch size.  `overflow` (device int32) counts lookups that did
    not fit their bucket; check_overflow() raises when it is non-zero (raise `slack`)."""

    def __init__(self, cate_list, E, cont_num, cont_embed, comm, device, backend=_ops, slack=1.5):
        self.comm, self.ops = comm, backend
        self.W, self.rank = comm.world, comm.rank
        if self.W > 64:
            raise ValueError('row sharding supports up to 64 ranks')
        self.cate_list = [int(v) for v in cate_list]
        self.F = len(self.cate_list)
        self.E = int(E)
        self.C = int(cont_num)
        self.cont_embed = bool(cont_embed)
        self.mode = 'S'
        self.Ncat = self.F
        self.N = self.F + (self.C if cont_embed else 0)
        self.slack = float(slack)
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
        self.overflow = torch.zeros(1, dtype=torch.int32, device=device)
        self.oor_count = None     # optional device int32[1]: out-of-range ids of this rank's batch are counted here
        self._buf = {}
        self._own_ws = None

    def local_rows(self, f):
        """Global row ids of field f held by this rank, in local order."""
        return np.arange(row_owner_start(f, self.rank, self.W), self.cate_list[f], self.W)

    def capacity(self, B):
        """Slots per destination rank: the even share of B*F lookups times the slack, a multiple of 64."""
        n = B * self.F
        return min(max(n, 1), (int(np.ceil(n / self.W * self.slack)) + 63) // 64 * 64)

    def _buffers(self, B):
        if B not in self._buf:
            n, E, W, dev = B * self.F, self.E, self.W, self.device
            cap = self.capacity(B)
            i32 = lambda k: torch.zeros(max(k, 1), dtype=torch.int32, device=dev)
            f32 = lambda r: torch.zeros(max(r, 1), E, dtype=torch.float32, device=dev)
            self._buf[B] = dict(cap=cap, dest=i32(n), lkey=i32(n), arange=torch.arange(max(n, 1), dtype=torch.int32, device=dev),
                                send_keys=i32(W * cap), slot_perm=i32(W * cap), recv_keys=i32(W * cap),
                                rows=f32(W * cap), back=f32(W * cap), skeys=i32(W * cap), ssrc=i32(W * cap))
        return self._buf[B]

    def forward(self, idx, x_cont, *, z_out, cont_tail=False, cont_w=None, first_out=None, s_out=None,
                second_out=None, drop_first=None, drop_second=None, oor_count=None, ws=None):
        if z_out is None:
            raise ValueError('the sharded front end assembles Z in z_out; it cannot be None')
        ops, E, W, F = self.ops, self.E, self.W, self.F
        B = idx.shape[0]
        n = B * F
        bf = self._buffers(B)
        cap = bf['cap']
        ws = ws if ws is not None else self._ws()
        ops.shard_route(idx, self.full_offset, self.local_offset, W, bf['dest'], bf['lkey'],
                        oor_count=oor_count if oor_count is not None else self.oor_count)
        sdest, perm = ops.sort_pairs(bf['dest'][:n], bf['arange'][:n], self.key_bits, ws)
        # a lookup that does not fit its bucket is counted, and its row of Z is zeroed (nothing will be scattered there)
        ops.shard_bucket(sdest, perm, bf['lkey'], n, W, cap, bf['send_keys'], bf['slot_perm'], self.overflow,
                         z_zero=z_out, F=F, E=E)
        self.comm.all_to_all(bf['recv_keys'][:W * cap], bf['send_keys'][:W * cap])
        # empty slots carry key -1: zero rows (they are not out-of-range ids, so they are not counted)
        ops.embed_fwd(self.t.weight, self.shard_offset, bf['recv_keys'][:W * cap].view(W * cap, 1), E=E,
                      z_out=bf['rows'][:W * cap])
        self.comm.all_to_all(bf['back'].view(-1)[:W * cap * E], bf['rows'].view(-1)[:W * cap * E])
        ops.rows_scatter(bf['back'], bf['slot_perm'], W * cap, z_out, F=F, E=E)
        if self.cont_embed:
            ops.embed_fwd(self.t.weight, self.t.row_offset[:1], idx[:, :0], x_cont, cont_w, E=E,
                          z_out=z_out[:, F * E:], cont_tail=cont_tail)
        elif cont_tail:
            ops.copy2d(x_cont, z_out[:, F * E:F * E + self.C])
        if first_out is not None or s_out is not None or second_out is not None:
            ops.fm_fwd(z_out, N=self.N, E=E, first_out=first_out, s_out=s_out, second_out=second_out,
                       drop_first=drop_first, drop_second=drop_second)

    def _ws(self):
        if self._own_ws is None:
            self._own_ws = self.ops.Workspace(self.device)
        return self._own_ws

    def check_overflow(self):
        """Raises when a send bucket overflowed on ANY rank since the last check.  Collective: every rank calls it at
        the same point of its program (the counters are summed over the ranks first, so that all of them raise
        together instead of one raising while the others wait in the next exchange).  Synchronises the host."""
        total = self.overflow.clone()
        if self.W > 1:
            self.comm.all_reduce_sum(total)
        n = int(total.item())
        if n:
            self.overflow.zero_()
            raise RuntimeError('%d lookup(s) did not fit the fixed-capacity send buckets of the row-sharded tables '
                               '(slack %.2f, all ranks together); raise args.shard_slack' % (n, self.slack))

    def backward_tables(self, idx, dz, keys, src, ws, opt, lazy):
        ops, E, F, W = self.ops, self.E, self.F, self.W
        B = idx.shape[0]
        bf = self._buffers(B)
        cap = bf['cap']
        n_recv = W * cap
        ops.rows_gather(dz, bf['slot_perm'], n_recv, bf['back'], F=F, E=E)            # empty slots: zero rows
        t, total = self.t, max(self.total_local, 1)
        self._sort_beside(
            lambda sws: ops.emb_sort(bf['recv_keys'][:n_recv].view(n_recv, 1), self.shard_offset, N=1, total_rows=total,
                                     rmode=False, keys_out=bf['skeys'], src_out=bf['ssrc'], ws=sws),
            lambda: self.comm.all_to_all(bf['rows'].view(-1)[:n_recv * E], bf['back'].view(-1)[:n_recv * E]), ws)
        if opt.kind == OPT_ADAM and not lazy and t.grad is None:
            ops.emb_sparse_apply(bf['skeys'], bf['ssrc'], n_recv, N=1, E=E, total_rows=total, dz=bf['rows'][:n_recv],
                                 opt=opt, tables=t.weight, slot0=t.slot0, slot1=t.slot1, ws=ws)
            ops.adam_untouched_rows(bf['skeys'], n_recv, total_rows=total, E=E, opt=opt, tables=t.weight,
                                    slot0=t.slot0, slot1=t.slot1, touched=t.touched)
        elif opt.kind == OPT_ADAM and not lazy:
            t.grad.zero_()
            ops.emb_sparse_apply(bf['skeys'], bf['ssrc'], n_recv, N=1, E=E, total_rows=total, dz=bf['rows'][:n_recv],
                                 opt=ops.make_opt(OPT_GRAD_ONLY), slot0=t.grad, ws=ws)
            ops.dense_update(t.weight, t.grad, opt, t.slot0, t.slot1)
        else:
            ops.emb_sparse_apply(bf['skeys'], bf['ssrc'], n_recv, N=1, E=E, total_rows=total, dz=bf['rows'][:n_recv],
                                 opt=opt, tables=t.weight, slot0=t.slot0, slot1=t.slot1, ws=ws)

    def backward_cont_w(self, dz, x_cont, g_cont_w, ws):
        E = self.E
        self.ops.colsum(dz[:, self.Ncat * E:(self.Ncat + self.C) * E], g_cont_w.reshape(-1), w=x_cont,
                        w_group=E, ws=ws)

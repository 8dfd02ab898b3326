"""CPU (gloo, world_size 2) test of the sharded-embedding exchange plan: split sizes, buffer
layouts and block assembly of recommendation_amd.distributed.ShardedFront, with a torch stand-in
for the device kernels (the stand-in lives here, in the tests; the product has no CPU path)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class TorchBackend:
    """Minimal CPU restatement of the operators ShardedFront calls."""

    class _Opt:
        def __init__(self, kind, lr=0.0):
            self.kind, self.lr = kind, lr

    def make_opt(self, kind, lr=0.0, **kw):
        return self._Opt(kind, lr)

    def embed_fwd(self, tables, row_offset, idx, x_cont=None, cont_w=None, *, E, z_out=None, cont_tail=False,
                  oor_count=None, **kw):
        B, F = idx.shape
        for f in range(F):
            off = int(row_offset[f])
            V = int(row_offset[f + 1]) - off
            x = idx[:, f].long()
            ok = (x >= 0) & (x < V)
            rows = tables[off + torch.where(ok, x, torch.zeros_like(x))] * ok.float().view(-1, 1)
            z_out[:, f * E:(f + 1) * E] = rows

    class Workspace:
        def __init__(self, device):
            self.device = device

    def shard_route(self, idx, row_offset, local_offset, W, dest_out, lkey_out, oor_count=None):
        B, F = idx.shape
        x = idx.long()
        V = (row_offset[1:] - row_offset[:-1]).view(1, F)
        ok = (x >= 0) & (x < V)
        fcol = torch.arange(F).view(1, F).expand(B, F)
        d = torch.where(ok, (x + fcol) % W, fcol % W)
        key = torch.where(ok, local_offset.long()[d, torch.arange(F).view(1, F).expand(B, F)] + x // W,
                          torch.full_like(x, -1))
        dest_out[:B * F] = d.reshape(-1).int()
        lkey_out[:B * F] = key.reshape(-1).int()
        if oor_count is not None:
            oor_count += int((~ok).sum())

    def sort_pairs(self, keys, vals, key_bits, ws):
        order = torch.sort(keys.long(), stable=True).indices
        return keys[order], vals[order]

    def rows_gather(self, src, perm, n, out, *, F, E):
        p = perm[:n].long()
        live = p >= 0                                # -1 = an empty slot: zero row
        ps = torch.where(live, p, torch.zeros_like(p))
        flat = out.view(-1)
        for e in range(E):
            v = src[ps // F, (ps % F) * E + e]
            flat[torch.arange(n) * E + e] = torch.where(live, v, torch.zeros_like(v))

    def rows_scatter(self, rows, perm, n, dst, *, F, E):
        p = perm[:n].long()
        live = p >= 0
        r = rows.reshape(-1, E)[:n][live]
        p = p[live]
        for e in range(E):
            dst[p // F, (p % F) * E + e] = r[:, e]

    def shard_bucket(self, dest_sorted, perm, lkey, n, W, cap, slot_key, slot_perm, overflow, z_zero=None, F=1, E=1):
        slot_key[:W * cap] = -1
        slot_perm[:W * cap] = -1
        d = dest_sorted[:n].long()
        start = torch.searchsorted(d, torch.arange(W))
        r = torch.arange(n) - start[d]
        ok = r < cap
        slot = (d * cap + r)[ok]
        slot_key[slot] = lkey[perm[:n].long()[ok]]
        slot_perm[slot] = perm[:n][ok]
        overflow += int((~ok).sum())
        if z_zero is not None:
            for p in perm[:n].long()[~ok].tolist():
                z_zero[p // F, (p % F) * E:(p % F + 1) * E] = 0

    def copy2d(self, src, dst, accumulate=False):
        dst[:, :src.shape[1]] = src

    def fm_fwd(self, z, **kw):
        raise AssertionError('not used in this test')

    def embed_assemble_fwd(self, blocks, fbase, fstride, x_cont=None, cont_w=None, *, F, E, z_out, cont_tail=False,
                           **kw):
        B = z_out.shape[0]
        rows = torch.arange(B).view(B, 1)
        for n in range(F):
            src = int(fbase[n]) + rows * int(fstride[n]) + torch.arange(E).view(1, E)
            z_out[:, n * E:(n + 1) * E] = blocks[src]
        if cont_tail:
            z_out[:, F * E:F * E + x_cont.shape[1]] = x_cont

    def emb_sort(self, idx, row_offset, *, N, total_rows, rmode, keys_out, src_out, ws):
        self._idx, self._off = idx.clone(), row_offset

    def emb_sparse_apply(self, keys, src, n_items, *, N, E, total_rows, dz, opt, ws, tables=None, slot0=None,
                         slot1=None, src_scale=None):
        idx = self._idx
        for f in range(idx.shape[1]):
            off = int(self._off[f])
            ok = idx[:, f] >= 0
            tables.index_add_(0, off + idx[ok, f].long(), -opt.lr * dz[:idx.shape[0]][ok, f * E:(f + 1) * E])


def p2p_all_to_all(out, inp, out_splits, in_splits, rank, world):
    """all-to-all-v from point-to-point sends (gloo has no native alltoall)."""
    outs, ins = list(out.split(out_splits)), list(inp.split(in_splits))
    outs[rank].copy_(ins[rank])
    reqs = []
    for j in range(world):
        if j != rank:
            if ins[j].numel():
                reqs.append(dist.isend(ins[j].contiguous(), j))
            if outs[j].numel():
                reqs.append(dist.irecv(outs[j], j))
    for r in reqs:
        r.wait()


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from recommendation_amd.distributed import ShardedFront, pack_fields

    class Comm:
        def __init__(self):
            self.world, self.rank = world, rank

        def all_to_all(self, out, inp, out_splits, in_splits):
            p2p_all_to_all(out, inp, out_splits, in_splits, rank, world)

        def all_reduce_sum(self, t):
            dist.all_reduce(t)

    cards, E, C, B = [5, 9, 4, 7, 3], 4, 2, 6
    rng = np.random.default_rng(0)                       # same tables on both ranks
    full = [torch.tensor(rng.standard_normal((v, E)), dtype=torch.float32) for v in cards]
    front = ShardedFront(cards, E, C, False, Comm(), torch.device('cpu'), backend=TorchBackend())
    own = front.own_fields
    assert own == pack_fields(cards, world)[rank] and own == ([1, 2] if rank == 0 else [0, 3, 4])
    front.t.weight.copy_(torch.cat([full[f] for f in own]))
    r2 = np.random.default_rng(100 + rank)               # different batch per rank
    idx = torch.tensor(np.stack([r2.integers(0, v, B) for v in cards], 1), dtype=torch.int32)
    xc = torch.tensor(r2.standard_normal((B, C)), dtype=torch.float32)
    z = torch.zeros(B, len(cards) * E + C)
    front.forward(idx, xc, z_out=z, cont_tail=True)
    ref = torch.cat([full[f][idx[:, f].long()] for f in range(5)] + [xc], dim=1)
    assert torch.equal(z, ref)                           # rows are copies: exact
    # backward: SGD on the owners with gradients from BOTH ranks' batches
    dz = torch.tensor(r2.standard_normal((B, len(cards) * E)), dtype=torch.float32)
    front.backward_tables(idx, dz, None, None, None, TorchBackend._Opt(3, lr=0.5), lazy=True)
    all_idx = [torch.zeros_like(idx) for _ in range(world)]
    all_dz = [torch.zeros_like(dz) for _ in range(world)]
    dist.all_gather(all_idx, idx)
    dist.all_gather(all_dz, dz)
    for fl, f in enumerate(own):
        exp = full[f].clone()
        for i2, d2 in zip(all_idx, all_dz):
            exp.index_add_(0, i2[:, f].long(), -0.5 * d2[:, f * E:(f + 1) * E])
        got = front.t.table(fl)
        assert torch.allclose(got, exp, atol=1e-6), (rank, f)
    dist.barrier()
    dist.destroy_process_group()


def _row_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from recommendation_amd.distributed import RowShardedFront, row_shard_layout

    class Comm:
        def __init__(self):
            self.world, self.rank = world, rank

        def all_to_all(self, out, inp, out_splits=None, in_splits=None):
            out_splits = out_splits or [out.shape[0] // world] * world
            in_splits = in_splits or [inp.shape[0] // world] * world
            p2p_all_to_all(out, inp, out_splits, in_splits, rank, world)

        def all_reduce_sum(self, t):
            dist.all_reduce(t)

    cards, E, C, B = [5, 9, 1, 7, 3], 4, 2, 6               # field 2 has one row: rank 1 owns none of it
    rng = np.random.default_rng(0)
    full = [torch.tensor(rng.standard_normal((v, E)), dtype=torch.float32) for v in cards]
    front = RowShardedFront(cards, E, C, False, Comm(), torch.device('cpu'), backend=TorchBackend())
    loc_cards, _ = row_shard_layout(cards, world)
    assert [sum(c[f] for c in loc_cards) for f in range(5)] == cards
    for f in range(5):
        assert list(front.local_rows(f)) == [r for r in range(cards[f]) if (r + f) % world == rank]
        front.t.table(f).copy_(full[f][(rank - f) % world::world])
    r2 = np.random.default_rng(100 + rank)
    idx = torch.tensor(np.stack([r2.integers(0, v, B) for v in cards], 1), dtype=torch.int32)
    idx[0, 1] = 99                                           # out of range: zero row, no update
    xc = torch.tensor(r2.standard_normal((B, C)), dtype=torch.float32)
    z = torch.zeros(B, len(cards) * E + C)
    front.oor_count = torch.zeros(1, dtype=torch.int32)
    front.forward(idx, xc, z_out=z, cont_tail=True)
    assert int(front.oor_count) == 1                         # counted on the rank whose batch holds the id
    safe = idx.clone()
    safe[0, 1] = 0
    ref = torch.cat([full[f][safe[:, f].long()] for f in range(5)] + [xc], dim=1)
    ref[0, E:2 * E] = 0
    assert torch.equal(z, ref)
    dz = torch.tensor(r2.standard_normal((B, len(cards) * E)), dtype=torch.float32)
    front.backward_tables(idx, dz, None, None, None, TorchBackend._Opt(3, lr=0.5), lazy=True)
    all_idx = [torch.zeros_like(idx) for _ in range(world)]
    all_dz = [torch.zeros_like(dz) for _ in range(world)]
    dist.all_gather(all_idx, idx)
    dist.all_gather(all_dz, dz)
    for f in range(5):
        exp = full[f].clone()
        for i2, d2 in zip(all_idx, all_dz):
            ok = i2[:, f] < cards[f]
            exp.index_add_(0, i2[ok, f].long(), -0.5 * d2[ok, f * E:(f + 1) * E])
        assert torch.allclose(front.t.table(f), exp[(rank - f) % world::world], atol=1e-6), (rank, f)
    assert int(front.overflow) == 0
    # a bucket that cannot hold its share is counted, not silently dropped
    tight = RowShardedFront(cards, E, C, False, Comm(), torch.device('cpu'), backend=TorchBackend(), slack=0.1)
    tight.capacity = lambda B: 1
    hot = torch.zeros(B, len(cards), dtype=torch.int32)     # every lookup of a field goes to one rank
    if rank == 1:
        hot = idx.clone()                                    # rank 1's own buckets may or may not overflow ...
    stale = torch.full_like(z, 7.0)                          # ... and a dropped lookup must not keep a stale row
    tight.forward(hot, xc, z_out=stale, cont_tail=True)
    if rank == 0:
        assert int(tight.overflow) > 0
        rows = stale[:, :len(cards) * E].reshape(B, len(cards), E)
        assert bool(((rows == 0).all(-1) | (rows != 7.0).all(-1)).all())     # every row was written or zeroed
    with pytest.raises(RuntimeError):                        # ... yet both ranks raise, together
        tight.check_overflow()
    dist.barrier()
    dist.destroy_process_group()


def test_row_sharded_front_plan_world2():
    mp.spawn(_row_worker, args=(2, 29743, None), nprocs=2, join=True)


def test_sharded_front_plan_world2():
    mp.spawn(_worker, args=(2, 29741, None), nprocs=2, join=True)


def test_pack_fields_balances_rows():
    from recommendation_amd.distributed import pack_fields
    from recommendation_amd.synthetic import CRITEO_1TB
    packs = pack_fields(CRITEO_1TB, 8)
    assert sorted(f for p in packs for f in p) == list(range(26))
    loads = [sum(CRITEO_1TB[f] for f in p) for p in packs]
    big = [f for f in range(26) if CRITEO_1TB[f] > 100_000_000]          # four tables above 100 M rows
    assert len({next(j for j, p in enumerate(packs) if f in p) for f in big}) == 4
    assert max(loads) == max(CRITEO_1TB)                                 # the largest table alone sets the maximum
    assert pack_fields([5, 9, 4, 7, 3], 2) == [[1, 2], [0, 3, 4]]
    assert pack_fields([3, 2, 1], 4) == [[0], [1], [2], []]

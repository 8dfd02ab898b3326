"""One-rank check of the library's own collectives (ctr_comm_init / ctr_all2all / ctr_all2allv / ctr_allreduce_sum_f32
over RCCL) in a process of its own: raw calls through the C ABI, then both sharded front ends driven through
distributed.CtrComm against the unsharded front end, bit for bit.  No torch.distributed group is involved."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    torch.cuda.set_device(0)
    import recommendation_amd as ra
    from recommendation_amd import _lib, ops
    from recommendation_amd._lib import OPT_SGD
    from recommendation_amd.distributed import CtrComm, RowShardedFront, ShardedFront
    from recommendation_amd.engine import EmbedFront, EmbeddingTables
    sess = ra.Session('cuda:0', seed=1)
    dev = sess.device
    lib = _lib.load()
    # raw ABI: id, init, the three collectives on one rank (every piece goes to the rank itself), destroy
    ident = C.create_string_buffer(_lib.COMM_ID_BYTES)
    assert lib.ctr_comm_unique_id(ident) == 0, lib.ctr_last_error()
    h = C.c_void_p()
    assert lib.ctr_comm_init(ident, 1, 0, C.byref(h)) == 0, lib.ctr_last_error()
    assert lib.ctr_comm_world(h) == 1 and lib.ctr_comm_rank(h) == 0
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    a = torch.arange(1000, dtype=torch.float32, device=dev)
    b = torch.zeros_like(a)
    assert lib.ctr_all2all(h, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), 4000, st) == 0, lib.ctr_last_error()
    off, nb = (C.c_int64 * 1)(400), (C.c_int64 * 1)(1200)
    roff = (C.c_int64 * 1)(0)
    c = torch.zeros(300, dtype=torch.float32, device=dev)
    assert lib.ctr_all2allv(h, C.c_void_p(a.data_ptr()), off, nb, C.c_void_p(c.data_ptr()), roff, nb, st) == 0, lib.ctr_last_error()
    g = torch.arange(8, dtype=torch.float32, device=dev)
    assert lib.ctr_allreduce_sum_f32(h, C.c_void_p(g.data_ptr()), 8, st) == 0, lib.ctr_last_error()
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(c, a[100:400]) and torch.equal(g.cpu(), torch.arange(8, dtype=torch.float32))
    assert lib.ctr_comm_init(ident, 2, 5, C.byref(h)) != 0 and b'rank 5 of 2' in lib.ctr_last_error()
    assert lib.ctr_comm_destroy(h) == 0
    # the sharded front ends over CtrComm
    comm = CtrComm(world=1, rank=0)
    cards, E, Cn, B = [7, 50, 3, 11, 200], 8, 3, 96
    rng = np.random.default_rng(0)
    W0 = torch.tensor(rng.standard_normal((sum(cards), E)), dtype=torch.float32, device=dev)
    idx = torch.tensor(np.stack([rng.integers(0, v, B) for v in cards], 1), dtype=torch.int32, device=dev)
    idx[3, 1] = 77
    xc = torch.tensor(rng.standard_normal((B, Cn)), dtype=torch.float32, device=dev)
    dz = torch.tensor(rng.standard_normal((B, len(cards) * E)), dtype=torch.float32, device=dev)
    opt = ops.make_opt(OPT_SGD, lr=0.5)
    outs = []
    for kind in ('plain', 'table', 'row'):
        if kind == 'plain':
            t = EmbeddingTables(cards, E, dev)
            front = EmbedFront(t, 'S', Cn, False)
        else:
            front = (ShardedFront if kind == 'table' else RowShardedFront)(cards, E, Cn, False, comm, dev)
            t = front.t
        t.weight[:sum(cards)].copy_(W0)
        z = torch.zeros(B, (len(cards) * E + Cn + 3) // 4 * 4, device=dev)
        front.forward(idx, xc, z_out=z, cont_tail=True)
        keys = torch.zeros(B * len(cards), dtype=torch.int32, device=dev)
        src = torch.zeros_like(keys)
        front.backward_tables(idx, dz, keys, src, sess.ws, opt, lazy=True)
        if hasattr(front, 'check_overflow'):
            front.check_overflow()
        torch.cuda.synchronize()
        outs.append((z.cpu().numpy(), t.weight[:sum(cards)].cpu().numpy()))
    for z, w in outs[1:]:
        assert np.array_equal(z, outs[0][0]) and np.array_equal(w, outs[0][1])
    cnt = torch.tensor([3], dtype=torch.int32, device=dev)
    comm.all_reduce_sum(cnt)
    assert int(cnt.item()) == 3
    comm.close()
    print('CTR_COMM_OK')


if __name__ == '__main__':
    main()

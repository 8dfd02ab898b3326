"""One-rank RCCL check (launched by torch.distributed.run with the nccl backend = RCCL): both sharded front ends
driven through distributed.TorchComm - all_to_all_single with explicit and with equal splits, all_reduce - must
reproduce the unsharded front end bit for bit (rows are copies; one rank keeps the summation order)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', device_id=torch.device('cuda:0'))
    import recommendation_amd as ra
    from recommendation_amd import ops
    from recommendation_amd._lib import OPT_SGD
    from recommendation_amd.distributed import RowShardedFront, ShardedFront, TorchComm
    from recommendation_amd.engine import EmbedFront, EmbeddingTables
    sess = ra.Session('cuda:0', seed=1)
    dev = sess.device
    comm = TorchComm()
    assert comm.world == 1
    cards, E, C, B = [7, 50, 3, 11, 200], 8, 3, 96
    rng = np.random.default_rng(0)
    W0 = torch.tensor(rng.standard_normal((sum(cards), E)), dtype=torch.float32, device=dev)
    idx = torch.tensor(np.stack([rng.integers(0, v, B) for v in cards], 1), dtype=torch.int32, device=dev)
    idx[3, 1] = 77                                       # out of range: zero row, no update
    xc = torch.tensor(rng.standard_normal((B, C)), dtype=torch.float32, device=dev)
    dz = torch.tensor(rng.standard_normal((B, len(cards) * E)), dtype=torch.float32, device=dev)
    opt = ops.make_opt(OPT_SGD, lr=0.5)
    outs = []
    for kind in ('plain', 'table', 'row'):
        if kind == 'plain':
            t = EmbeddingTables(cards, E, dev)
            front = EmbedFront(t, 'S', C, False)
        else:
            front = (ShardedFront if kind == 'table' else RowShardedFront)(cards, E, C, False, comm, dev)
            t = front.t
        t.weight[:sum(cards)].copy_(W0)
        z = torch.zeros(B, len(cards) * E + C + 1, device=dev)[:, :len(cards) * E + C]
        z = torch.zeros(B, (len(cards) * E + C + 3) // 4 * 4, device=dev)
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
    g = torch.arange(8, dtype=torch.float32, device=dev)
    comm.all_reduce_sum(g)
    assert torch.equal(g.cpu(), torch.arange(8, dtype=torch.float32))
    print('RCCL_OK backend=%s' % dist.get_backend())
    dist.destroy_process_group()


if __name__ == '__main__':
    main()

"""Worker of the 2-rank equivalence test (launched by torch.distributed.run, gloo group).

Both ranks share cuda:0 on the one-GPU box, so the collectives are staged through host memory
over gloo (test-only transport; production uses RCCL through distributed.TorchComm).  Each rank
trains on its half of a global batch; the result must match a single-rank model trained on the
whole batch: dense variables on every rank, table shards on their owners."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _checkpoint_round_trip(cls, cfg, dcfg, m, single, sess, Xc, Xf, y, lo, hi, rank, world, shard):
    """Per-rank shard files + manifest: save, restore into a fresh sharded model and continue bit-identically;
    restore the same files into a one-process model (re-sharding) and the one-process file into a sharded model."""
    import copy
    import tempfile
    import numpy as np
    F, C = cfg.cate_num, cfg.cont_num
    d = [tempfile.mkdtemp() if rank == 0 else None]
    dist.broadcast_object_list(d, src=0)
    path = os.path.join(d[0], 'model.ckpt')
    m.saver.save(sess, path)
    dist.barrier()
    assert os.path.exists('%s.rank%d-of-%d.safetensors' % (path, rank, world)) and os.path.exists(path + '.manifest.json')
    m2 = cls(copy.copy(dcfg), F, C, cfg.cate_list)
    m2.build()
    m2.initialize(sess, seed=99)                      # other weights, then overwritten by the restore
    m2.saver.restore(sess, path)
    assert m2.global_step == m.global_step
    for _ in range(2):                                # continue training: the restored model follows the original bit for bit
        a = m.run(sess, Xc[lo:hi], Xf[lo:hi], y[lo:hi], True)
        b = m2.run(sess, Xc[lo:hi], Xf[lo:hi], y[lo:hi], True)
        assert a == b
    for (k, v), v2 in zip(m.get_params().items(), m2.get_params().values()):
        assert np.array_equal(v, v2), (rank, k)
    # the sharded checkpoint into ONE process (every rank does it; they must agree with the one-process reference)
    solo = cls(cfg, F, C, cfg.cate_list)
    solo.build()
    solo.initialize(sess, seed=5)
    solo.saver.restore(sess, path)
    ref = single.get_params()                         # `single` took the same three steps on the whole batch
    from tests.helpers import rel_err
    for k, v in solo.get_params().items():
        assert rel_err(v, ref[k]) < 2e-5, (rank, k)
    assert solo.tables.slot0 is None or float(solo.tables.slot0.abs().sum()) > 0
    # and a one-process checkpoint into the sharded layout
    if rank == 0:
        single.saver.save(sess, os.path.join(d[0], 'solo.ckpt'))
    dist.barrier()
    m3 = cls(copy.copy(dcfg), F, C, cfg.cate_list)
    m3.build()
    m3.initialize(sess, seed=7)
    m3.saver.restore(sess, os.path.join(d[0], 'solo.ckpt'))
    for k, v in m3.get_params().items():
        want = ref[k]
        if shard == 'row' and k.startswith('embedding/embed_cate_w_'):
            f = int(k.rsplit('_', 1)[1])
            want = want[(rank - f) % world::world]
        assert np.array_equal(v, want), (rank, k)
    s0 = single.tables.slot0
    if s0 is not None:
        for fl, f in enumerate(m3._own_fields):
            want = single.tables.table(f, s0).cpu().numpy()
            if shard == 'row':
                want = want[(rank - f) % world::world]
            assert np.array_equal(m3.tables.table(fl, m3.tables.slot0).cpu().numpy(), want), (rank, f)


def _bad_ids_and_overflow(cls, dcfg, cfg, P, sess, Xc, Xf, y, lo, hi, rank, world, shard):
    """An id outside its table in ONE rank's batch is counted once over the ranks and, with strict_ids, raises on
    every rank; a send bucket that overflows on one rank raises on every rank (the ranks stay in step)."""
    import copy
    F, C = cfg.cate_num, cfg.cont_num
    bad = Xc[lo:hi].copy()
    if rank == 0:
        bad[1, 3] = 10 ** 6
    m = cls(copy.copy(dcfg), F, C, cfg.cate_list)
    m.build()
    m.initialize(sess, params=P)
    m.run(sess, bad, Xf[lo:hi], y[lo:hi], True)
    assert m.out_of_range_lookups() == 1 and m.out_of_range_lookups() == 0, rank
    scfg = copy.copy(dcfg)
    scfg.strict_ids = True
    ms = cls(scfg, F, C, cfg.cate_list)
    ms.build()
    ms.initialize(sess, params=P)
    try:
        ms.run(sess, bad, Xf[lo:hi], y[lo:hi], True)
        raise AssertionError('strict_ids did not raise on rank %d' % rank)
    except IndexError:
        pass
    if shard == 'row':
        tcfg = copy.copy(dcfg)
        tcfg.shard_slack = 1.0                    # 120 lookups per rank, 64 slots per destination
        mt = cls(tcfg, F, C, cfg.cate_list)
        mt.build()
        mt.initialize(sess, params=P)
        nb = 24
        fcol = np.arange(F)[None, :]
        # rank 0: every id 0, so three of the five fields go to one rank (72 lookups > 64 slots);
        # rank 1: destinations alternate with the sample (60 + 60): nothing overflows there
        hot = np.zeros((nb, F), dtype=np.int64) if rank == 0 else (np.arange(nb)[:, None] + fcol) % 2
        xf, yy = np.zeros((nb, C)), np.zeros(nb)
        mt.train_step(*mt._upload(hot, xf, yy))
        own = int(mt.front.overflow.item())
        assert (own == 0) == (rank == 1), (rank, own)          # only rank 0 dropped lookups ...
        try:
            mt.check_overflow()
            raise AssertionError('bucket overflow did not raise on rank %d' % rank)
        except RuntimeError as e:                              # ... and both ranks raise
            assert 'send buckets' in str(e), e
    dist.barrier()


def main(model_name, shard='table', arch='tiny', dense_allreduce='ring'):
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    import recommendation_amd as ra
    from recommendation_amd.distributed import HostStagedComm
    from oracle import ctr_oracle as O
    from tests.helpers import rel_err
    ra.enable_zoo_imports()
    import importlib
    cls = getattr(importlib.import_module('%s.%s' % (model_name, model_name)), model_name)
    cfg = O.default_cfg(model_name, embed_size=8, mode='S')
    cfg.cate_list = [7, 50, 3, 11, 200]
    cfg.cate_num, cfg.cont_num = 5, 3
    for k in ('dropout_fm', 'dropout_deep', 'dropout_bi', 'dropout'):
        if hasattr(cfg, k):
            setattr(cfg, k, 0.0)
    if hasattr(cfg, 'deep_layers'):
        cfg.deep_layers = [16, 8]
    if model_name == 'PNN':
        cfg.hidden_layers, cfg.product_size = [16, 8], 6
    cfg.adam_mode = 'tf1'
    std, Bg = 0.2, 64
    if arch == 'default':        # the reference's default tower on a Criteo-like layout: chained / fused kernels, E = 16
        cfg = O.default_cfg(model_name, embed_size=16, mode='S')
        cfg.cate_list = [int(v) for v in np.random.default_rng(11).integers(3, 200, 26)]
        cfg.cate_num, cfg.cont_num = 26, 13
        for k in ('dropout_fm', 'dropout_deep', 'dropout_bi', 'dropout'):
            if hasattr(cfg, k):
                setattr(cfg, k, 0.0)
        cfg.adam_mode = 'tf1'
        std, Bg = 0.05, 256
    F, C = cfg.cate_num, cfg.cont_num
    P = O.init_params(model_name, cfg, seed=1, std=std)
    rng = np.random.default_rng(5)
    Xc = np.stack([rng.integers(0, v, Bg) for v in cfg.cate_list], 1)
    Xf = rng.standard_normal((Bg, C))
    y = rng.integers(0, 2, Bg)
    sess = ra.Session('cuda:0', seed=3)
    import copy
    dcfg = copy.copy(cfg)
    dcfg.world_size, dcfg.rank, dcfg.comm, dcfg.shard = world, rank, HostStagedComm(), shard
    if dense_allreduce == 'nobuf':                # tf.train.AdamOptimizer's every-row update without the dense gradient buffer
        cfg.table_grads = dcfg.table_grads = False
        dense_allreduce = 'ring'
    dcfg.dense_allreduce = dense_allreduce        # 'direct': reduce-scatter + all-gather from two all-to-alls
    m = cls(dcfg, F, C, cfg.cate_list)
    m.build()
    m.initialize(sess, params=P)
    lo, hi = rank * Bg // world, (rank + 1) * Bg // world
    for _ in range(3):
        m.run(sess, Xc[lo:hi], Xf[lo:hi], y[lo:hi], True)
    single = cls(cfg, F, C, cfg.cate_list)
    single.build()
    single.initialize(sess, params=P)
    for _ in range(3):
        single.run(sess, Xc, Xf, y, True)
    ref = single.get_params()
    got = m.get_params()
    n_tab = 0
    for k, v in got.items():
        want = ref[k]
        if shard == 'row' and k.startswith('embedding/embed_cate_w_'):
            f = int(k.rsplit('_', 1)[1])
            want = want[(rank - f) % world::world]         # this rank holds rows r with (r + f) % world == rank
        assert v.shape == want.shape, (rank, k, v.shape, want.shape)
        assert rel_err(v, want) < 2e-5, (rank, k, rel_err(v, want))
        n_tab += k.startswith('embedding/embed_cate_w_')
    assert n_tab == (F if shard == 'row' else m.front.nf_own)
    dist.barrier()
    if arch == 'tiny':
        _checkpoint_round_trip(cls, cfg, dcfg, m, single, sess, Xc, Xf, y, lo, hi, rank, world, shard)
        _bad_ids_and_overflow(cls, dcfg, cfg, P, sess, Xc, Xf, y, lo, hi, rank, world, shard)
    dist.barrier()
    if rank == 0:
        print('DIST_OK %s %s' % (model_name, shard))
    dist.destroy_process_group()


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else 'table', sys.argv[3] if len(sys.argv) > 3 else 'tiny',
         sys.argv[4] if len(sys.argv) > 4 else 'ring')

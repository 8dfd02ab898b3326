#!/usr/bin/env python
"""Headline benchmark: DeepFM training step on Criteo-Kaggle-shape synthetic data.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (embedding gather + FM + tower forward, loss, backward,
sorted segment-reduce + sparse Adam, dense Adam) over one batch that already sits in HBM.
Rank 0 prints ONE JSON line (see README/DESIGN for the fields).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PROFILE_STEPS = 5
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec
MFMA_F32_PEAK_TFLOPS = 157.3   # fp32-input MFMA (exact fp32) dense peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak; the split GEMM spends 6 bf16 MFMAs per fp32 product


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=0, help='per-GPU batch (default: 65536 DeepFM, 8192 DCN)')
    ap.add_argument('--embed', type=int, default=0, help='embedding dim (default: 16 DeepFM, 64 DCN)')
    ap.add_argument('--alpha', type=float, default=1.05, help='Zipf exponent of the ids (0 = uniform)')
    ap.add_argument('--model', default='DeepFM', choices=sorted(MODEL_ARGS),
                    help='DeepFM is the headline workload; the others time the same step for their family')
    ap.add_argument('--comm', default='rccl', choices=['rccl', 'host-staged'],
                    help='host-staged: gloo through host memory, to rehearse N>1 with all ranks on one GPU')
    ap.add_argument('--shard', default=None, choices=['table', 'row'],
                    help='N>1: whole tables per rank (default for the Criteo-Kaggle models), or row r of field f on rank '
                         '(r + f) %% N (default for DCN: its Criteo-1TB tables do not fit one owner)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timers', action='store_true')
    return ap.parse_args()


# reference flag defaults of each model's main.py (DeepFM/main.py:48-59 and siblings); the DCN entry is
# BASELINE configs[2] (6 cross layers, 1024x3 tower), the others run on the Criteo-Kaggle shape
MODEL_ARGS = {
    'DeepFM': dict(deep_layers=[256, 128, 64, 32], dropout_fm=0.5, dropout_deep=0.2),
    'DCN': dict(deep_layers=[1024, 1024, 1024], cross_layer_num=6, dropout=0.2),
    'NFM': dict(deep_layers=[256, 128, 64, 32], dropout_bi=0.5, dropout_deep=0.2),
    'ESMM': dict(deep_layers=[256, 128, 64, 32], dropout=0.2),
    'PNN': dict(inner_product=True, product_size=50, hidden_layers=[32, 64, 128], dropout=0.2),
    'PNN_outer': dict(inner_product=False, product_size=50, hidden_layers=[32, 64, 128], dropout=0.2),
    'AFM': dict(attention_size=10, dropout=0.2),
    'AFM_pairs': dict(attention_size=10, dropout=0.2, attention='pairs'),
}


def build_model(args, sess, cards, rank, world):
    import importlib
    import recommendation_amd as ra
    ra.enable_zoo_imports()
    name = args.model.split('_')[0]
    margs = SimpleNamespace(embed_size=args.embed, regular_rate=0.1, learning_rate=0.001, decay_steps=10000,
                            decay_rate=0.99, mode='S', adam_mode='lazy', init_std=0.01, **MODEL_ARGS[args.model])
    cls = getattr(importlib.import_module('%s.%s' % (name, name)), name)
    if world > 1:
        margs.world_size, margs.rank, margs.shard = world, rank, args.shard
        if args.comm == 'host-staged':
            from recommendation_amd.distributed import HostStagedComm
            margs.comm = HostStagedComm()
    m = cls(margs, len(cards), 13, cards)
    m.build()
    m.initialize(sess, seed=1234)
    return m


def cpu_baseline(args, cards):
    """The oracle (CPU restatement of the reference graph; TensorFlow is unavailable) timed on
    the host's cores on a bounded sample of the same workload."""
    from oracle import ctr_oracle as O
    from recommendation_amd.synthetic import scaled
    threads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    div = 64
    small = scaled(cards, div)
    B = 4096
    cfg = O.default_cfg('DeepFM', embed_size=args.embed, mode='S')
    cfg.cate_num, cfg.cont_num, cfg.cate_list = len(small), 13, small
    rng = np.random.default_rng(0)
    P = O.init_params('DeepFM', cfg, seed=0)
    m = O.OracleModel('DeepFM', cfg, P, dtype=torch.float32)
    Xc = np.stack([rng.integers(0, v, B) for v in small], 1)
    Xf = rng.standard_normal((B, 13)).astype(np.float32)
    y = rng.integers(0, 2, B)
    m.run(Xc, Xf, y, True)
    t0 = time.perf_counter()
    n = 0
    while n < 3 or (time.perf_counter() - t0 < 10.0 and n < 50):
        m.run(Xc, Xf, y, True)
        n += 1
    dt = time.perf_counter() - t0
    return dict(value=B * n / dt, unit='samples/s', cores=threads, kind='port',
                sample='CPU restatement of the reference graph (TensorFlow unavailable): DeepFM S-mode, '
                       'torch-CPU fp32, B=%d, %d steps, table rows scaled 1/%d (sum V=%d), TF1 dense Adam'
                       % (B, n, div, int(np.sum(small))))


def load_traffic():
    """HBM bytes per launch from the committed rocprofv3 PMC pass (FETCH_SIZE/WRITE_SIZE collected in
    separate runs, gfx950 x2 read correction; see tools/pmc_summary.py), keyed by kernel symbol."""
    path = os.path.join(ROOT, 'profiles', 'r01_g_deepfm_c2_pmc_traffic.json')
    out = {}
    if os.path.exists(path):
        for k, v in json.load(open(path)).items():
            for sym in ('embed_fwd_kernel', 'sparse_apply_kernel', 'embed_bwd_dz_kernel'):
                if sym + '<' in k or sym + '(' in k:
                    out[sym] = out.get(sym, 0) + v['hbm_bytes_per_launch']
    return out


def main():
    args = parse()
    if not args.batch:
        args.batch = 8192 if args.model == 'DCN' else 65536
    if not args.embed:
        args.embed = 64 if args.model == 'DCN' else 16
    if args.shard is None:
        args.shard = 'row' if args.model == 'DCN' else 'table'
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d '
                             '--master-addr 127.0.0.1 --master-port 29500 bench.py --gpus %d ...'
                             % (args.gpus, args.gpus))
        raise SystemExit('--gpus %d does not match WORLD_SIZE %d' % (args.gpus, world))
    import recommendation_amd as ra
    from recommendation_amd import ops
    from recommendation_amd.synthetic import CRITEO_1TB, CRITEO_KAGGLE, make_batches, scaled

    if args.comm == 'host-staged':
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = 'cuda:%d' % local_rank
    if world > 1:
        import torch.distributed as dist
        if args.comm == 'host-staged':
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device(dev))
    sess = ra.Session(dev, seed=11 + rank)
    # DeepFM: BASELINE configs[1] (Criteo-Kaggle cardinalities).  DCN: configs[2] per-GPU share —
    # Criteo-1TB cardinalities / 8 (one GPU of the 8-GPU shard), dim 64, 6 cross layers, 1024x3 tower.
    cards = CRITEO_KAGGLE if args.model != 'DCN' else scaled(CRITEO_1TB, 8 // min(world, 8))
    shape_name = 'Criteo-Kaggle' if args.model != 'DCN' else 'Criteo-1TB/%d' % (8 // min(world, 8))
    if rank == 0:
        print('[bench] building %s: %d table rows on %d rank(s)' % (args.model, sum(cards), world), file=sys.stderr, flush=True)
    model = build_model(args, sess, cards, rank, world)
    if rank == 0:
        print('[bench] model built, generating batches', file=sys.stderr, flush=True)
    B = args.batch
    batches = make_batches(cards, 13, B, 8, args.alpha, 20240601 + rank, dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # distinct table rows touched per step (for the algorithmic bytes of the sparse update)
    offs = torch.as_tensor(np.concatenate([[0], np.cumsum(cards)[:-1]]), device=dev)
    uniq = float(np.mean([torch.unique(b[0].to(torch.int64) + offs).numel() for b in batches]))
    if rank == 0:
        print('[bench] warm-up', file=sys.stderr, flush=True)
    for i in range(args.warmup):
        model.train_step(*batches[i % len(batches)])
    # Untimed profiling pass: every operator between HIP events, to split the step by kernel and to
    # find the dominant one.  The timed region then keeps events only around that kernel and the two
    # embedding kernels (a pair of events per operator costs ~10 % of a 1.5 ms step).
    profile, timer = None, None
    if not args.no_kernel_timers:
        prof = ops.KernelTimer(pool=PROFILE_STEPS * 160)
        ops.set_timer(prof)
        sess.side_enabled = False      # one stream: every operator's time is its own, not a shared GPU's
        for i in range(PROFILE_STEPS):
            model.train_step(*batches[i % len(batches)])
        sess.side_enabled = True
        ops.set_timer(None)
        profile = prof.summary()
        dom_name = max(profile, key=lambda n: profile[n]['total_ms'])
        timer = ops.KernelTimer(only={dom_name, 'embed_fwd', 'emb_sparse_apply'}, pool=args.steps * 8)
        ops.set_timer(timer)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        model.train_step(*batches[i % len(batches)])
    barrier()
    dt = time.perf_counter() - t0
    ops.set_timer(None)
    loss = float(model._loss_dev.item())
    model.check_overflow()      # row-sharded tables: a fixed-capacity send bucket that overflowed is an error
    if world > 1:
        t = torch.tensor([dt], device='cpu' if args.comm == 'host-staged' else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    value = world * B * args.steps / dt
    out = dict(metric='train samples/sec (node) DeepFM+DCN Criteo-shape; % HBM roofline on emb kernels',
               value=value, unit='samples/s', n_gpus=world, steps=args.steps, warmup=args.warmup,
               ms_per_step=1e3 * dt / args.steps, higher_is_better=True, scaling='weak', vs_baseline=None,
               dtype='f32', data='synthetic',
               gemm_arithmetic='fp32 result from 6 bf16 MFMAs per product (exact 3-way split, fp32 accumulate; '
                               '~1e-7 relative)',
               config=dict(workload='%s %s shape: 13 dense + 26 sparse fields, sum(V)=%d rows, '
                                    'embed dim %d, per-GPU batch %d, Zipf(%.2f) ids, fwd+bwd+Adam (sparse rows: '
                                    'row-wise Adam on touched rows), dropout at the reference rates'
                                    % (args.model, shape_name, int(np.sum(cards)), args.embed, B, args.alpha),
                           global_batch=world * B, parallelism='dp%d + %s-sharded tables' % (world, args.shard) if world > 1 else 'single'),
               final_loss=loss if np.isfinite(loss) else None)
    if timer is not None:
        ks = timer.summary()
        F, E = len(cards), args.embed
        traffic = load_traffic()
        from recommendation_amd import _lib
        gemm_mode = int(_lib.load().ctr_get_gemm_mode())

        def hbm_roof(name, kernel, nbytes, avg_us, sym):
            ach = nbytes / (avg_us * 1e-6) / 1e9
            return dict(kernel=kernel, bound='hbm', achieved=ach, peak=HBM_PEAK_GBS, unit='GB/s',
                        frac=ach / HBM_PEAK_GBS, traffic=traffic.get(sym), avg_us=avg_us, bytes_per_launch=nbytes)

        def mfma_roof(name, M, N, K, avg_us):
            # fp32-equivalent flops against the dense fp32 MFMA peak (the path's dtype).  In split mode the
            # kernel issues six bf16 MFMAs per fp32 product, so what it occupies of the matrix cores is
            # 6 * achieved / bf16 peak, reported beside it.
            ach = 2.0 * M * N * K / (avg_us * 1e-6) / 1e12
            split = gemm_mode == 1
            out = dict(kernel=name + (' (fp32 via 3-way bf16 split, 6 bf16 MFMAs per product)' if split else
                                      ' (fp32 MFMA)'),
                       bound='mfma', achieved=ach, peak=MFMA_F32_PEAK_TFLOPS, unit='TFLOP/s',
                       frac=ach / MFMA_F32_PEAK_TFLOPS, traffic=None, avg_us=avg_us)
            if split:
                out['bf16_pipe_frac'] = 6.0 * ach / MFMA_BF16_PEAK_TFLOPS
            return out

        def roof_of(name):
            k = ks[name]
            if name == 'embed_fwd':
                # algorithmic bytes per sample: ids + row read + row write (embedding gather forward)
                return hbm_roof(name, 'embed_fwd_kernel (gather + FM first/second order + concat)',
                                F * (4 + E * 4 + E * 4) * B, k['avg_us'], 'embed_fwd_kernel')
            if name == 'emb_sparse_apply':
                # per lookup: key + src + gradient row; per unique row: read+write of W, m, v
                return hbm_roof(name, 'sparse_apply_kernel (+tail): segment-reduce + row-wise Adam',
                                int(F * B * (8 + E * 4) + uniq * 6 * E * 4), k['avg_us'], 'sparse_apply_kernel')
            if name.startswith('gemm_') or name.startswith('wgrad_'):
                M, N, K = [int(v) for v in name.split('_')[1].split('x')]
                return mfma_roof(name, M, N, K, k['avg_us'])
            return dict(kernel=name, bound='hbm', achieved=None, peak=HBM_PEAK_GBS, unit='GB/s', frac=None,
                        traffic=None, avg_us=k['avg_us'])

        out['roofline'] = roof_of(dom_name)
        out['roofline_emb'] = roof_of('embed_fwd') if 'embed_fwd' in ks else None
        out['roofline_emb_bwd'] = roof_of('emb_sparse_apply') if 'emb_sparse_apply' in ks else None
        if out['roofline_emb_bwd'] and 'emb_sparse_apply' in profile:
            # in the timed region the sparse update shares the GPU with the tower weight gradients (side
            # stream); its launch time alone on the GPU comes from the single-stream profiling pass
            r = out['roofline_emb_bwd']
            alone_us = 1e3 * profile['emb_sparse_apply']['total_ms'] / max(profile['emb_sparse_apply']['calls'], 1)
            r['shares_gpu_with'] = 'tower weight gradients (side stream)'
            r['alone_avg_us'] = alone_us
            r['alone_frac'] = r['bytes_per_launch'] / (alone_us * 1e-6) / 1e9 / HBM_PEAK_GBS
        out['unique_rows_per_step'] = uniq
        out['profile_steps'] = PROFILE_STEPS
        out['kernel_ms_per_step'] = {n: round(v['total_ms'] / PROFILE_STEPS, 4) for n, v in
                                     sorted(profile.items(), key=lambda kv: -kv[1]['total_ms'])}
    if not args.no_cpu_baseline and world == 1:
        out['cpu_baseline'] = cpu_baseline(args, cards)
    elif world == 1:
        out['cpu_baseline'] = None
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

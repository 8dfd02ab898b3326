#!/usr/bin/env python
"""Headline benchmark: DeepFM training step on Criteo-Kaggle-shape synthetic data.  At N = 1 the same line also
carries, under `legs`, the other workloads of the measurement plan, each with its own `config.workload`: the DCN step
at the Criteo-1TB per-GPU share, DeepFM with uniform ids and at batch 4096 / 16384, PNN (inner, outer) and AFM
(reference attention, attention over the pairs) over 39 fields at batch 65536, DIN at history length 200 (H = 32, 128)
and ESMM.

  python bench.py --gpus 1 --steps K --warmup W
  python bench.py --gpus N ...          (starts N ranks under torch.distributed.run itself when WORLD_SIZE is unset)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (embedding gather + FM + tower forward, loss, backward,
sorted segment-reduce + sparse Adam, dense Adam) over one batch that already sits in HBM.  After W warm-up steps the
block of K steps is timed REPEATS times (barrier + device synchronisation on both sides of every block); `value` comes
from the median block, every block's time is listed under `timing`.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import subprocess
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PROFILE_STEPS = 5
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec
HBM_COPY_GBS = 6290.0          # what a streaming copy reaches on this part (the achievable ceiling)
MFMA_F32_PEAK_TFLOPS = 157.3   # fp32-input MFMA (exact fp32) dense peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak; the split GEMM spends 6 bf16 MFMAs per fp32 product


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=0, help='per-GPU batch (default: 65536 DeepFM, 8192 DCN)')
    ap.add_argument('--no-bf16-leg', action='store_true', help='skip the extra timing of the reduced-precision GEMM mode')
    ap.add_argument('--embed', type=int, default=0, help='embedding dim (default: 16 DeepFM, 64 DCN)')
    ap.add_argument('--alpha', type=float, default=1.05, help='Zipf exponent of the ids (0 = uniform)')
    ap.add_argument('--model', default='DeepFM', choices=sorted(MODEL_ARGS),
                    help='DeepFM is the headline workload; the others time the same step for their family')
    ap.add_argument('--comm', default='rccl', choices=['rccl', 'rccl-abi', 'host-staged'],
                    help='rccl: torch.distributed (nccl backend); rccl-abi: the library\'s own ctr_comm_* entry points; '
                         'host-staged: gloo through host memory, to rehearse N>1 with all ranks on one GPU')
    ap.add_argument('--shard', default=None, choices=['table', 'row'],
                    help='N>1: whole tables per rank (default for the Criteo-Kaggle models), or row r of field f on rank '
                         '(r + f) %% N (default for DCN: its Criteo-1TB tables do not fit one owner)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timers', action='store_true')
    ap.add_argument('--no-dcn', action='store_true', help='N = 1, DeepFM: skip the DCN leg reported under "dcn"')
    ap.add_argument('--no-legs', action='store_true', help='N = 1, DeepFM: skip the other workloads reported under "legs"')
    ap.add_argument('--repeats', type=int, default=10, help='how many times the block of --steps steps is timed')
    ap.add_argument('--master-port', type=int, default=29511, help='rendezvous port when this script starts the ranks')
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
    # the headline model with the table update exactly as tf.train.AdamOptimizer applies an IndexedSlices gradient
    # (DeepFM/DeepFM.py:32,100): m and v of EVERY row decay and every row moves, each step
    'DeepFM_tf1adam': dict(deep_layers=[256, 128, 64, 32], dropout_fm=0.5, dropout_deep=0.2, adam_mode='tf1',
                           table_grads=False),
    'DCN_tf1adam': dict(deep_layers=[1024, 1024, 1024], cross_layer_num=6, dropout=0.2, adam_mode='tf1', table_grads=False),
}


def build_model(variant, embed, shard, comm, sess, cards, rank, world):
    import importlib
    import recommendation_amd as ra
    ra.enable_zoo_imports()
    name = variant.split('_')[0]
    margs = SimpleNamespace(**dict(dict(embed_size=embed, regular_rate=0.1, learning_rate=0.001, decay_steps=10000,
                                        decay_rate=0.99, mode='S', adam_mode='lazy', init_std=0.01), **MODEL_ARGS[variant]))
    cls = getattr(importlib.import_module('%s.%s' % (name, name)), name)
    if world > 1:
        margs.world_size, margs.rank, margs.shard = world, rank, shard
        if comm == 'rccl-abi':
            from recommendation_amd.distributed import CtrComm
            margs.comm = CtrComm()
        if comm == 'host-staged':
            from recommendation_amd.distributed import HostStagedComm
            margs.comm = HostStagedComm()
    m = cls(margs, len(cards), 13, cards)
    m.build()
    m.initialize(sess, seed=1234)
    if name == 'AFM':
        # AFM feeds log_loss an un-squashed y_out (AFM/AFM.py:85-88): with weights of both signs y_out goes negative
        # and the loss is NaN.  Positive weights (and positive dense inputs, below) keep 0 < y_out < 1, so the step
        # that is timed trains on finite numbers.
        m.tables.weight.abs_()
        m.dense.flat.abs_()
    return m


def cpu_baseline(args, cards):
    """The oracle (CPU restatement of the reference graph; TensorFlow is unavailable) timed on the host's cores on
    bounded samples: (c1) the reference's own CPU case exactly as its graph executes it - reference-exact embedding
    block with the dense [B, sum(V), E] tensors, B = 1024, 26 x 101 + 13 fields, E = 8, TF-1.x Adam; and the headline
    workload's shape scaled to the host (standard lookup, tables 1/64, B = 4096, row-wise Adam as on the GPU)."""
    from oracle import ctr_oracle as O
    from recommendation_amd.synthetic import scaled
    threads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)

    def timed(m, batch, budget_s, max_steps):
        m.run(*batch, True)
        t0 = time.perf_counter()
        n = 0
        while n < 3 or (time.perf_counter() - t0 < budget_s and n < max_steps):
            m.run(*batch, True)
            n += 1
        return n, time.perf_counter() - t0

    rng = np.random.default_rng(0)
    # c1: DeepFM/main.py defaults on 26 categorical fields of 101 values + 13 continuous
    c1 = O.default_cfg('DeepFM', embed_size=8, mode='R')
    c1.cate_num, c1.cont_num, c1.cate_list = 26, 13, [101] * 26
    m1 = O.OracleModel('DeepFM', c1, O.init_params('DeepFM', c1, seed=0), dtype=torch.float32)
    b1 = (rng.integers(0, 101, (1024, 26)), rng.standard_normal((1024, 13)).astype(np.float32), rng.integers(0, 2, 1024))
    n1, dt1 = timed(m1, b1, 8.0, 40)
    del m1
    div, B = 64, 4096
    small = scaled(cards, div)
    cfg = O.default_cfg('DeepFM', embed_size=args.embed, mode='S')
    cfg.cate_num, cfg.cont_num, cfg.cate_list, cfg.adam_mode = len(small), 13, small, 'lazy'
    m = O.OracleModel('DeepFM', cfg, O.init_params('DeepFM', cfg, seed=0), dtype=torch.float32)
    b2 = (np.stack([rng.integers(0, v, B) for v in small], 1), rng.standard_normal((B, 13)).astype(np.float32),
          rng.integers(0, 2, B))
    n2, dt2 = timed(m, b2, 10.0, 50)
    return dict(value=B * n2 / dt2, unit='samples/s', cores=threads, kind='port',
                sample='CPU restatement of the reference graph (TensorFlow unavailable): DeepFM standard lookup, '
                       'torch-CPU fp32, B=%d, %d steps, table rows scaled 1/%d (sum V=%d), row-wise Adam'
                       % (B, n2, div, int(np.sum(small))),
                c1=dict(value=1024 * n1 / dt1, unit='samples/s', cores=threads, kind='port',
                        sample='BASELINE configs[0] as the reference graph executes it: reference-exact embedding block '
                               '(dense [1024, 2626, 8] tensors), 26 x 101 + 13 fields, E=8, default tower, TF-1.x Adam, '
                               'torch-CPU fp32, %d steps' % n1))


def load_traffic(model):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this model's bench run (FETCH_SIZE / WRITE_SIZE
    collected in separate runs, gfx950 x2 read correction; tools/pmc_summary.py), keyed by kernel symbol; the newest
    profiles/*_<model>_*pmc_traffic.json wins.  Empty when no pass of this model is committed."""
    tag = {'DeepFM': 'deepfm', 'DCN': 'dcn'}.get(model)
    out = {}
    if tag is None:
        return out
    paths = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_%s_*pmc_traffic.json' % tag)))
    if paths:
        # operator -> the kernel symbols one call of it launches (every operator has a general and an aligned /
        # short-instruction-stream form; the cross backward adds the two kernels of the dw/db reduction, the sparse
        # update its tail)
        parts = {'embed_fwd_kernel': ('embed_fwd_kernel', 'embed_fwd_wide_kernel', 'embed_fwd_lean_kernel'),
                 'sparse_apply_kernel': ('sparse_apply_kernel', 'sparse_apply_lean_kernel', 'sparse_apply_tail_kernel'),
                 'embed_bwd_dz_kernel': ('embed_bwd_dz_kernel',),
                 'cross_fwd_kernel': ('cross_fwd_kernel', 'cross_fwd_lds_kernel', 'cross_fwd_r16_kernel'),
                 'cross_bwd_kernel': ('cross_bwd_kernel', 'cross_bwd_lds_kernel', 'cross_bwd_r16_kernel',
                                      'cross_bwd_partial_kernel', 'cross_bwd_partial_fast_kernel',
                                      'cross_bwd_final_kernel', 'cross_bwd_final_fast_kernel')}
        for k, v in json.load(open(paths[-1])).items():
            for op, syms in parts.items():
                if any(sym + '<' in k or sym + '(' in k for sym in syms):
                    out[op] = out.get(op, 0) + v['hbm_bytes_per_launch']
        out['_file'] = os.path.basename(paths[-1])
    return out


def time_blocks(run_step, steps, repeats, barrier):
    """Seconds of `repeats` blocks of `steps` steps, each between barrier()s (which synchronise the device)."""
    out = []
    for _ in range(max(repeats, 1)):
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            run_step(i)
        barrier()
        out.append(time.perf_counter() - t0)
    return out


def timing_of(blocks, steps):
    ms = sorted(1e3 * t / steps for t in blocks)
    return dict(blocks=len(ms), ms_per_step_median=float(np.median(ms)), ms_per_step_min=ms[0], ms_per_step_max=ms[-1],
                ms_per_step_all=[round(1e3 * t / steps, 4) for t in blocks])


# flops per SAMPLE of the interaction operators that are not plain products (N rows of Z, E floats each, P product
# units, A attention units, T history positions, H hidden width), for their fraction of the fp32 MFMA peak
def interaction_flops(name, N, E, P=50, A=10, T=200, H=32):
    pairs = N * (N - 1) // 2
    table = {'pnn_inner_fwd': 2 * P * N * E, 'pnn_inner_bwd': 4 * P * N * E,
             'afm_pair_fwd': 2 * N * N * E, 'afm_pair_bwd': 2 * N * N * E,
             'afm_attn_fwd': pairs * (E + 2 * E * A + 2 * A + 2 * E), 'afm_attn_bwd': 3 * pairs * (E + 2 * E * A + 2 * A + 2 * E),
             'din_attn_mlp_fwd': T * 2 * (4 * H * 80 + 80 * 40 + 40), 'din_attn_mlp_bwd': 2 * T * 2 * (4 * H * 80 + 80 * 40 + 40),
             'din_attn_mlp_wgrad': T * 2 * (4 * H * 80 + 80 * 40 + 40)}
    return table.get(name)


def run_model(variant, args, sess, rank, world, dist, steps, warmup, batch, embed, shard, alpha=None, light=False):
    """Builds the model, warms up, profiles per operator (untimed), then times `steps` steps args.repeats times.
    Returns the result dict (rank 0) or None.  light: a leg of the measurement plan - no reduced-precision timing, no
    cold-copy yardsticks."""
    alpha = args.alpha if alpha is None else alpha
    from recommendation_amd import ops
    from recommendation_amd.synthetic import CRITEO_1TB, CRITEO_KAGGLE, make_batches, scaled
    dev = sess.device
    is_dcn = variant.startswith('DCN')
    # DeepFM: BASELINE configs[1] (Criteo-Kaggle cardinalities).  DCN: configs[2] per-GPU share —
    # Criteo-1TB cardinalities / 8 (one GPU of the 8-GPU shard), dim 64, 6 cross layers, 1024x3 tower.
    cards = CRITEO_KAGGLE if not is_dcn else scaled(CRITEO_1TB, 8 // min(world, 8))
    shape_name = 'Criteo-Kaggle' if not is_dcn else 'Criteo-1TB/%d' % (8 // min(world, 8))
    if rank == 0:
        print('[bench] building %s: %d table rows on %d rank(s)' % (variant, sum(cards), world), file=sys.stderr, flush=True)
    model = build_model(variant, embed, shard, args.comm, sess, cards, rank, world)
    B = batch
    batches = make_batches(cards, 13, B, 8, alpha, 20240601 + rank, dev)
    if variant.startswith('AFM'):
        batches = [(i, x.abs_().mul_(0.1), y) for i, x, y in batches]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # distinct table rows touched per step (for the algorithmic bytes of the sparse update)
    offs = torch.as_tensor(np.concatenate([[0], np.cumsum(cards)[:-1]]), device=dev)
    uniq = float(np.mean([torch.unique(b[0].to(torch.int64) + offs).numel() for b in batches]))
    if rank == 0:
        print('[bench] %s warm-up' % variant, file=sys.stderr, flush=True)
    for i in range(warmup):
        model.train_step(*batches[i % len(batches)])
    # Untimed profiling pass: every operator between HIP events, to split the step by kernel and to
    # find the dominant one.  The timed region then keeps events only around that kernel and the
    # embedding / interaction kernels (a pair of events per operator costs ~10 % of a 1 ms step).
    profile, timer, dom_name = None, None, None
    watched = {'embed_fwd', 'emb_sparse_apply', 'cross_fwd', 'cross_bwd'}
    if not args.no_kernel_timers:
        prof = ops.KernelTimer(pool=PROFILE_STEPS * 200)
        ops.set_timer(prof)
        sess.side_enabled = False      # one stream: every operator's time is its own, not a shared GPU's
        for i in range(PROFILE_STEPS):
            model.train_step(*batches[i % len(batches)])
        sess.side_enabled = True
        ops.set_timer(None)
        profile = prof.summary()
        dom_name = max(profile, key=lambda n: profile[n]['total_ms'])
        timer = ops.KernelTimer(only=watched | {dom_name}, pool=steps * 12 * max(args.repeats, 1))
        ops.set_timer(timer)
    blocks = time_blocks(lambda i: model.train_step(*batches[i % len(batches)]), steps, args.repeats, barrier)
    ops.set_timer(None)
    eager_blocks = None
    from recommendation_amd.model_base import GRAPH_MAX_BATCH
    if world == 1 and model.graph_pays(B):
        # small batches are launch-bound: run() replays the step from a captured graph there, and that is the figure
        # of this workload (the eager time is kept beside it)
        for i in range(2):
            model.train_step_graphed(*batches[0])          # eager step + capture, one replay
        eager_blocks = blocks
        blocks = time_blocks(lambda i: model.train_step_graphed(*batches[i % len(batches)]), steps, args.repeats, barrier)
    loss = float(model._loss_dev.item())
    model.check_overflow()      # row-sharded tables: a fixed-capacity send bucket that overflowed is an error
    # the same steps once more with the tile GEMM in its opt-in reduced-precision mode (bf16 operands, one MFMA per
    # product): reported beside the headline, never as it
    bf16 = None
    if world == 1 and not args.no_bf16_leg and not light:
        prev = ops.set_gemm_mode('bf16')
        try:
            for i in range(3):
                model.train_step(*batches[i % len(batches)])
            barrier()
            tb = time.perf_counter()
            for i in range(steps):
                model.train_step(*batches[i % len(batches)])
            barrier()
            dtb = time.perf_counter() - tb
            bf16 = dict(value=B * steps / dtb, unit='samples/s', ms_per_step=1e3 * dtb / steps,
                        final_loss=float(model._loss_dev.item()),
                        note='tile GEMMs with bf16 operands and fp32 accumulate (ctr_set_gemm_mode(CTR_GEMM_BF16)), '
                             'continuing from the fp32-accurate steps above; not the headline figure')
        finally:
            ops.set_gemm_mode(prev)
    if world > 1:                       # every block: the slowest rank's time
        t = torch.tensor(blocks, device='cpu' if args.comm == 'host-staged' else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        blocks = [float(v) for v in t.tolist()]
    if rank != 0:
        return None
    dt = float(np.median(blocks))
    out = dict(value=world * B * steps / dt, unit='samples/s', steps=steps, warmup=warmup, ms_per_step=1e3 * dt / steps,
               timing=timing_of(blocks, steps),
               config=dict(workload='%s %s shape: 13 dense + 26 sparse fields, sum(V)=%d rows, '
                                    'embed dim %d, per-GPU batch %d, Zipf(%.2f) ids (0 = uniform), fwd+bwd+%s (sparse rows: '
                                    '%s), dropout at the reference rates'
                                    % (variant, shape_name, int(np.sum(cards)), embed, B, alpha, model.OPTIMIZER,
                                       'every table row updated each step, as tf.train.AdamOptimizer does'
                                       if getattr(model, 'adam_mode', '') == 'tf1' else 'row-wise update of touched rows'),
                           global_batch=world * B,
                           parallelism='dp%d + %s-sharded tables' % (world, shard) if world > 1 else 'single'),
               final_loss=loss if np.isfinite(loss) else None)
    tb = getattr(getattr(model, 'tables', None), 'touched', None)
    if tb is not None:
        # the decay pass of tf.train.AdamOptimizer skips rows that never received a gradient (m = v = 0: the step
        # leaves them as they are), so its cost follows the share of rows with optimizer state: the share reached on
        # the synthetic batches (a cycle of 8) above, and the other end - every row given state - timed here
        share, at = float((tb != 0).sum().item()) / tb.numel(), int(model.global_step)
        model.tables.slot1.fill_(1e-30)
        tb.fill_(2)
        for i in range(2):
            model.train_step(*batches[i % len(batches)])
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            model.train_step(*batches[i % len(batches)])
        barrier()
        dta = time.perf_counter() - t0
        out['rows_with_optimizer_state'] = dict(
            share=share, after_steps=at,
            note='rows without state are exact no-ops of the dense Adam step and are not read',
            every_row_with_state=dict(value=world * B * steps / dta, ms_per_step=1e3 * dta / steps,
                                      note='v of every row set to 1e-30 first: the same steps with all rows decaying'))
    if eager_blocks is not None:
        out['step_runs_as'] = 'captured graph replay (launch-bound step: batch <= %d and narrow towers, as run() does)' % GRAPH_MAX_BATCH
        out['eager'] = dict(value=B * steps / float(np.median(eager_blocks)), timing=timing_of(eager_blocks, steps))
    if bf16 is not None:
        out['bf16_tower_mode'] = bf16
    if timer is None:
        return out
    ks = timer.summary()
    F, E = len(cards), embed
    D = F * E + 13
    traffic = load_traffic(variant)
    from recommendation_amd import _lib
    gemm_mode = int(_lib.load().ctr_get_gemm_mode())
    # algorithmic HBM bytes per launch (per sample: ids + row read + row write for the gather; key + src + gradient
    # row per lookup and read + write of W, m, v per unique row for the update; x0 in / x_L out for the cross forward,
    # x0, g in and dx0 read-modify-write for its backward; the pair sort: 8-byte pairs written once, then read and
    # written once per radix pass)
    algo = {'embed_fwd': F * (4 + E * 4 + E * 4) * B,
            'emb_sparse_apply': int(F * B * (8 + E * 4) + uniq * 6 * E * 4),
            'embed_bwd_dz': 3 * F * E * 4 * B,
            'emb_sort': F * B * (4 + 8 + 3 * 16),
            'cross_fwd': 2 * D * 4 * B, 'cross_bwd': 4 * D * 4 * B}
    sym = {'embed_fwd': 'embed_fwd_kernel', 'emb_sparse_apply': 'sparse_apply_kernel', 'cross_fwd': 'cross_fwd_kernel',
           'cross_bwd': 'cross_bwd_kernel', 'embed_bwd_dz': 'embed_bwd_dz_kernel'}
    label = {'embed_fwd': 'embed_fwd_kernel (gather + FM first/second order + concat)',
             'emb_sparse_apply': 'sparse_apply_kernel (+tail): segment-reduce + row-wise Adam',
             'cross_fwd': 'cross_fwd_kernel (all cross layers, one pass)',
             'cross_bwd': 'cross_bwd_kernel (+ the fixed-order dw/db reduction)'}

    def hbm_roof(name, avg_us):
        nbytes = algo[name]
        ach = nbytes / (avg_us * 1e-6) / 1e9
        r = dict(kernel=label.get(name, name), bound='hbm', achieved=ach, peak=HBM_PEAK_GBS, unit='GB/s',
                 frac=ach / HBM_PEAK_GBS, frac_of_measured_copy_peak=ach / HBM_COPY_GBS,
                 traffic=traffic.get(sym.get(name)), avg_us=avg_us, bytes_per_launch=nbytes)
        if r['traffic'] is not None:     # not measured in this run: read from the committed counter pass
            r['traffic_source'] = 'committed profile profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)' % traffic['_file']
        return r

    def mfma_roof(name, M, N, K, avg_us):
        # fp32-equivalent flops against the dense fp32 MFMA peak (the path's dtype).  In split mode the
        # kernel issues six bf16 MFMAs per fp32 product, so what it occupies of the matrix cores is
        # 6 * achieved / bf16 peak, reported beside it.
        ach = 2.0 * M * N * K / (avg_us * 1e-6) / 1e12
        split = gemm_mode == 1
        f16 = split and getattr(getattr(model, 'tower', None), 'am', None) is not None
        per = 3.0 if f16 else 6.0
        r = dict(kernel=name + (' (fp32 via 2-term fp16 split, 3 fp16 MFMAs per product)' if f16 else
                                ' (fp32 via 3-way bf16 split, 6 bf16 MFMAs per product)' if split else ' (fp32 MFMA)'),
                 bound='mfma', achieved=ach, peak=MFMA_F32_PEAK_TFLOPS, unit='TFLOP/s',
                 frac=ach / MFMA_F32_PEAK_TFLOPS, traffic=None, avg_us=avg_us)
        if split:
            r['matrix_pipe_frac'] = per * ach / MFMA_BF16_PEAK_TFLOPS    # of the 16-bit dense MFMA peak
        return r

    Nrows = F + (13 if getattr(model, 'CONT_EMBED', False) else 0)

    def roof_of(name):
        k = ks[name]
        if name in algo:
            return hbm_roof(name, k['avg_us'])
        if name.startswith('gemm_') or name.startswith('wgrad_'):
            M, N, K = [int(v) for v in name.split('_')[1].split('x')]
            return mfma_roof(name, M, N, K, k['avg_us'])
        fl = interaction_flops(name, Nrows, E, P=int(MODEL_ARGS[variant].get('product_size', 50)),
                               A=int(MODEL_ARGS[variant].get('attention_size', 10)))
        if fl:          # interaction layers formed on chip: compute-bound, against the fp32 MFMA peak
            ach = fl * B / (k['avg_us'] * 1e-6) / 1e12
            return dict(kernel=name, bound='mfma', achieved=ach, peak=MFMA_F32_PEAK_TFLOPS, unit='TFLOP/s',
                        frac=ach / MFMA_F32_PEAK_TFLOPS, traffic=None, avg_us=k['avg_us'], flops_per_sample=fl)
        return dict(kernel=name, bound='hbm', achieved=None, peak=HBM_PEAK_GBS, unit='GB/s', frac=None,
                    traffic=None, avg_us=k['avg_us'])

    out['roofline'] = roof_of(dom_name)
    if variant == 'DeepFM' and world == 1 and not light and getattr(model, 'fuse_dz', False) and dom_name.startswith('gemm_%dx%dx' % (B, D)):
        # the dominant launch is the dx0 product WITH dZ's FM terms in its epilogue (the 65 us dZ pass it replaced has
        # no launch any more): its duration covers both, its flops only the product.  The same product without the
        # epilogue, on the same buffers, timed here for the matrix-core fraction of the product by itself.
        bufs = model._buffers(B)
        g0, w0 = bufs['gbufs'][0], model.dense.p('deep_part/deep_w_0')
        scratch = torch.empty_like(bufs['dx0'])
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(42)]
        t = model.tower
        kw = dict(a_amax=t._slot('g', 0), b_amax=t._slot('w', 0)) if t.am is not None else {}   # the step's arithmetic
        for i in range(21):
            evs[2 * i].record()
            ops.gemm(g0, w0, scratch[:, :D], trans_b=True, ws=sess.ws, **kw)
            evs[2 * i + 1].record()
        torch.cuda.synchronize()
        plain_us = 1e3 * float(np.mean([evs[2 * i].elapsed_time(evs[2 * i + 1]) for i in range(1, 21)]))
        r = out['roofline']
        r['kernel'] += ' + dZ epilogue (FM terms of the embedding backward: 124 MB of operand reads, no launch of its own)'
        r['product_alone'] = dict(avg_us=plain_us, achieved=2.0 * B * D * w0.shape[1] / (plain_us * 1e-6) / 1e12,
                                  frac=2.0 * B * D * w0.shape[1] / (plain_us * 1e-6) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                                  note='same product, same buffers, epilogue off, timed after the run')
        del scratch
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
    for name in ('cross_fwd', 'cross_bwd'):
        if name in ks:
            r = out['roofline_' + name] = roof_of(name)
            if name == 'cross_bwd' and name in profile:
                # like the sparse update, the cross backward runs beside the tower weight gradients of the side stream
                alone_us = 1e3 * profile[name]['total_ms'] / max(profile[name]['calls'], 1)
                r['shares_gpu_with'] = 'tower weight gradients (side stream)'
                r['alone_avg_us'] = alone_us
                r['alone_frac'] = r['bytes_per_launch'] / (alone_us * 1e-6) / 1e9 / HBM_PEAK_GBS
    # the whole embedding path of a step (gather, dZ, pair sort, segment-reduce + row update), single-stream times
    # of the profiling pass: algorithmic bytes of the pieces over their summed time
    path = [n for n in ('embed_fwd', 'embed_bwd_dz', 'emb_sort', 'emb_sparse_apply') if n in profile]
    if path:
        us = {n: 1e3 * profile[n]['total_ms'] / PROFILE_STEPS for n in path}
        tot_b, tot_us = sum(algo[n] for n in path), sum(us.values())
        ach = tot_b / (tot_us * 1e-6) / 1e9
        # the same bytes as cold streaming copies, one per kernel: the time a launch of that size needs at the least
        # when it starts with the caches full of the rest of the step
        out['roofline_emb_path'] = dict(kernels=path, bound='hbm', bytes_per_step=tot_b, us_per_step=tot_us, achieved=ach,
                                        peak=HBM_PEAK_GBS, unit='GB/s', frac=ach / HBM_PEAK_GBS,
                                        frac_of_measured_copy_peak=ach / HBM_COPY_GBS,
                                        us={n: round(v, 2) for n, v in us.items()})
        if not light:
            cold = {n: cold_copy_us(algo[n], dev) for n in path}
            out['roofline_emb_path'].update(cold_copy_us={n: round(v, 2) for n, v in cold.items()},
                                            frac_of_cold_copy=sum(cold.values()) / tot_us)
        if 'embed_bwd_dz' not in path and variant == 'DeepFM':
            out['roofline_emb_path']['note'] = ('dZ (FM terms added to the tower gradient) has no launch of its own: it rides '
                                                'the epilogue of the dx0 product (gemm_%dx%dx%d_NT)' % (B, D, 256))
    out['pmc_traffic_file'] = traffic.get('_file')
    out['unique_rows_per_step'] = uniq
    out['profile_steps'] = PROFILE_STEPS
    out['kernel_ms_per_step'] = {n: round(v['total_ms'] / PROFILE_STEPS, 4) for n, v in
                                 sorted(profile.items(), key=lambda kv: -kv[1]['total_ms'])}
    return out


def cold_copy_us(nbytes, dev, reps=5):
    """Time of a streaming device copy that moves `nbytes` in total (half read, half written) when it starts the way
    a kernel inside a training step does: caches holding other data (a 1 GiB fill runs before every repetition).
    The yardstick for what a launch of this size can reach at all; mean of `reps` event-timed runs."""
    n = max(int(nbytes) // 8, 1)
    src = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    ts = []
    for r in range(reps + 1):
        flush.fill_(float(r))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        if r:
            ts.append(e0.elapsed_time(e1) * 1e3)
    del src, dst, flush
    return float(np.mean(ts))


def run_din(sess, args, H, steps, warmup, B=2048, T=200):
    """DIN training step at the Amazon-Books shape (367 982 items, 1 600 categories, history length T with lengths
    ~U[20, T], Zipf-distributed history ids) on one GPU: eager and replayed from a captured graph."""
    import recommendation_amd as ra
    from recommendation_amd import ops
    ra.enable_zoo_imports()
    from DIN import DIN
    items, cates, users = 367982, 1600, 603668
    rng = np.random.default_rng(0)
    cate_list = rng.integers(0, cates, items).astype(np.int32)
    margs = SimpleNamespace(hidden_units=H, train_batch_size=B, test_batch_size=B, learning_rate=0.1, epoch=1)
    m = DIN.DIN(margs, users, items, cates, cate_list)
    m.initialize(sess, seed=1)
    dev = sess.device
    batches = []
    for _ in range(4):
        sl = rng.integers(20, T + 1, B)
        hist = rng.zipf(1.2, (B, T)) % items
        hist[np.arange(T)[None, :] >= sl[:, None]] = 0
        batches.append((torch.as_tensor(rng.integers(0, items, B).astype(np.int32)).to(dev),
                        torch.as_tensor((rng.random(B) < 0.5).astype(np.float32)).to(dev),
                        torch.as_tensor(hist.astype(np.int32)).to(dev), torch.as_tensor(sl.astype(np.int32)).to(dev)))
    sync = lambda: torch.cuda.synchronize()
    for k in range(max(warmup, 2)):
        m.train_step(*batches[k % 4], 0.1)
    prof = ops.KernelTimer(pool=PROFILE_STEPS * 200)
    ops.set_timer(prof)
    for k in range(PROFILE_STEPS):
        m.train_step(*batches[k % 4], 0.1)
    ops.set_timer(None)
    profile = prof.summary()
    blocks = time_blocks(lambda i: m.train_step(*batches[i % 4], 0.1), steps, args.repeats, sync)
    for k in range(2):
        m.train_step_graphed(*batches[0], 0.1)          # eager step + capture, one replay
    gblocks = time_blocks(lambda i: m.train_step_graphed(*batches[0], 0.1), steps, args.repeats, sync)
    dt, gdt = float(np.median(blocks)), float(np.median(gblocks))
    dom = max(profile, key=lambda n: profile[n]['total_ms'])
    out = dict(value=B * steps / dt, unit='samples/s', ms_per_step=1e3 * dt / steps, timing=timing_of(blocks, steps),
               graph_replay=dict(value=B * steps / gdt, ms_per_step=1e3 * gdt / steps,
                                 note='the same step replayed from a captured graph (one batch; bit-identical arithmetic)'),
               config=dict(workload='DIN Amazon-Books shape: %d items, %d categories, B=%d, T=%d (lengths U[20,%d]), '
                                    'hidden_units=%d, fwd+bwd+SGD, %s attention scoring network'
                                    % (items, cates, B, T, T, H, 'fused' if m.fused_attention else 'layer-by-layer')),
               final_loss=float(m._loss_dev.item()),
               kernel_ms_per_step={n: round(v['total_ms'] / PROFILE_STEPS, 4) for n, v in
                                   sorted(profile.items(), key=lambda kv: -kv[1]['total_ms'])[:12]})
    us = 1e3 * profile[dom]['total_ms'] / max(profile[dom]['calls'], 1)
    fl = interaction_flops(dom, 0, 0, T=T, H=H)
    if fl:
        ach = fl * B / (us * 1e-6) / 1e12
        out['roofline'] = dict(kernel=dom, bound='mfma', achieved=ach, peak=MFMA_F32_PEAK_TFLOPS, unit='TFLOP/s',
                               frac=ach / MFMA_F32_PEAK_TFLOPS, traffic=None, avg_us=us, flops_per_sample=fl,
                               flops_note='flops of the reference graph (first layer 4H wide, DIN/DIN.py:169-185); the kernel '
                                          'multiplies 2H columns per row plus H per sample (q(W1a+W1c) + k(W1b-W1c) + (q*k)W1d)')
    else:
        out['roofline'] = dict(kernel=dom, bound='hbm', achieved=None, peak=HBM_PEAK_GBS, unit='GB/s', frac=None,
                               traffic=None, avg_us=us)
    m.close()
    return out


def spawn_ranks(args):
    """--gpus N without a launcher: start the N ranks as children under torch.distributed.run (a fresh process tree,
    started before this process touches the GPU) and relay rank 0's JSON line."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(args.master_port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith('{')]
    if lines:
        print(lines[-1])
    raise SystemExit(proc.returncode if proc.returncode or lines else 1)


def main():
    args = parse()
    if not args.batch:
        args.batch = 8192 if args.model == 'DCN' else 65536
    if not args.embed:
        args.embed = 64 if args.model == 'DCN' else 16
    if args.shard is None:
        args.shard = 'row' if args.model == 'DCN' else 'table'
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        spawn_ranks(args)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        raise SystemExit('--gpus %d does not match WORLD_SIZE %d' % (args.gpus, world))
    import recommendation_amd as ra

    if args.comm == 'host-staged':
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = 'cuda:%d' % local_rank
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.comm == 'host-staged':
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device(dev))
    sess = ra.Session(dev, seed=11 + rank)
    res = run_model(args.model, args, sess, rank, world, dist, args.steps, args.warmup, args.batch, args.embed, args.shard)
    import gc

    def leg(fn, *a, **k):
        """One more workload; a failure is reported in place of its numbers, never in place of the line."""
        gc.collect()
        torch.cuda.empty_cache()
        try:
            return fn(*a, **k)
        except Exception as e:                       # noqa: BLE001
            return dict(error='%s: %s' % (type(e).__name__, e))

    dcn, legs = None, {}
    full = world == 1 and args.model == 'DeepFM'
    if full and not args.no_dcn:
        # the metric names DeepFM and DCN: the DCN step at BASELINE configs[2]'s per-GPU share rides along (its own
        # tables, its own timed region after the headline's), reported under "dcn"
        dcn = leg(run_model, 'DCN', args, sess, rank, world, dist, max(args.steps, 20), max(args.warmup, 3), 8192, 64, 'row')
    if full and not args.no_legs:
        st, wu = max(min(args.steps, 20), 5), max(min(args.warmup, 3), 2)
        legs['deepfm_uniform_ids'] = leg(run_model, 'DeepFM', args, sess, rank, world, dist, st, wu, args.batch, args.embed,
                                         args.shard, alpha=0.0, light=True)
        for b in (4096, 16384):
            legs['deepfm_B%d' % b] = leg(run_model, 'DeepFM', args, sess, rank, world, dist, st, wu, b, args.embed,
                                         args.shard, light=True)
        legs['deepfm_tf1_adam'] = leg(run_model, 'DeepFM_tf1adam', args, sess, rank, world, dist, st, wu, args.batch,
                                      args.embed, args.shard, light=True)
        legs['dcn_tf1_adam'] = leg(run_model, 'DCN_tf1adam', args, sess, rank, world, dist, st, wu, 8192, 64, 'row', light=True)
        legs['deepfm_tf1_adam_uniform_ids'] = leg(run_model, 'DeepFM_tf1adam', args, sess, rank, world, dist, st, wu,
                                                  args.batch, args.embed, args.shard, alpha=0.0, light=True)
        for v in ('PNN', 'PNN_outer', 'AFM', 'AFM_pairs', 'ESMM', 'NFM'):      # BASELINE configs[3] (+ ESMM of configs[4])
            legs[v] = leg(run_model, v, args, sess, rank, world, dist, st, wu, 65536, 16, 'table', light=True)
        for h in (32, 128):                                                    # BASELINE configs[4]
            legs['DIN_H%d' % h] = leg(run_din, sess, args, h, st, wu)
        gc.collect()
        torch.cuda.empty_cache()
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    out = dict(metric='train samples/sec (node) DeepFM+DCN Criteo-shape; % HBM roofline on emb kernels',
               value=res['value'], unit='samples/s', n_gpus=world, steps=args.steps, warmup=args.warmup,
               ms_per_step=res['ms_per_step'], higher_is_better=True, scaling='weak', vs_baseline=None,
               dtype='f32', data='synthetic',
               gemm_arithmetic='fp32 result from 16-bit MFMAs: the tower products whose operands\' largest magnitudes '
                               'are tracked run as a two-term fp16 split (power-of-two scaling per matrix, 3 MFMAs per '
                               'product, ~3e-7 relative), the others as a three-term bf16 split (6 MFMAs, ~1e-7); fp32 '
                               'accumulate')
    out.update({k: v for k, v in res.items() if k not in ('value', 'unit', 'steps', 'warmup', 'ms_per_step')})
    if dcn is not None:
        out['dcn'] = dcn
    if legs:
        out['legs'] = legs
    if not args.no_cpu_baseline and world == 1:
        from recommendation_amd.synthetic import CRITEO_KAGGLE
        out['cpu_baseline'] = cpu_baseline(args, CRITEO_KAGGLE)
    elif world == 1:
        out['cpu_baseline'] = None
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

"""DIN training step at the Amazon-Books shape (item 367 982 / cate 1 600 / user 603 668, history
length T with lengths ~U[20, T]) on one GPU: samples/s and the per-operator time split.

  python tools/din_bench.py [--B 2048] [--T 200] [--H 32] [--steps 20]
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--B', type=int, default=2048)
    ap.add_argument('--T', type=int, default=200)
    ap.add_argument('--H', type=int, default=32)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--graph', type=int, default=0, help='1: replay the step from a captured graph')
    a = ap.parse_args()
    import recommendation_amd as ra
    from recommendation_amd import ops
    ra.enable_zoo_imports()
    from DIN import DIN
    sess = ra.Session('cuda:0', seed=5)
    items, cates, users = 367982, 1600, 603668
    rng = np.random.default_rng(0)
    cate_list = rng.integers(0, cates, items).astype(np.int32)
    args = SimpleNamespace(hidden_units=a.H, train_batch_size=a.B, test_batch_size=a.B, learning_rate=0.1, epoch=1)
    m = DIN.DIN(args, users, items, cates, cate_list)
    m.initialize(sess, seed=1)
    dev = sess.device
    batches = []
    for k in range(4):
        sl = rng.integers(20, a.T + 1, a.B)
        hist = rng.zipf(1.2, (a.B, a.T)) % items
        hist[np.arange(a.T)[None, :] >= sl[:, None]] = 0
        batches.append((torch.as_tensor(rng.integers(0, items, a.B).astype(np.int32)).to(dev),
                        torch.as_tensor((rng.random(a.B) < 0.5).astype(np.float32)).to(dev),
                        torch.as_tensor(hist.astype(np.int32)).to(dev), torch.as_tensor(sl.astype(np.int32)).to(dev)))
    step = m.train_step_graphed if a.graph else m.train_step
    for k in range(3):
        step(*batches[k % 4], 0.1)
    timer = ops.KernelTimer(pool=a.steps * 120)
    if not a.graph:                      # per-operator events cannot be recorded into a replayed graph
        ops.set_timer(timer)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(*batches[k % 4], 0.1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ops.set_timer(None)
    ks = timer.summary()
    print(json.dumps(dict(workload='DIN Amazon-Books shape, B=%d T=%d H=%d, fwd+bwd+SGD%s' % (
                              a.B, a.T, a.H, ', graph replay' if a.graph else ''),
                          samples_per_s=a.B * a.steps / dt, ms_per_step=1e3 * dt / a.steps,
                          loss=float(m._loss_dev.item()),
                          kernel_ms_per_step={n: round(v['total_ms'] / a.steps, 4) for n, v in
                                              sorted(ks.items(), key=lambda kv: -kv[1]['total_ms'])})))


if __name__ == '__main__':
    main()

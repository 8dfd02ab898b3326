"""End-to-end rate with host-resident inputs: numpy arrays -> Feeder (pinned, copy stream) ->
train_step, on the DeepFM Criteo-Kaggle shape.  Reports the PCIe-inclusive samples/s for inputs in
the reference's dtypes (int64 / float64 / int64) and for already-narrow inputs (int32 / float32).

  python tools/feed_bench.py [--B 65536] [--batches 24]
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
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--B', type=int, default=65536)
    ap.add_argument('--batches', type=int, default=24)
    a = ap.parse_args()
    import recommendation_amd as ra
    from recommendation_amd.feeder import Feeder
    from recommendation_amd.synthetic import CRITEO_KAGGLE
    sess = ra.Session('cuda:0', seed=1)
    model = bench.build_model(SimpleNamespace(model='DeepFM', embed=16, comm='rccl', shard='table'), sess,
                              CRITEO_KAGGLE, 0, 1)
    rng = np.random.default_rng(0)
    n = a.B * a.batches + 1                       # +1: the reference loop stops at `end < total`
    Xc = np.stack([rng.integers(0, v, n) for v in CRITEO_KAGGLE], 1)
    Xf = rng.standard_normal((n, 13))
    y = rng.integers(0, 2, n)
    ranges = [(i * a.B, (i + 1) * a.B) for i in range(a.batches)]
    out = {}
    for name, (c, f, l) in (('int64/float64 inputs (reference dtypes)', (Xc, Xf, y)),
                            ('int32/float32 inputs', (Xc.astype(np.int32), Xf.astype(np.float32),
                                                      y.astype(np.float32)))):
        for rep in range(2):                      # first pass warms the buffers
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for idx, xc, yy in Feeder(sess, c, f, l, ranges):
                model.train_step(idx, xc, yy)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        out[name] = dict(samples_per_s=a.B * a.batches / dt, ms_per_step=1e3 * dt / a.batches)
    print(json.dumps(dict(workload='DeepFM c2 from host numpy through Feeder, B=%d' % a.B, host_threads=torch.get_num_threads(),
                          **out)))


if __name__ == '__main__':
    main()

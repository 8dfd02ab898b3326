"""Training step of DeepFM at the Criteo-Kaggle shape for the reference's mini-batch sizes: eager launches against
the HIP-graph replay of the same step (CTRModel.train_step_graphed)."""
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import recommendation_amd as ra  # noqa: E402
from recommendation_amd.synthetic import CRITEO_KAGGLE, make_batches  # noqa: E402


def main():
    sess = ra.Session('cuda:0', seed=3)
    args = SimpleNamespace(model='DeepFM', embed=16, shard='table', comm='rccl')
    model = bench.build_model(args, sess, CRITEO_KAGGLE, 0, 1)
    sizes = [int(a) for a in sys.argv[1:]] or [128, 1024, 4096, 16384]
    for B in sizes:
        batches = make_batches(CRITEO_KAGGLE, 13, B, 4, 1.05, 7, 'cuda:0')
        res = {}
        for name, fn in (('eager', model.train_step), ('graph', model.train_step_graphed)):
            for i in range(5):
                fn(*batches[i % 4])
            torch.cuda.synchronize()
            n = 200
            t0 = time.perf_counter()
            for i in range(n):
                fn(*batches[i % 4])
            torch.cuda.synchronize()
            res[name] = (time.perf_counter() - t0) / n * 1e6
        # the reference's loop: numpy feeds in, python float loss out (model.run), one host round trip per step
        host = [(b[0].cpu().numpy().astype('int64'), b[1].cpu().numpy().astype('float64'), b[2].cpu().numpy())
                for b in batches]
        for i in range(5):
            model.run(sess, *host[i % 4], is_train=True)
        t0 = time.perf_counter()
        for i in range(n):
            model.run(sess, *host[i % 4], is_train=True)
        res['run'] = (time.perf_counter() - t0) / n * 1e6
        print(json.dumps(dict(workload='DeepFM Criteo-Kaggle shape, fwd+bwd+Adam', batch=B,
                              eager_us=round(res['eager'], 1), graph_us=round(res['graph'], 1), run_numpy_feeds_us=round(res['run'], 1),
                              eager_samples_per_s=round(B / res['eager'] * 1e6),
                              graph_samples_per_s=round(B / res['graph'] * 1e6))), flush=True)


if __name__ == '__main__':
    main()

"""Times the DeepFM c2 step with the table update exactly as tf.train.AdamOptimizer applies it (every row, every step):
through the dense gradient buffer (table_grads=True) and as row-wise pass + decay pass (table_grads=False)."""
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import recommendation_amd as ra                                            # noqa: E402
from recommendation_amd import ops                                         # noqa: E402
from recommendation_amd.synthetic import CRITEO_KAGGLE, make_batches       # noqa: E402

ra.enable_zoo_imports()
from DeepFM.DeepFM import DeepFM                                           # noqa: E402

sess = ra.Session('cuda:0', seed=3)
out = {}
for keep in (True, False):
    args = SimpleNamespace(embed_size=16, regular_rate=0.1, learning_rate=0.001, decay_steps=10000, decay_rate=0.99, mode='S',
                           adam_mode='tf1', init_std=0.01, deep_layers=[256, 128, 64, 32], dropout_fm=0.5, dropout_deep=0.2,
                           table_grads=keep)
    m = DeepFM(args, len(CRITEO_KAGGLE), 13, CRITEO_KAGGLE)
    m.build()
    m.initialize(sess, seed=1234)
    batches = make_batches(CRITEO_KAGGLE, 13, 65536, 4, 1.05, 7, sess.device)
    for i in range(3):
        m.train_step(*batches[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(10):
        m.train_step(*batches[i % 4])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    prof = ops.KernelTimer(pool=400)
    ops.set_timer(prof)
    sess.side_enabled = False
    m.train_step(*batches[0])
    sess.side_enabled = True
    ops.set_timer(None)
    ks = {k: round(v['total_ms'], 3) for k, v in prof.summary().items() if v['total_ms'] > 0.05}
    out['table_grads=%s' % keep] = dict(ms_per_step=round(1e3 * dt, 3), samples_per_s=round(65536 / dt), kernels_ms=ks)
    m.close()
    del m
    torch.cuda.empty_cache()
print(json.dumps(out))

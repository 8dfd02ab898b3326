"""Per-operator times of the DCN share's step (Criteo-1TB / 8 tables, B = 8192, E = 64) with the row-wise ('lazy') Adam and
with tf.train.AdamOptimizer's every-row update without a gradient buffer ('tf1', table_grads=False)."""
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import recommendation_amd as ra                                            # noqa: E402
from recommendation_amd import ops                                         # noqa: E402
from recommendation_amd.synthetic import CRITEO_1TB, make_batches, scaled  # noqa: E402

ra.enable_zoo_imports()
from DCN.DCN import DCN                                                    # noqa: E402

cards = scaled(CRITEO_1TB, 8)
sess = ra.Session('cuda:0', seed=3)
out = {}
for mode in ('lazy', 'tf1'):
    args = SimpleNamespace(embed_size=64, regular_rate=0.1, learning_rate=0.001, decay_steps=10000, decay_rate=0.99, mode='S',
                           adam_mode=mode, init_std=0.01, deep_layers=[1024, 1024, 1024], cross_layer_num=6, dropout=0.2,
                           table_grads=False)
    m = DCN(args, len(cards), 13, cards)
    m.build()
    m.initialize(sess, seed=1234)
    batches = make_batches(cards, 13, 8192, 4, 1.05, 7, sess.device)
    for i in range(3):
        m.train_step(*batches[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        m.train_step(*batches[i % 4])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    prof = ops.KernelTimer(pool=400)
    ops.set_timer(prof)
    sess.side_enabled = False
    m.train_step(*batches[0])
    sess.side_enabled = True
    ops.set_timer(None)
    ks = {k: round(v['total_ms'], 3) for k, v in prof.summary().items() if v['total_ms'] > 0.02}
    out[mode] = dict(ms_per_step=round(1e3 * dt, 3), kernels_ms=ks)
    m.close()
    del m
    torch.cuda.empty_cache()
print(json.dumps(out))

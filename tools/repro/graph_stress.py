"""Stress of the captured-training-step lifetime: many small models captured in turn in one process, older models
(and their graphs) dropped and collected at chosen points.  CTR_GRAPH_KEEP=0 lets the graphs be destroyed.
  python tools/repro/graph_stress.py [rounds] [gc_point]   gc_point: none | after_capture | before_replay | random"""
import ctypes
import gc
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if os.environ.get('CTR_SEGV_TRACE'):
    ctypes.CDLL(os.path.join(ROOT, 'tools', 'repro', 'libsegvtrace.so')).segv_trace_install()
import numpy as np
import torch

import recommendation_amd as ra
from oracle import ctr_oracle as O
from tests.helpers import tiny_batch, tiny_cfg
from tests.test_gpu_models import _build

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
gc_point = sys.argv[2] if len(sys.argv) > 2 else 'random'
sess = ra.Session('cuda:0', seed=7)
rnd = random.Random(0)
names = ['DeepFM', 'DCN', 'NFM', 'PNN', 'ESMM', 'AFM']
old = []
for r in range(rounds):
    name = names[r % len(names)]
    cfg = tiny_cfg(name, 'S' if r % 3 else 'R')
    cfg.adam_mode = 'tf1'
    P = O.init_params(name, cfg, seed=r, std=0.1)
    if name == 'AFM':
        P = {k: np.abs(v) for k, v in P.items()}
    m = _build(sess, name, cfg, P)
    B = [8, 48, 64, 300][r % 4]
    Xc, Xf, y = tiny_batch(cfg, B=B, seed=r)
    if name == 'AFM':
        Xf = np.abs(Xf) * 0.1
    idx, xc, yy = m._upload(Xc, Xf, y)
    sess.side_enabled = (r % 5 != 4)
    m.train_step_graphed(idx, xc, yy)              # eager step + capture
    if gc_point in ('after_capture', 'before_replay') or (gc_point == 'random' and rnd.random() < 0.5):
        old.clear()
        gc.collect()
    for _ in range(3):
        m.train_step_graphed(idx, xc, yy)          # replays
    m.predict_step_graphed(idx, xc)
    torch.cuda.synchronize()
    old.append(m)
    if len(old) > 3 or (gc_point == 'random' and rnd.random() < 0.3):
        old.pop(0)
    if r % 10 == 0:
        print('round', r, name, 'ok', flush=True)
sess.side_enabled = True
print('done: %d rounds, gc_point=%s, keep=%s' % (rounds, gc_point, os.environ.get('CTR_GRAPH_KEEP', '1')))

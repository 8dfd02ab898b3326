"""Runs the table backward of the DeepFM benchmark shape alone (pair sort, segment-reduce + row-wise Adam) on one
stream, a 1 GiB fill before every pass so that nothing is cached; meant to be run under rocprofv3 --kernel-trace
--stats for the per-kernel times (tools/apply_prof.sh)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recommendation_amd as ra
from recommendation_amd import ops
from recommendation_amd._lib import OPT_ADAM
from recommendation_amd.synthetic import CRITEO_KAGGLE, zipf_ids

sess = ra.Session('cuda:0')
B, E, C = 65536, int(os.environ.get('APPLY_E', 16)), 13
cards = CRITEO_KAGGLE
F, N = len(cards), len(cards) + C
rows = int(np.sum(cards))
gen = torch.Generator(device='cuda'); gen.manual_seed(1)
idx = zipf_ids(cards, B, 1.05, gen, 'cuda')
offs = torch.as_tensor(np.concatenate([[0], np.cumsum(cards)]).astype(np.int64), device='cuda')
W = torch.randn(rows, E, device='cuda') * 0.01
if os.environ.get('APPLY_MV', '0') == '1':      # Adam's m | v side by side: one 128-byte line per touched row at E = 16
    mv = torch.zeros(rows, 2 * E, device='cuda')
    m, v = mv[:, :E], mv[:, E:]
else:
    m = torch.zeros_like(W); v = torch.zeros_like(W)
dz = torch.randn(B, N * E, device='cuda')
keys = torch.empty(B * F, dtype=torch.int32, device='cuda'); src = torch.empty_like(keys)
junk = torch.empty(1 << 28, device='cuda')
opt = ops.make_opt(OPT_ADAM, lr=1e-3)
for it in range(12):
    junk.fill_(float(it))
    ops.emb_sort(idx, offs, N=F, total_rows=rows, rmode=False, keys_out=keys, src_out=src, ws=sess.ws)
    ops.emb_sparse_apply(keys, src, B * F, N=F, E=E, total_rows=rows, dz=dz, opt=opt, tables=W, slot0=m, slot1=v,
                         ws=sess.ws)
torch.cuda.synchronize()
print('done')

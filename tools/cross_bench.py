"""Times ctr_cross_fwd / ctr_cross_bwd alone at the DCN benchmark shape (B = 8192, D = 1677) for several L, next to a
plain copy of the same bytes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recommendation_amd as ra
from recommendation_amd import ops

sess = ra.Session('cuda:0')
B, D = 8192, 1677
DP = (D + 3) // 4 * 4
x0 = torch.randn(B, DP, device='cuda')[:, :D]
PAD = int(os.environ.get('CROSS_STACK_LD', DP + 1024))      # row stride of the [x_L | tower] buffers (2701 = unpadded)
xL = torch.empty(B, PAD, device='cuda')[:, :D]
g = torch.randn(B, PAD, device='cuda')[:, :D]
dx0 = torch.zeros(B, DP, device='cuda')[:, :D]


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for L in (6,):
    w = (torch.randn(L, DP, device="cuda") * 0.01)[:, :D]
    b = (torch.randn(L, DP, device="cuda") * 0.01)[:, :D]
    aux = torch.zeros(B, 2 * L, device='cuda')
    dw, db = torch.zeros(L, DP, device='cuda')[:, :D], torch.zeros(L, DP, device='cuda')[:, :D]
    tf = timeit(lambda: ops.cross_fwd(x0, w, b, xL, aux, ws=sess.ws))
    tb = timeit(lambda: ops.cross_bwd(x0, w, b, aux, g, dx0, dw, db, ws=sess.ws, accumulate=True))
    print('L=%d  fwd %.1f us (%.2f TB/s of 2*D*4*B)   bwd %.1f us (%.2f TB/s of 4*D*4*B)' % (
        L, tf, 2 * D * 4 * B / tf * 1e-6, tb, 4 * D * 4 * B / tb * 1e-6), flush=True)
a = torch.empty(B * DP, device='cuda')
c = torch.empty_like(a)
t = timeit(lambda: c.copy_(a))
print('copy of x0-sized buffer: %.1f us (%.2f TB/s read+write)' % (t, 2 * a.numel() * 4 / t * 1e-6))

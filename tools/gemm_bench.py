"""Micro-benchmark of ctr_gemm_f32 on the tower shapes (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import recommendation_amd as ra
from recommendation_amd import ops

sess = ra.Session('cuda:0')
if os.environ.get('GEMM_MODE'):          # fp32 | split | bf16
    ops.set_gemm_mode(os.environ['GEMM_MODE'])
ws = sess.ws
B = 65536
if len(sys.argv) > 1 and sys.argv[1] == 'ksweep':   # time = a + b*K: fixed cost (prologue + C store) vs per-k-step cost
    shapes = [('k%d' % k, B, 256, k, 0, 0) for k in (16, 32, 64, 128, 256, 432, 864, 1728)]
    shapes += [('t%d' % k, B, 256, k, 0, 1) for k in (16, 128, 432)]
    shapes += [('k%d' % k, B, 256, k, 0, 0) for k in (416, 428, 429, 430, 431, 432, 433, 448)]
else:
  shapes = [  # (name, M, N, K, ta, tb)
    ('fwd0', B, 256, 429, 0, 0), ('fwd1', B, 128, 256, 0, 0), ('fwd2', B, 64, 128, 0, 0), ('fwd3', B, 32, 64, 0, 0),
    ('dx0', B, 429, 256, 0, 1), ('dx1', B, 256, 128, 0, 1), ('dx2', B, 128, 64, 0, 1), ('dx3', B, 64, 32, 0, 1),
    ('dw0', 429, 256, B, 1, 0), ('dw1', 256, 128, B, 1, 0), ('dw2', 128, 64, B, 1, 0), ('dw3', 64, 32, B, 1, 0),
    ('sq4k', 4096, 4096, 4096, 0, 0),
  ]
if len(sys.argv) > 1 and sys.argv[1] == 'dcn':         # the 1024-wide tower of the DCN shape at B = 8192
    shapes = [('fwd0', 8192, 1024, 1677, 0, 0), ('fwd1', 8192, 1024, 1024, 0, 0),
              ('dx0', 8192, 1677, 1024, 0, 1), ('dx1', 8192, 1024, 1024, 0, 1), ('dx1m', 8192, 1024, 1024, 0, 1),
              ('fwd1m', 8192, 1024, 1024, 0, 0),
              ('dw0', 1677, 1024, 8192, 1, 0), ('dw1', 1024, 1024, 8192, 1, 0)]
if len(sys.argv) > 2 and sys.argv[1] == 'one':         # a single named shape, for counter passes
    shapes = [sh for sh in shapes if sh[0] in sys.argv[2:]]
if len(sys.argv) > 1 and sys.argv[1] == 'wgrad':
    shapes = []
for name, M, N, K, ta, tb in shapes:
    def mk(r, c):
        ld = (c + 3) // 4 * 4
        return torch.randn(r, ld, device='cuda')[:, :c]
    A = mk(K, M) if ta else mk(M, K)
    Bm = mk(N, K) if tb else mk(K, N)
    C = mk(M, N)
    bias = torch.randn(N, device='cuda')
    kw = {}
    if name.endswith('m'):                      # with the relu/dropout-backward mask epilogue
        kw = dict(mask_src=mk(M, N), mask_scale=1.25)
    if os.environ.get('F16'):                   # both operands' maxima known: the two-term fp16 split
        am = torch.zeros(3, ops.AMAX_FLOATS, device='cuda')
        ops.absmax(A, am[0]); ops.absmax(Bm, am[1])
        kw.update(a_amax=am[0], b_amax=am[1], amax_out=am[2])
    for _ in range(3):
        ops.gemm(A, Bm, C, trans_a=bool(ta), trans_b=bool(tb), ws=ws, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        ops.gemm(A, Bm, C, trans_a=bool(ta), trans_b=bool(tb), ws=ws, **kw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    err = ''
    if M * N * K <= 2 ** 33:
        ref = (A.double().t() if ta else A.double()) @ (Bm.double().t() if tb else Bm.double())
        if 'mask_src' in kw:
            ref = ref * torch.where(kw['mask_src'] > 0, 1.25, 0.0)
        err = '  relerr %.2e' % ((C.double() - ref).abs().max() / ref.abs().max()).item()
    print('%-6s M=%6d N=%5d K=%6d  %8.1f us  %6.1f TF%s' % (name, M, N, K, us, 2.0 * M * N * K / us / 1e6, err))
if len(sys.argv) > 1 and sys.argv[1] != 'wgrad':
    sys.exit(0)
print('--- dense_wgrad')
for name, M, N, K in [('dw0', 429, 256, B), ('dw1', 256, 128, B), ('dw2', 128, 64, B), ('dw3', 64, 32, B), ('dwo', 74, 1, B),
                      ('din1', 128, 80, 409600), ('din2', 80, 40, 409600), ('din3', 512, 80, 409600)]:
    x = torch.randn(K, (M + 3) // 4 * 4, device='cuda')[:, :M]
    g = torch.randn(K, (N + 3) // 4 * 4, device='cuda')[:, :N]
    dW = torch.zeros(M, N, device='cuda'); db = torch.zeros(N, device='cuda')
    for _ in range(3):
        ops.dense_wgrad(x, g, dW, db, ws=ws)
    torch.cuda.synchronize()
    ref = x.double().t() @ g.double()
    err = ((dW.double() - ref).abs().max() / ref.abs().max()).item()
    errb = ((db.double() - g.double().sum(0)).abs().max() / g.double().sum(0).abs().max()).item()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.dense_wgrad(x, g, dW, db, ws=ws)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print('%-4s M=%4d N=%4d K=%6d %8.1f us %6.1f TF  relerr %.2e  bias %.2e' % (name, M, N, K, us, 2.0 * M * N * K / us / 1e6, err, errb))

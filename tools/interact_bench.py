"""Times the interaction kernels at the 39-field / 741-pair / batch-65536 shape with HIP events.

  python tools/interact_bench.py [--B 65536] [--N 39] [--E 16] [--A 16]
"""
import argparse
import json

import torch

from recommendation_amd import ops
from recommendation_amd.session import Session


def timed(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--B', type=int, default=65536)
    ap.add_argument('--N', type=int, default=39)
    ap.add_argument('--E', type=int, default=16)
    ap.add_argument('--A', type=int, default=16)
    ap.add_argument('--P', type=int, default=10, help='PNN product_size')
    a = ap.parse_args()
    sess = Session('cuda:0')
    dev, ws = sess.device, sess.ws
    B, N, E, A = a.B, a.N, a.E, a.A
    g = torch.Generator(device=dev).manual_seed(1)
    r = lambda *s: torch.randn(*s, device=dev, generator=g) * 0.3
    z, dz, x, dx = r(B, N * E), torch.zeros(B, N * E, device=dev), torch.zeros(B, E, device=dev), r(B, E)
    W, b, h = r(E, A), r(A), r(A)
    dW, db, dh = torch.zeros(E, A, device=dev), torch.zeros(A, device=dev), torch.zeros(A, device=dev)
    drop = ops.make_dropout(0.3, 1, 4)
    zbytes = B * N * E * 4
    out = {}
    t = timed(lambda: ops.afm_attn_fwd(z, W, b, h, x, N=N, E=E, pair_drop=drop))
    out['afm_attn_fwd'] = dict(ms=t, GBps=(zbytes + B * E * 4) / t / 1e6)
    t = timed(lambda: ops.afm_attn_bwd(z, dx, W, b, h, dz, dW, db, dh, N=N, E=E, ws=ws, pair_drop=drop))
    out['afm_attn_bwd'] = dict(ms=t, GBps=(2 * zbytes + B * E * 4) / t / 1e6)
    t = timed(lambda: ops.afm_pair_fwd(z, x, N=N, E=E, pair_drop=drop))
    out['afm_pair_fwd'] = dict(ms=t, GBps=(zbytes + B * E * 4) / t / 1e6)
    t = timed(lambda: ops.afm_pair_bwd(z, dx, dz, N=N, E=E, pair_drop=drop))
    out['afm_pair_bwd'] = dict(ms=t, GBps=(2 * zbytes + B * E * 4) / t / 1e6)
    Pn = a.P
    th, usq, coef, dth = r(Pn, N), torch.zeros(B, Pn, device=dev), r(B, Pn), torch.zeros(Pn, N, device=dev)
    t = timed(lambda: ops.pnn_inner_fwd(z, th, usq, N=N, E=E))
    out['pnn_inner_fwd'] = dict(ms=t, GBps=zbytes / t / 1e6)
    t = timed(lambda: ops.pnn_inner_bwd(z, th, coef, dz, dth, N=N, E=E, ws=ws))
    out['pnn_inner_bwd'] = dict(ms=t, GBps=2 * zbytes / t / 1e6)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()

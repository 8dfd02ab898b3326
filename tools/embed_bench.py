"""Times the embedding front-end kernel at the DeepFM Criteo-Kaggle shape in pieces: gather only, gather + FM
sums, uniform and Zipf ids, and a plain streaming copy of the same byte count for the ceiling."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendation_amd import ops

V = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992, 5461306, 10,
     5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]


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


def timeit_cold(fn, n=10):
    """Each timed call follows a 1 GiB fill, so neither L2 nor the Infinity Cache holds table rows."""
    junk = torch.empty(1 << 28, device='cuda')
    tot = 0.0
    for i in range(n + 2):
        junk.fill_(float(i))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        if i >= 2:
            tot += a.elapsed_time(b)
    return tot / n * 1e3


def main():
    B, F, C, E = 65536, 26, 13, 16
    dev = torch.device('cuda:0')
    off = np.concatenate([[0], np.cumsum(V)]).astype(np.int64)
    tables = torch.randn(int(off[-1]), E, device=dev) * 0.01
    row_offset = torch.from_numpy(off).to(dev)
    rng = np.random.default_rng(1)
    xc = torch.randn(B, C, device=dev)
    z = torch.empty(B, 432, device=dev)
    concat = torch.empty(B, 74, device=dev)
    s = torch.empty(B, E, device=dev)
    alg = B * F * (4 + 2 * E * 4)
    for name in ('uniform', 'zipf'):
        cols = []
        for v in V:
            if name == 'uniform':
                cols.append(rng.integers(0, v, B))
            else:
                cols.append(np.minimum(rng.zipf(1.05, B) - 1, v - 1) if v > 1 else np.zeros(B, np.int64))
        idx = torch.from_numpy(np.stack(cols, 1).astype(np.int32)).to(dev)
        t0 = timeit(lambda: ops.embed_fwd(tables, row_offset, idx, xc, None, E=E, z_out=z, cont_tail=True))
        t1 = timeit(lambda: ops.embed_fwd(tables, row_offset, idx, xc, None, E=E, z_out=z, cont_tail=True,
                                          first_out=concat[:, :F], s_out=s, second_out=concat[:, F:F + E],
                                          drop_first=ops.make_dropout(0.5, 1, 1), drop_second=ops.make_dropout(0.5, 1, 2)))
        print('%s ids: gather only %.1f us (%.2f TB/s)   gather + FM sums + dropout %.1f us (%.2f TB/s)' % (
            name, t0, alg / t0 * 1e-6, t1, alg / t1 * 1e-6), flush=True)
        t2 = timeit_cold(lambda: ops.embed_fwd(tables, row_offset, idx, xc, None, E=E, z_out=z, cont_tail=True,
                                               first_out=concat[:, :F], s_out=s, second_out=concat[:, F:F + E],
                                               drop_first=ops.make_dropout(0.5, 1, 1), drop_second=ops.make_dropout(0.5, 1, 2)))
        print('%s ids, caches flushed before each call: gather + FM sums + dropout %.1f us (%.2f TB/s)' % (
            name, t2, alg / t2 * 1e-6), flush=True)
    a = torch.empty(alg // 8, device=dev)
    b = torch.empty_like(a)
    tc = timeit(lambda: b.copy_(a))
    print('streaming copy of the same bytes: %.1f us (%.2f TB/s)' % (tc, alg / tc * 1e-6))
    tc = timeit_cold(lambda: b.copy_(a))
    print('streaming copy, caches flushed before each call: %.1f us (%.2f TB/s)' % (tc, alg / tc * 1e-6))


if __name__ == '__main__':
    main()

"""Times the chained tower-layer kernel against one GEMM launch per layer (same widths, same dropout)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendation_amd import ops


def timeit(fn, n=20):
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


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    dev = torch.device('cuda:0')
    ws = ops.Workspace(dev)
    for K0, widths in ((128, (64, 32)), (128, (64, 32, 16)), (256, (64, 32))):
        x = torch.randn(M, K0, device=dev)
        Ws, bs, outs, kin = [], [], [], K0
        for w in widths:
            Ws.append(torch.randn(kin, w, device=dev) / kin ** 0.5)
            bs.append(torch.randn(w, device=dev) * 0.1)
            outs.append(torch.empty(M, w, device=dev))
            kin = w
        for rate in (0.0, 0.2):
            drops = [ops.make_dropout(rate, 5, 20 + i) for i in range(len(widths))]

            def chain():
                ops.mlp_chain_fwd(x, Ws, bs, drops, outs)

            def layers():
                h = x
                for i in range(len(widths)):
                    ops.gemm(h, Ws[i], outs[i], bias=bs[i], relu=True, dropout=drops[i], ws=ws)
                    h = outs[i]
            flops = 2.0 * M * sum(a * b for a, b in zip((K0,) + widths[:-1], widths))
            tc, tl = timeit(chain), timeit(layers)
            print('M=%d %d->%s dropout %.1f: chain %.1f us (%.1f TF)  per-layer %.1f us (%.1f TF)' % (
                M, K0, '->'.join(map(str, widths)), rate, tc, flops / tc * 1e-6, tl, flops / tl * 1e-6), flush=True)


if __name__ == '__main__':
    main()

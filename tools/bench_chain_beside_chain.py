"""A chain of short kernels on the main stream beside a chain of long machine-filling kernels on a side stream (both as replayed
linear graphs, hand-over as in the step): how much longer does the main chain take than alone?  Variants of the side kernels:
many small workgroups (a streaming elementwise op) against few workgroups (a narrow launch)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd import ops

dev = torch.device('cuda:0')
main = torch.cuda.Stream()
side = ops.concurrent_stream(dev, [torch.cuda.current_stream(), main])
small = torch.zeros(1 << 20, device=dev)
big = torch.zeros(1 << 27, device=dev)            # 512 MB
mm_a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
mm_b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
mm_c = torch.empty(8192, 8192, device=dev, dtype=torch.bfloat16)


def graph_of(stream, fn):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=stream):
            fn()
    torch.cuda.synchronize()
    return g


def main_chain():
    for _ in range(100):
        small.add_(1.0)


sides = {
    'streaming elementwise, 512 MB per kernel': lambda: [big.mul_(1.0001) for _ in range(12)],
    'bf16 GEMM 8192^3': lambda: [torch.mm(mm_a, mm_b, out=mm_c) for _ in range(4)],
}
gm = graph_of(main, main_chain)


def timed(gs):
    ts = []
    for _ in range(7):
        torch.cuda.synchronize()
        a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        with torch.cuda.stream(main):
            a.record()
            if gs is not None:
                side.wait_stream(main)
        if gs is not None:
            with torch.cuda.stream(side):
                gs.replay(); c.record()
        with torch.cuda.stream(main):
            gm.replay(); b.record()
        torch.cuda.synchronize()
        ts.append((a.elapsed_time(b) * 1e3, a.elapsed_time(c) * 1e3 if gs is not None else 0.0))
    ts.sort()
    return ts[len(ts) // 2]


t0 = timed(None)[0]
print('main chain (100 short kernels) alone: %.0f us' % t0)
for name, fn in sides.items():
    gs = graph_of(side, fn)
    tm, tsd = timed(gs)
    print('beside a side chain of %s: main %.0f us (x%.2f), side chain %.0f us' % (name, tm, tm / t0, tsd))

"""Does launching a batch of kernels on a side stream hold up the NEXT launch on the main stream?  (kernel trace of the step: the
main queue idles 160-840 us behind the hand-over of the three largest weight-gradient batches, tools/trace_gaps.py.)
main: A1 (short kernels) | hand-over: side waits for main, B = nb kernels of `long_us` each on the side stream | main: A2 (one
short kernel).  Reported: time from the end of A1 to the end of A2 on the main stream (ideal: one short kernel)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd import ops

dev = torch.device('cuda:0')
main = torch.cuda.Stream()
side = ops.concurrent_stream(dev, [torch.cuda.current_stream(), main])
x = torch.zeros(1 << 20, device=dev)
# calibrate _sleep
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(1000000); e1.record(); torch.cuda.synchronize()
cyc_per_us = 1000000 / (e0.elapsed_time(e1) * 1e3)
print('_sleep: %.1f cycles per us' % cyc_per_us)


def sleep_us(us):
    torch.cuda._sleep(int(us * cyc_per_us))


def make_graph(stream, fn):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        fn()                                     # warm-up
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=stream):
            fn()
    torch.cuda.synchronize()
    return g


def trial(nb, long_us, b_graph, a_graph, wait=True, reps=5):
    fa1 = lambda: [sleep_us(20) for _ in range(10)]
    fa2 = lambda: sleep_us(20)
    fb = lambda: [sleep_us(long_us) for _ in range(nb)]
    ga1 = make_graph(main, fa1) if a_graph else None
    ga2 = make_graph(main, fa2) if a_graph else None
    gb = make_graph(side, fb) if b_graph else None
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        ev1, ev2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(main):
            ga1.replay() if ga1 else fa1()
            ev1.record()
            if wait:
                side.wait_stream(main)
        with torch.cuda.stream(side):
            gb.replay() if gb else fb()
        with torch.cuda.stream(main):
            ga2.replay() if ga2 else fa2()
            ev2.record()
        torch.cuda.synchronize()
        out.append(ev1.elapsed_time(ev2) * 1e3)
    return min(out), sorted(out)[len(out) // 2]


for nb, long_us in ((1, 100), (8, 100), (32, 100), (64, 50), (64, 10)):
    for b_graph, a_graph in ((True, True), (False, True), (True, False), (False, False)):
        mn, med = trial(nb, long_us, b_graph, a_graph)
        print('side batch %2d x %3d us as %s, main segments as %s: end of A1 -> end of A2 = %.0f us (median %.0f)' % (
            nb, long_us, 'graph' if b_graph else 'eager launches', 'graphs' if a_graph else 'eager launches', mn, med))
mn, med = trial(32, 100, True, True, wait=False)
print('side batch 32 x 100 us as graph, no wait on main: %.0f us (median %.0f)' % (mn, med))

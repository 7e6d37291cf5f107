"""External event-record nodes inside a captured graph (torch.cuda.Event(external=True) -> hipEventRecordWithFlags(External)):
(1) do they work on this runtime - a side stream that waits for the event starts when the main graph reaches the node, not at
its end and not at once; (2) what does a node cost inside a linear graph?  If cheap, the step's main chain could be ONE graph
with record nodes at the hand-over points instead of ~15 linear segments (each boundary: ~15 us)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd import ops

import ctypes
hip = ctypes.CDLL('libamdhip64.so')
hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
hip.hipEventRecordWithFlags.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
hip.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]


class XEvent:
    """external event through the HIP API (torch refuses external=True on ROCm)"""
    def __init__(self):
        self.h = ctypes.c_void_p()
        assert hip.hipEventCreateWithFlags(ctypes.byref(self.h), 2) == 0          # hipEventDisableTiming

    def record(self):
        rc = hip.hipEventRecordWithFlags(self.h, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), 1)      # hipEventRecordExternal
        if rc != 0:
            raise RuntimeError('hipEventRecordWithFlags -> %d' % rc)

    def wait_on(self, stream):
        rc = hip.hipStreamWaitEvent(ctypes.c_void_p(stream.cuda_stream), self.h, 0)
        if rc != 0:
            raise RuntimeError('hipStreamWaitEvent -> %d' % rc)


dev = torch.device('cuda:0')
main = torch.cuda.Stream()
side = ops.concurrent_stream(dev, [torch.cuda.current_stream(), main])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(1000000); e1.record(); torch.cuda.synchronize()
cpu = 1000000 / (e0.elapsed_time(e1) * 1e3)
sl = lambda us: torch.cuda._sleep(int(us * cpu))


def capture(fn):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(main):
        with torch.cuda.graph(g, stream=main):
            fn()
    torch.cuda.synchronize()
    return g


nk = 20
for nev in (0, 1, 4, 10, 19):
    evs = [XEvent() for _ in range(nev)]
    every = nk // (nev + 1) if nev else nk + 1

    def body():
        k = 0
        for i in range(nk):
            sl(20)
            if nev and (i + 1) % every == 0 and k < nev:
                evs[k].record()
                k += 1
    try:
        g = capture(body)
    except Exception as ex:
        print('capture with %d external record nodes failed: %s  (ROCm 7.2: hipEventRecordWithFlags(External) = hipErrorInvalidValue '
              'under capture; torch refuses Event(external=True) on ROCm for the same reason)' % (nev, str(ex)[:200]))
        sys.exit(0)
    ts = []
    for _ in range(7):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(main):
            a.record(); g.replay(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    print('graph of %d x 20 us kernels with %2d external record nodes: %.0f us (median %.0f)' % (nk, nev, min(ts), sorted(ts)[3]))

# semantics: side waits for a node in the middle of the main graph
ev = XEvent()


def body2():
    for i in range(10):
        sl(20)
    ev.record()
    for i in range(10):
        sl(20)
g = capture(body2)
for order in ('launch main, then side waits', ):
    res = []
    for _ in range(5):
        torch.cuda.synchronize()
        a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        with torch.cuda.stream(main):
            a.record(); g.replay(); c.record()
        with torch.cuda.stream(side):
            ev.wait_on(side)
            sl(50)
            b.record()
        torch.cuda.synchronize()
        res.append((a.elapsed_time(b) * 1e3, a.elapsed_time(c) * 1e3))
    print('side kernel (50 us) behind the record node in the middle of a 400 us main graph ends at %s us after the start (main graph ends at %s); '
          'expected ~250 if the wait follows the node, ~50 if it is ignored, ~450 if it waits for the whole graph' % (
              ' '.join('%.0f' % r[0] for r in res), ' '.join('%.0f' % r[1] for r in res)))

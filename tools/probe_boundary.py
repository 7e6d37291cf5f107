"""Does a graph-launch boundary on one stream stall while another stream is busy?  (round 5: the main chain of the step idles ~0.86 ms
behind every hand-over of a weight-gradient batch although half of the CUs are free - profiles/r05_handover.txt.)
Main stream: graphs of N tiny kernels; side stream: a graph of wide streaming kernels started by an event of the main stream, then of
narrow kernels that hold half of the CUs (tools/probe/spin.hip; build it first:
hipcc --offload-arch=gfx950 -O3 -fPIC -shared tools/probe/spin.hip -o tools/probe/libspin.so)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd import ops

dev = torch.device('cuda:0')
main = torch.cuda.Stream(dev)
side = ops.concurrent_stream(dev, [main])
small = torch.zeros(1 << 14, device=dev)
big = torch.zeros(int(os.environ.get('BIG', 1 << 26)), device=dev)
N = int(os.environ.get('N', '60'))
NSIDE = int(os.environ.get('NSIDE', '10'))


def graph_of(fn, stream):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=stream):
            fn()
    return g


def tiny(n):
    def f():
        for _ in range(n):
            small.add_(1.0)
    return f


def wide():
    for _ in range(NSIDE):
        big.mul_(1.0001)


g1, g2, g12 = graph_of(tiny(N), main), graph_of(tiny(N), main), graph_of(tiny(2 * N), main)
gs = graph_of(wide, side)
torch.cuda.synchronize()


def timed(body, reps=20):
    ts = []
    for _ in range(reps + 3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(main):
            e0.record(main)
            body()
            e1.record(main)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[3:])
    return ts[len(ts) // 2]


def side_alone():
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        e0.record(side); gs.replay(); e1.record(side)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


def handover(after):
    def f():
        g1.replay()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            gs.replay()
        after()
    return f


def eager2():
    for _ in range(N):
        small.add_(1.0)


print(f'side graph alone ({NSIDE} x streaming kernel over {big.numel() * 4 >> 20} MB): {side_alone():8.1f} us')
print(f'main: graph of {N} tiny kernels, then another            : {timed(lambda: (g1.replay(), g2.replay())):8.1f} us')
print(f'main: one graph of {2 * N}                                   : {timed(lambda: g12.replay()):8.1f} us')
print(f'main: graph, hand-over to the side graph, second graph   : {timed(handover(lambda: g2.replay())):8.1f} us')
print(f'main: graph, hand-over, the second {N} as eager launches   : {timed(handover(eager2)):8.1f} us')


def beside_whole():
    with torch.cuda.stream(side):
        gs.replay()
    g12.replay()


print(f'side graph started first (no dependency), then one graph of {2 * N} on main: {timed(beside_whole):8.1f} us')


def beside_two():
    with torch.cuda.stream(side):
        gs.replay()
    g1.replay(); g2.replay()


print(f'side graph started first, then the two graphs on main                : {timed(beside_two):8.1f} us')

# ---- narrow side kernels: `NWG` workgroups (one per CU: LDS_KB of LDS each) that hold their CUs for SPIN_US microseconds -----------------
import ctypes
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'probe', 'libspin.so'))
lib.probe_spin.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p]
NWG, LDS_KB, SPIN_US, NSPIN = (int(os.environ.get(k, d)) for k, d in (('NWG', '128'), ('LDS_KB', '96'), ('SPIN_US', '100'), ('NSPIN', '6')))


def narrow(stream_mem):
    def f():
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(NSPIN):
            lib.probe_spin(NWG, LDS_KB * 1024, float(SPIN_US), big.data_ptr() if stream_mem else None, big.numel(), s)
    return f


for mem in (False, True):
    gs = graph_of(narrow(mem), side)
    what = f'{NSPIN} x ({NWG} workgroups, {LDS_KB} KB LDS, {SPIN_US} us' + (', streaming memory)' if mem else ', registers only)')
    print(f'side graph alone, {what}: {side_alone():8.1f} us')
    print(f'   main: graph, hand-over to it, second graph              : {timed(handover(lambda: g2.replay())):8.1f} us')
    print(f'   main: graph, hand-over to it, one graph of {2 * N}          : {timed(handover(lambda: g12.replay())):8.1f} us')

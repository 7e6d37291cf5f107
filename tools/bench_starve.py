"""What starves a chain of short kernels beside a chain of long ones on another stream?  Side kernels = workgroups that hold a given
amount of LDS and spin (no memory traffic, no ALU pressure): the only thing varied is how many workgroups there are and how much
LDS each holds.  Main chain: 100 short kernels of 4 096 small workgroups.  Both are replayed linear graphs."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd import ops

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'probe', 'libprobe.so'))
lib.probe_spin.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_void_p]
lib.probe_short.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
lib.probe_stream.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device('cuda:0')
main = torch.cuda.Stream()
side = ops.concurrent_stream(dev, [torch.cuda.current_stream(), main])
x = torch.zeros(1 << 20, device=dev)
out = torch.zeros(4, device=dev)
cur = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


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
        assert lib.probe_short(ctypes.c_void_p(x.data_ptr()), x.numel(), cur()) == 0


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
print('main chain (100 short kernels of 4 096 workgroups) alone: %.0f us' % t0)
TICKS = 10000        # 100 us at 100 MHz
for wgs, lds in ((128, 100 << 10), (256, 100 << 10), (256, 40 << 10), (512, 64 << 10), (512, 16 << 10), (1024, 16 << 10), (2048, 16 << 10), (2048, 1 << 10),
                 (4096, 1 << 10), (8192, 1 << 10)):
    def side_chain(wgs=wgs, lds=lds):
        for _ in range(10):
            assert lib.probe_spin(wgs, lds, TICKS, ctypes.c_void_p(out.data_ptr()), cur()) == 0
    gs = graph_of(side, side_chain)
    tm, tsd = timed(gs)
    print('side: 10 kernels of %4d workgroups x %3d KB LDS spinning 100 us: main chain %.0f us (x%.2f), side chain %.0f us' % (
        wgs, lds >> 10, tm, tm / t0, tsd))

# the same with side workgroups that STREAM from HBM instead of spinning (128 workgroups x 98 KB LDS, like the grouped weight-gradient
# kernel at half width): does memory traffic on the side delay the dispatch of main's kernels?
src = torch.zeros(1 << 28, dtype=torch.uint8, device=dev)        # 256 MB
nvec = src.numel() // 16
for wgs, lds, mb_per_wg in ((128, 98 << 10, 2), (128, 98 << 10, 4), (256, 98 << 10, 2), (512, 31 << 10, 1), (1024, 29 << 10, 1)):
    vec_per_wg = mb_per_wg * (1 << 20) // 16

    def side_chain(wgs=wgs, lds=lds, vec_per_wg=vec_per_wg):
        for _ in range(10):
            assert lib.probe_stream(wgs, lds, ctypes.c_void_p(src.data_ptr()), vec_per_wg, nvec, ctypes.c_void_p(out.data_ptr()), cur()) == 0
    gs = graph_of(side, side_chain)
    tm, tsd = timed(gs)
    print('side: 10 kernels of %4d workgroups x %3d KB LDS streaming %d MB each: main chain %.0f us (x%.2f), side chain %.0f us (%.1f TB/s)' % (
        wgs, lds >> 10, mb_per_wg, tm, tm / t0, tsd, 10 * wgs * mb_per_wg * (1 << 20) / (tsd * 1e-6) / 1e12))

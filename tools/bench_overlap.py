"""Do two independent chains of persistent kernels overlap when a HIP graph carries them on two branches?
chain A: dgrad-like projections, chain B: weight gradients (+ reduce); timed as one stream vs two forked streams."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd import ops, _lib
from lintransunet_amd.ops import _p, _s, _ptr_array

M, K, N = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (114816, 128, 256)))
L = 8
x = torch.randn(M, K, device='cuda').bfloat16()
g = torch.randn(M, N, device='cuda').bfloat16()
w = (torch.randn(N, K, device='cuda') * 0.05)
wb = w.bfloat16()
b = torch.zeros(N, device='cuda')
y = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
dw, db = torch.zeros(N, K, device='cuda'), torch.zeros(N, device='cuda')
ws = torch.empty(_lib.load().ltu_wgrad_ws_floats(M, N, K), device='cuda')

def chain_a():
    for _ in range(L):
        _lib.call('ltu_linear_fwd', _p(x), K, _ptr_array([wb]), 1, _ptr_array([b]), _p(y), N, M, N, K, 0, 1, _s())
def chain_b():
    for _ in range(L):
        _lib.call('ltu_linear_wgrad', _p(g), N, _p(x), K, _ptr_array([dw]), _ptr_array([db]), 1, M, N, K, _p(ws), 0, 1, _s())

def timed_graph(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        gr.replay()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3

side = torch.cuda.Stream()
def both_serial():
    chain_a(); chain_b()
def both_forked():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        chain_b()
    chain_a()
    cur.wait_stream(side)
def interleaved_forked():            # one fork/join per pair, as a per-layer overlap would look
    cur = torch.cuda.current_stream()
    for _ in range(L):
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            _lib.call('ltu_linear_wgrad', _p(g), N, _p(x), K, _ptr_array([dw]), _ptr_array([db]), 1, M, N, K, _p(ws), 0, 1, side.cuda_stream)
        _lib.call('ltu_linear_fwd', _p(x), K, _ptr_array([wb]), 1, _ptr_array([b]), _p(y), N, M, N, K, 0, 1, _s())
        cur.wait_stream(side)

def one_way():                        # side waits on main before each of its launches; main never waits until the final join
    cur = torch.cuda.current_stream()
    for _ in range(L):
        _lib.call('ltu_linear_fwd', _p(x), K, _ptr_array([wb]), 1, _ptr_array([b]), _p(y), N, M, N, K, 0, 1, _s())
        side.wait_stream(cur)
        _lib.call('ltu_linear_wgrad', _p(g), N, _p(x), K, _ptr_array([dw]), _ptr_array([db]), 1, M, N, K, _p(ws), 0, 1, side.cuda_stream)
    cur.wait_stream(side)

ta, tb = timed_graph(chain_a), timed_graph(chain_b)
tow = timed_graph(one_way)
print(f'one-way dependencies (side waits on main per launch, single join): {tow:.1f} us')
ts, tf, ti = timed_graph(both_serial), timed_graph(both_forked), timed_graph(interleaved_forked)
print(f'M={M} K={K} N={N}, {L} launches per chain: A alone {ta:.1f} us, B alone {tb:.1f} us, one stream {ts:.1f} us, '
      f'two branches {tf:.1f} us, fork/join per pair {ti:.1f} us')

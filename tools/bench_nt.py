"""Graph-replayed micro-benchmark of ltu_linear_fwd / ltu_linear_wgrad (bf16) through the C-ABI: pure GPU time per launch.
usage: bench_nt.py [M K N]...   (env LTU_NT_VARIANT / LTU_NT_DBG select kernel variants / ablations)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd import ops, _lib
from lintransunet_amd.ops import _p, _ptr_array, _s

REP = 20

def timed(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REP):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / (5 * REP) * 1e3

def run(M, K, N):
    # rotate over several buffers so that the input is not L2/MALL resident from the previous launch
    nb = max(2, min(8, int(600e6 // (M * (K + N) * 2))))
    xs = [torch.randn(M, K, device='cuda').bfloat16() for _ in range(nb)]
    ys = [torch.empty(M, N, device='cuda', dtype=torch.bfloat16) for _ in range(nb)]
    w = torch.randn(N, K, device='cuda').mul_(0.05).bfloat16()
    b = torch.zeros(N, device='cuda')
    cnt = [0]
    def fwd():
        i = cnt[0] % nb; cnt[0] += 1
        _lib.call('ltu_linear_fwd', _p(xs[i]), K, _ptr_array([w]), 1, _ptr_array([b]), _p(ys[i]), N, M, N, K, 0, 1, _s())
    t = timed(fwd)
    byts = (M * K + M * N + N * K) * 2
    dw = torch.zeros(N, K, device='cuda'); db = torch.zeros(N, device='cuda')
    nws = _lib.load().ltu_wgrad_ws_floats(M, N, K)
    ws = torch.empty(nws, device='cuda')
    def wg():
        i = cnt[0] % nb; cnt[0] += 1
        _lib.call('ltu_linear_wgrad', _p(ys[i]), N, _p(xs[i]), K, _ptr_array([dw]), _ptr_array([db]), 1, M, N, K, _p(ws), ws.numel(), 0, 1, _s())
    tw = timed(wg)
    print(f'M={M:7d} K={K:4d} N={N:4d}: fwd {t:7.1f} us  {byts / t / 1e6:5.2f} TB/s {2 * M * K * N / t / 1e6:6.0f} TF | '
          f'wgrad(+reduce) {tw:7.1f} us {byts / tw / 1e6:5.2f} TB/s', flush=True)

if __name__ == "__main__":
    shapes = [(114816, 128, 128), (114816, 128, 256), (114816, 256, 128), (114816, 128, 384), (114816, 384, 128),
              (21504, 256, 256), (21504, 256, 512), (21504, 512, 256), (21504, 256, 768), (8640, 256, 512), (1024, 256, 512)]
    if len(sys.argv) > 3:
        a = list(map(int, sys.argv[1:]))
        shapes = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)]
    print('variant', os.environ.get('LTU_NT_VARIANT', '0'), 'dbg', os.environ.get('LTU_NT_DBG', '0'))
    for s in shapes:
        run(*s)


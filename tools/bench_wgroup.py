"""Graph-replayed micro-benchmark of the grouped projection weight gradient of one transformer layer (qkv, out, ffn1, ffn2) per
level.  usage: bench_wgroup.py [KNOB=value ...]"""
import os, sys, torch
for a in sys.argv[1:]:
    k, v = a.split('=')
    os.environ[k] = v
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd import ops
from bench_nt import timed


def run(M, d, layers=1):
    shapes = [(3 * d, d, 3), (d, d, 1), (2 * d, d, 1), (d, 2 * d, 1)]
    jobs = []
    for _ in range(layers):
        for N, K, nw in shapes:
            g = torch.randn(M, N, device='cuda').mul_(0.1).bfloat16()
            x = torch.randn(M, K, device='cuda').bfloat16()
            dws = [torch.zeros(N // nw, K, device='cuda') for _ in range(nw)]
            dbs = [torch.zeros(N // nw, device='cuda') for _ in range(nw)]
            jobs.append((g, x, dws, dbs, M, N, K))
    lc = ops.Context()

    def f():
        lc.wg_group = list(jobs)
        lc.wgrad_group_flush()
    t = timed(f)
    gf = sum(2.0 * M * N * K for _, _, _, _, M, N, K in jobs) / 1e9
    print(f'M={M:7d} d={d} layers={layers}: {t:6.1f} us ({gf * 1e3 / t:.0f} TFLOP/s)', flush=True)


CASES = [(114816, 128, 1), (21504, 256, 1), (8640, 256, 1), (8640, 256, 2), (8640, 256, 4), (1024, 256, 1), (1024, 256, 8)]
sel = os.environ.get('WG_CASES')
for i, (M, d, layers) in enumerate(CASES):
    if sel is None or str(i) in sel.split(','):
        run(M, d, layers)

"""What does folding per-workgroup partial sums INSIDE the producing launch (last arriver, csrc/misc.hip) save against the second
small launch the step uses?  Graph-replayed, 20 instances back to back, per instance: two-stage (partials kernel + fold kernel) vs
one launch with the last-arriver fold.  The producing kernel is the self-test's (uneven load, a few KB of input per workgroup)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd import _lib
from lintransunet_amd.ops import _p, _s
from bench_nt import timed

for nwg, n, skew in ((64, 32, 1), (256, 32, 1), (256, 128, 3), (512, 128, 5), (1024, 32, 3), (1024, 256, 5)):
    rpc = 4
    rows = sum(1 + (7 * i) % skew for i in range(nwg)) * rpc
    x = torch.randn(rows, n, device='cuda')
    part = torch.zeros(nwg * n, device='cuda')
    out = torch.empty(n, device='cuda')
    sink = torch.zeros(nwg, device='cuda')
    counter = torch.zeros(1, device='cuda', dtype=torch.int32)
    t = [timed(lambda m=m: _lib.call('ltu_selftest_last_arriver', _p(x), _p(part), _p(out), _p(counter), _p(sink), nwg, n, rpc, skew, m, _s()))
         for m in (0, 1)]
    print(f'{nwg:5d} workgroups x {n:3d} sums: two launches {t[0]:6.2f} us, last-arriver fold in one launch {t[1]:6.2f} us ({t[0] - t[1]:+.2f})', flush=True)

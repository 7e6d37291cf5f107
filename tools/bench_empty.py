"""Per-node cost of a graph replay: an (almost) empty kernel, and a small streaming kernel, timed like the other micro-benchmarks."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd import _lib
from lintransunet_amd.ops import _p, _s
from bench_nt import timed
for n in (8, 1 << 16, 1 << 20, 1 << 22, 1 << 24):
    u = torch.randn(n, device='cuda').bfloat16(); h = torch.empty_like(u)
    t = timed(lambda: _lib.call('ltu_gelu_dropout_fwd', _p(u), _p(h), n, 0.0, 1, 0, 1, _s()))
    print(f'gelu n={n:9d} ({2 * n * 2 / 1e6:8.2f} MB): {t:6.2f} us', flush=True)

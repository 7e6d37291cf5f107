"""Graph-replayed micro-benchmark of the attention gates' 1x1x1 convolutions (ltu_linear_fwd at K, N <= 128), bf16.
LTU_NO_PW_SMALL=1 selects the implicit GEMM they used before."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lintransunet_amd import _lib
from lintransunet_amd.ops import _p, _ptr_array, _s
from bench_nt import timed
for M, K, N in [(1 << 20, 16, 16), (1 << 20, 32, 16), (1 << 20, 16, 32), (1 << 18, 32, 32), (1 << 18, 64, 32), (1 << 18, 32, 64), (32768, 64, 64), (32768, 128, 64)]:
    nb = 4
    xs = [torch.randn(M, K, device='cuda').bfloat16() for _ in range(nb)]
    ys = [torch.empty(M, N, device='cuda', dtype=torch.bfloat16) for _ in range(nb)]
    w = (torch.randn(N, K, device='cuda') * 0.1).bfloat16()
    b = torch.zeros(N, device='cuda')
    cnt = [0]
    def f():
        i = cnt[0] % nb; cnt[0] += 1
        _lib.call('ltu_linear_fwd', _p(xs[i]), K, _ptr_array([w]), 1, _ptr_array([b]), _p(ys[i]), N, M, N, K, 0, 1, _s())
    t = timed(f)
    print(f'M={M} K={K} N={N}: {t:6.1f} us  {M * (K + N) * 2 / t * 1e-6:5.2f} TB/s', flush=True)

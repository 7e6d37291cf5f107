"""Graph-replayed micro-benchmark of the un-embedding's data gradient (ltu_upconv_dgrad), bf16, at the three bridge shapes.
LTU_NO_UPDGRAD_RING=1 selects the 64-tap gather implicit GEMM it replaced."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lintransunet_amd import ops, _lib
from lintransunet_amd.ops import _p, _s
from bench_nt import timed
for B, H, W, D, Ci, Co in [(2, 39, 23, 64, 128, 32), (2, 24, 14, 32, 256, 64), (2, 15, 9, 32, 256, 128)]:
    w = (torch.randn(Co, Ci, 3, 3, 3, device='cuda') * 0.05)
    prep = ops.upconv_prep(w, torch.bfloat16)
    g = torch.randn(B, 2 * H, 2 * W, 2 * D, Co, device='cuda').bfloat16()
    dx = torch.empty(B, H, W, D, Ci, device='cuda', dtype=torch.bfloat16)
    nws = _lib.load().ltu_igemm_ws_floats(B * H * W * D, Ci, 64 * Co)
    ws = torch.empty(max(nws, 1), device='cuda')
    f = lambda: _lib.call('ltu_upconv_dgrad', _p(g), _p(prep.wd), _p(dx), B, H, W, D, Ci, Co, _p(ws) if nws else 0, nws, 1, _s())
    t = timed(f)
    fl = 2.0 * B * H * W * D * 64 * Ci * Co
    print(f'upconv dgrad B={B} {H}x{W}x{D} Ci={Ci} Co={Co}: {t:7.1f} us ({fl / t / 1e6:.0f} TF)', flush=True)

"""Graph-replayed micro-benchmark of the un-embedding's weight gradient (ltu_upconv_wgrad: kernel + fold), bf16, at the three bridge
shapes, at the side stream's width (128 workgroups) and the stand-alone default.  LTU_UPW_RING=0 selects the first generation."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lintransunet_amd import ops, _lib
from lintransunet_amd.ops import _p, _s, _n
from bench_nt import timed
for B, H, W, D, Ci, Co in [(2, 39, 23, 64, 128, 32), (2, 24, 14, 32, 256, 64), (2, 15, 9, 32, 256, 128)]:
    g = torch.randn(B, 2 * H, 2 * W, 2 * D, Co, device='cuda').bfloat16()
    x = torch.randn(B, H, W, D, Ci, device='cuda').bfloat16()
    dweff = torch.zeros(8, Co, 8, Ci, device='cuda')
    dw, db = torch.zeros(Co, Ci, 3, 3, 3, device='cuda'), torch.zeros(Co, device='cuda')
    out = []
    for blocks in (128, 0):
        ws = torch.empty(_lib.load().ltu_upconv_wgrad_ws_floats(B * H * W * D, Co, Ci, blocks), device='cuda')
        f = lambda: _lib.call('ltu_upconv_wgrad', _p(g), _p(x), _p(dweff), _p(db), _p(dw), Co, Ci, _p(ws), _n(ws), blocks, B, H, W, D, Ci, Co, 1, _s())
        out.append(timed(f))
    fl = 2.0 * B * H * W * D * 64 * Ci * Co
    print(f'upconv wgrad B={B} {H}x{W}x{D} Ci={Ci} Co={Co}: 128 workgroups {out[0]:7.1f} us ({fl / out[0] / 1e6:.0f} TF)   default width {out[1]:7.1f} us', flush=True)

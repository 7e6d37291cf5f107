"""Graph-replayed micro-benchmark of the positional depthwise conv kernels (bf16).  usage: bench_dwconv.py [B H W D C]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd import _lib
from lintransunet_amd.ops import _p, _s
from bench_nt import timed

def run(B, H, W, D, C, p=float(os.environ.get("DW_P", "0.3"))):
    x = torch.randn(B, H, W, D, C, device='cuda').bfloat16()
    g = torch.randn(B, H, W, D, C, device='cuda').bfloat16()
    y = torch.empty_like(x)
    w = torch.randn(C, 27, device='cuda'); b = torch.randn(C, device='cuda')
    dw = torch.zeros(C, 27, device='cuda'); db = torch.zeros(C, device='cuda')
    f = lambda: _lib.call('ltu_dwconv_fwd', _p(x), _p(w), _p(b), _p(y), B, H, W, D, C, p, 1, 0, 1, _s())
    bw = lambda: _lib.call('ltu_dwconv_bwd', _p(g), 0, _p(x), _p(w), _p(y), _p(dw), _p(db), 0, 0, B, H, W, D, C, p, 1, 0, 1, _s())
    tf, tb = timed(f), timed(bw)
    mb = x.numel() * 2 / 1e6
    print(f'B={B} {H}x{W}x{D} C={C} ({mb:.1f} MB): fwd {tf:.1f} us  bwd(data+weight) {tb:.1f} us', flush=True)

# token counts of the four ROI transformers of the 128^3 step (57 408 / 10 752 / 4 320 / 512 per sample)
shapes = [(2, 52, 69, 16, 128), (2, 32, 21, 16, 256), (2, 20, 27, 8, 256), (2, 8, 8, 8, 256)]
if len(sys.argv) > 5:
    shapes = [tuple(map(int, sys.argv[1:6]))]
for s_ in shapes:
    run(*s_)

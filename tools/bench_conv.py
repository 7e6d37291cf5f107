"""Graph-replayed micro-benchmark of the 3x3x3 conv kernels (bf16) through the C-ABI.  usage: bench_conv.py [B H W D C0 C1 Co]..."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd import ops, _lib
from lintransunet_amd.ops import _p, _s
from bench_nt import timed

def run(B, H, W, D, C0, C1, Co):
    x0 = torch.randn(B, H, W, D, C0, device='cuda').bfloat16()
    x1 = torch.randn(B, H, W, D, C1, device='cuda').bfloat16() if C1 else None
    C = C0 + C1
    wf = (torch.randn(Co, 27, C, device='cuda') * 0.05).bfloat16()
    wd = (torch.randn(C, 27, Co, device='cuda') * 0.05).bfloat16()
    bias = torch.zeros(Co, device='cuda')
    y = torch.empty(B, H, W, D, Co, device='cuda', dtype=torch.bfloat16)
    g = torch.randn(B, H, W, D, Co, device='cuda').bfloat16()
    dx0 = torch.empty_like(x0); dx1 = torch.empty_like(x1) if C1 else None
    dw = torch.zeros(Co, C, 27, device='cuda'); db = torch.zeros(Co, device='cuda')
    ws = torch.empty(_lib.load().ltu_wgrad_ws_floats(B * H * W * D, Co, 27 * C), device='cuda')
    nosplit = os.environ.get('NO_SPLIT') is not None
    nf, nd = _lib.load().ltu_conv3d_ws_floats(B, H, W, D, C, Co), _lib.load().ltu_conv3d_ws_floats(B, H, W, D, Co, C)
    WSF = torch.empty(nf, device='cuda') if nf and not nosplit else None
    WSD = torch.empty(nd, device='cuda') if nd and not nosplit else None
    f = lambda: _lib.call('ltu_conv3d_fwd', _p(x0), _p(x1), _p(wf), _p(bias), _p(y), B, H, W, D, C0, C1, Co, 1, 1, 1, 0, _p(WSF), nf if WSF is not None else 0, 1, _s())
    dg = lambda: _lib.call('ltu_conv3d_dgrad', _p(g), _p(wd), _p(dx0), _p(dx1), B, H, W, D, C0, C1, Co, 1, 1, 1, _p(WSD), nd if WSD is not None else 0, 1, _s())
    wg = lambda: _lib.call('ltu_conv3d_wgrad', _p(g), _p(x0), _p(x1), _p(dw), _p(db), B, H, W, D, C0, C1, Co, 1, 1, 1, 0, Co, C, _p(ws), ws.numel(), 1, _s())
    tf, td, tw = timed(f), timed(dg), timed(wg)
    vox = B * H * W * D
    mb_f = vox * (C + Co) * 2 / 1e6
    print(f'conv B={B} {H}x{W}x{D} C={C0}+{C1} Co={Co}: fwd {tf:6.1f} us ({mb_f / tf:.2f} TB/s)  dgrad {td:6.1f} us  wgrad {tw:6.1f} us', flush=True)

shapes = [(2, 16, 16, 64, 128, 0, 64), (2, 8, 8, 64, 256, 0, 128), (2, 8, 8, 64, 128, 128, 128), (2, 8, 8, 64, 256, 0, 8), (2, 64, 64, 128, 16, 16, 16), (2, 64, 64, 128, 16, 0, 16), (2, 64, 64, 128, 8, 0, 16), (2, 32, 32, 128, 32, 32, 32), (2, 32, 32, 128, 32, 0, 32),
          (2, 16, 16, 64, 64, 64, 64), (2, 8, 8, 32, 128, 128, 128), (2, 4, 4, 16, 256, 0, 256)]
if len(sys.argv) > 7:
    a = list(map(int, sys.argv[1:]))
    shapes = [tuple(a[i:i + 7]) for i in range(0, len(a), 7)]
for s_ in shapes:
    run(*s_)

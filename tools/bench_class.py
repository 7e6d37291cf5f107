"""Graph-replayed micro-benchmark of the sub-pixel un-embedding (class convolution) kernels, bf16."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd import ops
from bench_nt import timed

def run(B, H, W, D, Ci, Co):
    x = torch.randn(B, H, W, D, Ci, device='cuda').bfloat16().requires_grad_(True)
    w = (torch.randn(Co, Ci, 3, 3, 3, device='cuda') * 0.05).requires_grad_(True)
    b = torch.zeros(Co, device='cuda', requires_grad=True)
    prep = ops.upconv_prep(w, torch.bfloat16)
    go = torch.randn(B, 2 * H, 2 * W, 2 * D, Co, device='cuda').bfloat16()
    def f():
        with torch.no_grad():
            ops._UpConv3d.apply(x, w, b, prep)
    tf = timed(f)
    fl = 2.0 * B * H * W * D * 64 * Ci * Co
    print(f'upconv B={B} {H}x{W}x{D} Ci={Ci} Co={Co}: fwd {tf:7.1f} us ({fl / tf / 1e6:.0f} TF)', flush=True)

shapes = [(2, 50, 33, 20, 128, 32), (2, 33, 20, 13, 256, 64), (2, 20, 13, 8, 256, 128), (2, 8, 8, 8, 256, 256)]
if len(sys.argv) > 6:
    a = list(map(int, sys.argv[1:]))
    shapes = [tuple(a[i:i + 6]) for i in range(0, len(a), 6)]
for s_ in shapes:
    run(*s_)

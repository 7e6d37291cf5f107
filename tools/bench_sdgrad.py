"""Graph-replayed micro-benchmark of the data gradient of the stride-2 convs (ltu_conv3d_dgrad), bf16, at the shapes of the step.
LTU_NO_SDGRAD_RING=1 selects the first-generation class kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lintransunet_amd import ops, _lib
from lintransunet_amd.ops import _p, _s
from bench_nt import timed

def run(B, Hl, Wl, Dl, Ci, Co, sd):
    Ho, Wo, Do = (Hl - 1) // 2 + 1, (Wl - 1) // 2 + 1, (Dl - 1) // sd + 1
    g = torch.randn(B, Ho, Wo, Do, Co, device='cuda').bfloat16()
    w = torch.randn(Co, Ci, 3, 3, 3, device='cuda') * 0.05
    wd = torch.empty(Ci, 27, Co, device='cuda', dtype=torch.bfloat16)
    _lib.call('ltu_pack_conv_weight', _p(w), 0, _p(wd), Co, Ci, Co, Ci, 1, _s())
    dx = torch.empty(B, Hl, Wl, Dl, Ci, device='cuda', dtype=torch.bfloat16)
    def f():
        _lib.call('ltu_conv3d_dgrad', _p(g), _p(wd), _p(dx), 0, B, Hl, Wl, Dl, Ci, 0, Co, 2, 2, sd, 0, 0, 1, _s())
    t = timed(f)
    fl = 2.0 * B * Ho * Wo * Do * 27 * Ci * Co
    print(f'dgrad B={B} fine {Hl}x{Wl}x{Dl} Ci={Ci} <- Co={Co} stride (2,2,{sd}): {t:7.1f} us ({fl / t / 1e6:.0f} TF)', flush=True)

# ROI embeds (L1, L2, L3), encoder blk1.conv2 / blk3.conv2 (2,2,2), blk0.conv2 / blk2.conv2 (2,2,1)
for s_ in [(2, 78, 46, 128, 32, 128, 2), (2, 48, 28, 64, 64, 256, 2), (2, 30, 18, 64, 128, 256, 2), (2, 32, 32, 128, 32, 64, 2), (2, 8, 8, 64, 128, 256, 2),
           (2, 64, 64, 128, 16, 32, 1), (2, 16, 16, 64, 64, 128, 1)]:
    run(*s_)

"""Diagnostic (GPU box): op-by-op relative L2 difference between the bf16 and the fp32 forward of the same model / input.
    python tools/diag_trace.py H W D [batch] [seed]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_model import build
from oracle import net as O_net, seedgen
from lintransunet_amd import ops

H, W, D = (int(v) for v in sys.argv[1:4])
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 1
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 800
cfg = O_net.NetConfig()
NAMES = ['window_embed', 'conv3d', 'conv3d_pair', 'upconv3d', 'linear', 'linear_gelu', 'instnorm_act', 'res_layernorm', 'linear_attention',
         'pos_conv', 'trilinear_up', 'roi_warp', 'roi_unwarp', 'head_softmax', 'final_softmax', 'attention_gate']


def trace(dtype):
    rec = []
    orig = {n: getattr(ops, n) for n in NAMES}

    def wrap(n, f):
        def g(*a, **k):
            y = f(*a, **k)
            outs = y if isinstance(y, tuple) else (y,)
            for i, o in enumerate(outs[:1] if n == 'res_layernorm' else outs):
                rec.append((f'{n}[{i}]', tuple(o.shape), o.detach().float().cpu()))
            return y
        return g
    for n, f in orig.items():
        setattr(ops, n, wrap(n, f))
    try:
        m = build(cfg, seed, dtype)
        x = seedgen.seeded_volume((batch, 1, H, W, D), seed + 1).cuda()
        with torch.no_grad():
            m(x)
        boxes = [b.cpu() for b in m.last_boxes]
    finally:
        for n, f in orig.items():
            setattr(ops, n, f)
    return rec, boxes


a, ba = trace(torch.float32)
b, bb = trace(torch.bfloat16)
print('boxes fp32', [t.tolist() for t in ba])
print('boxes bf16', [t.tolist() for t in bb])
for i, ((n, s, ta), (_, _, tb)) in enumerate(zip(a, b)):
    C = min(ta.shape[-1], tb.shape[-1])
    ta, tb = ta[..., :C].double(), tb[..., :C].double()
    e = ((ta - tb).norm() / ta.norm().clamp_min(1e-30)).item()
    flag = '  <<<<' if e > 5e-2 else ''
    print(f'{i:4d} {n:22s} {str(s):28s} rel-L2 {e:.3e}{flag}')

"""Diagnostic (GPU box): is the training step reproducible bit for bit, and if not, where does the difference enter?
Runs the small reference configuration (32^3 x 2, bf16 storage) several times with identical inputs and weights, records a checksum
of EVERY activation gradient through autograd hooks on the outputs of the ops layer, and reports the first one (in backward order)
that differs from the first run.  This is how the fp32 atomics of the level-loss sums were found in round 2 (the loss gradient
itself differed in the last bit; bf16 rounding downstream amplified it to 1e-2 of some parameter gradients).
    python tools/diag_repro.py [runs]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from lintransunet_amd import train, ops
from oracle import net as O_net, seedgen, step as O_step
import test_gpu_model as T

RUNS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = O_net.NetConfig()
x = seedgen.seeded_volume((2, 1, 32, 32, 32), 81).to('cuda')
label = seedgen.seeded_label((2, 1, 32, 32, 32), 82).to('cuda')
w = O_step.dynamic_weights(0)
NAMES = ['conv3d', 'conv3d_pair', 'upconv3d', 'linear', 'linear_gelu', 'instnorm_act', 'res_layernorm', 'linear_attention', 'layer_tail',
         'pos_conv', 'trilinear_up', 'roi_warp', 'roi_unwarp', 'head_softmax', 'final_softmax', 'attention_gate']


def run():
    rec = []
    orig = {n: getattr(ops, n) for n in NAMES if hasattr(ops, n)}
    cnt = [0]

    def wrap(n, f):
        def g(*a, **k):
            y = f(*a, **k)
            idx = cnt[0]
            cnt[0] += 1
            for i, o in enumerate(y if isinstance(y, tuple) else (y,)):
                if torch.is_tensor(o) and o.requires_grad:
                    def hook(gr, tag=f'{idx:4d} {n}[{i}] {tuple(o.shape)}'):
                        gd = gr.double()
                        rec.append((tag, gd.sum().item(), gd.abs().sum().item()))
                    o.register_hook(hook)
            return y
        return g
    for n, f in orig.items():
        setattr(ops, n, wrap(n, f))
    try:
        torch.manual_seed(99)
        m = T.build(cfg, 300, torch.bfloat16, dropout=0.0)
        t, _ = train.train_step(m, x, label, w)
        torch.cuda.synchronize()
    finally:
        for n, f in orig.items():
            setattr(ops, n, f)
    return rec, {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}


r0, g0 = run()
for i in range(RUNS):
    r, g = run()
    diff = [(a, b) for a, b in zip(r0, r) if a[1:] != b[1:]]
    bad = [k for k in g0 if not torch.equal(g[k], g0[k])]
    print(f'run {i}: {len(r)} activation gradients, {len(diff)} differ; {len(bad)} of {len(g0)} parameter gradients differ')
    if diff:
        pos = r0.index(diff[0][0])
        print(f'   first differing activation gradient (backward position {pos}): {diff[0][0][0]}  {diff[0][0][1:]} vs {diff[0][1][1:]}')
    if bad:
        print('   parameter gradients:', bad[:6], '...' if len(bad) > 6 else '')

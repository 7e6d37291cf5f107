"""Diagnostic: gradients of the chain-kernel path and of the op-by-op path (both bf16) against the fp32-storage run with the same
dropout masks: if both sit at the same distance from fp32, their mutual difference is bf16 noise."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_model import build, exact_zero_grad
from oracle import net as O_net, seedgen, step as O_step
from lintransunet_amd import train, ops
cfg = O_net.NetConfig()
x = seedgen.seeded_volume((2, 1, 32, 32, 32), 81).cuda(); lab = seedgen.seeded_label((2, 1, 32, 32, 32), 82).cuda()
w = O_step.dynamic_weights(0)
def run(dtype, tail, dropout):
    ops.USE_LAYER_TAIL = tail
    torch.manual_seed(99)
    m = build(cfg, 300, dtype, dropout=dropout)
    t, _ = train.train_step(m, x, lab, w)
    torch.cuda.synchronize()
    return sum(v.item() for v in t), {k: p.grad.double() for k, p in m.named_parameters() if p.grad is not None and not exact_zero_grad(k)}
def dist(a, b):
    per = sorted(((a[k] - b[k]).norm() / b[k].norm().clamp_min(1e-12)).item() for k in b)
    num = sum(((a[k] - b[k]) ** 2).sum().item() for k in b) ** 0.5; den = sum((b[k] ** 2).sum().item() for k in b) ** 0.5
    return f'overall {num / den:.2e} median {per[len(per) // 2]:.2e} worst {per[-1]:.2e}'
for dropout in (0.0, 0.3):
    lf, gf = run(torch.float32, False, dropout)
    lt, gt = run(torch.bfloat16, True, dropout)
    lo, go = run(torch.bfloat16, False, dropout)
    print(f'dropout {dropout}: loss fp32 {lf:.6f} tail {lt:.6f} op-by-op {lo:.6f}')
    print('   tail vs fp32     ', dist(gt, gf))
    print('   op-by-op vs fp32 ', dist(go, gf))
    print('   tail vs op-by-op ', dist(gt, go))

"""Diagnostic: run-to-run and fused-vs-autograd gradient differences on the small model (GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_model import build, SMALL, exact_zero_grad
from oracle import net as O_net, seedgen, step as O_step
from lintransunet_amd import train
cfg = O_net.NetConfig(**SMALL)
x = seedgen.seeded_volume((2, 1, 32, 32, 32), 11).cuda(); lab = seedgen.seeded_label((2, 1, 32, 32, 32), 12).cuda()
w = O_step.dynamic_weights(0)
def grads(fused, steps=1):
    m = build(cfg, 100)
    red = train.GradReducer(m, bucket_mb=0.5, unused=train.UNUSED_PARAMETERS) if fused else None
    for _ in range(steps):
        if red: red.zero_grad()
        else:
            for p in m.parameters(): p.grad = None
        train.train_step(m, x, lab, w, reducer=red)
    torch.cuda.synchronize()
    return {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}, [b.clone() for b in m.last_boxes]
def cmp(a, b, tag):
    rows = []
    for k in a:
        if exact_zero_grad(k): continue
        d = (a[k] - b[k]).abs().max().item() / max(b[k].abs().max().item(), 1e-3)
        rows.append((d, k))
    rows.sort(reverse=True)
    print(tag, ['%.1e %s' % r for r in rows[:4]])
r1, b1 = grads(False); r2, b2 = grads(False); f1, b3 = grads(True, 1); f2, b4 = grads(True, 2)
print('boxes equal', all(torch.equal(p, q) for p, q in zip(b1, b2)), all(torch.equal(p, q) for p, q in zip(b1, b4)))
cmp(r1, r2, 'ref vs ref      '); cmp(f1, r1, 'fused(1) vs ref '); cmp(f2, r1, 'fused(2) vs ref '); cmp(f2, f1, 'fused(2) vs (1) ')

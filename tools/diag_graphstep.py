"""Diagnostic (GPU box, single process): eager vs graph-replayed step, loss and per-bucket gradient differences."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_model import build, SMALL
from oracle import net as O_net, seedgen, step as O_step
from lintransunet_amd import train
cfg = O_net.NetConfig(**SMALL)
dt = torch.bfloat16 if len(sys.argv) > 1 and sys.argv[1] == 'bf16' else torch.float32
x = seedgen.seeded_volume((1, 1, 32, 32, 32), 41).cuda(); lab = seedgen.seeded_label((1, 1, 32, 32, 32), 51).cuda()
w = O_step.dynamic_weights(0)

def eager(step_times=1, use_scale=False):
    m = build(cfg, 100, dt)
    red = train.GradReducer(m, bucket_mb=0.5, unused=train.UNUSED_PARAMETERS)
    red.zero_grad()
    ls = torch.tensor([v / step_times for v in w], device='cuda') if use_scale else None
    t, _ = train.train_step(m, x, lab, w, step_times=step_times, reducer=red, level_scale=ls)
    torch.cuda.synchronize()
    return [f.clone() for f in red.flat], [v.item() for v in t], [b.clone() for b in m.last_boxes]

def graphed(step_times=1):
    m = build(cfg, 100, dt)
    red = train.GradReducer(m, bucket_mb=0.5, unused=train.UNUSED_PARAMETERS)
    g = train.GraphedStep(m, x, lab, w, red, step_times=step_times)
    t, _ = g(x, lab, micro=0)
    torch.cuda.synchronize()
    return [f.clone() for f in red.flat], [v.item() for v in t], [b.clone() for b in m.last_boxes]

def cmp(a, b, tag):
    fa, ta, ba = a; fb, tb, bb = b
    worst = max(((p - q).norm() / q.norm().clamp_min(1e-20)).item() for p, q in zip(fa, fb))
    print(f'{tag:34s} loss {sum(ta):.6f} vs {sum(tb):.6f}  levels {["%.5f" % (p - q) for p, q in zip(ta, tb)]}  worst bucket rel-L2 {worst:.2e}  boxes equal {all(torch.equal(p, q) for p, q in zip(ba, bb))}')

e1 = eager(); e2 = eager()
cmp(e1, e2, 'eager vs eager')
cmp(eager(use_scale=True), e1, 'eager(level_scale) vs eager')
cmp(graphed(), e1, 'graph vs eager')
e3 = eager(step_times=2)
cmp(eager(step_times=2, use_scale=True), e3, 'eager(st2, level_scale) vs eager(st2)')
cmp(graphed(step_times=2), e3, 'graph(st2, micro 0) vs eager(st2)')

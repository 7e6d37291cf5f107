import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_model import build, SMALL, _flat_grads
from oracle import net as O_net, seedgen, step as O_step
from lintransunet_amd import train
cfg = O_net.NetConfig(**SMALL)
DEV = 'cuda'
xs = [seedgen.seeded_volume((1, 1, 32, 32, 32), 41 + i).to(DEV) for i in range(2)]
ls = [seedgen.seeded_label((1, 1, 32, 32, 32), 51 + i).to(DEV) for i in range(2)]
w0, w1 = O_step.dynamic_weights(0), O_step.dynamic_weights(40)
print(w0, w1)
def eager(w):
    m = build(cfg, 100)
    red = train.GradReducer(m, bucket_mb=0.5, unused=train.UNUSED_PARAMETERS)
    red.zero_grad()
    tots = []
    for j in range(2):
        t, _ = train.train_step(m, xs[j], ls[j], w, step_times=2, reducer=red, reduce=(j == 1))
        tots.append([v.item() for v in t])
    torch.cuda.synchronize()
    return [f.clone() for f in red.flat], tots
m = build(cfg, 100)
red = train.GradReducer(m, bucket_mb=0.5, unused=train.UNUSED_PARAMETERS)
g = train.GraphedStep(m, xs[0], ls[0], w0, red, step_times=2)
for w in (w0, w1, w0):
    g.set_weights(w)
    tots = []
    for j in range(2):
        t, _ = g(xs[j], ls[j], micro=j)
        tots.append([v.item() for v in t])
    torch.cuda.synchronize()
    got = [f.clone() for f in red.flat]
    ref, rt = eager(w)
    worst = max(((p - q).norm() / q.norm().clamp_min(1e-20)).item() for p, q in zip(got, ref))
    for j in range(2):
        print('micro', j, ['%.6f' % v for v in tots[j]], 'ref', ['%.6f' % v for v in rt[j]])
    print('worst bucket rel-L2', worst)
print('--- single graph (step_times=1), weights changing')
m = build(cfg, 100)
red = train.GradReducer(m, bucket_mb=0.5, unused=train.UNUSED_PARAMETERS)
g = train.GraphedStep(m, xs[0], ls[0], w0, red, step_times=1)
def eager1(w, j):
    mm = build(cfg, 100)
    rr = train.GradReducer(mm, bucket_mb=0.5, unused=train.UNUSED_PARAMETERS)
    rr.zero_grad()
    t, _ = train.train_step(mm, xs[j], ls[j], w, reducer=rr)
    torch.cuda.synchronize()
    return [f.clone() for f in rr.flat], [v.item() for v in t]
for w, j in ((w0, 0), (w0, 1), (w1, 0), (w0, 0)):
    g.set_weights(w)
    t, _ = g(xs[j], ls[j])
    torch.cuda.synchronize()
    got = [f.clone() for f in red.flat]
    ref, rt = eager1(w, j)
    worst = max(((p - q).norm() / q.norm().clamp_min(1e-20)).item() for p, q in zip(got, ref))
    print(['%.6f' % v.item() for v in t], 'ref', ['%.6f' % v for v in rt], 'worst', worst)

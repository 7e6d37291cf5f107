import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_model import build, SMALL
from oracle import net as O_net, seedgen, step as O_step
from lintransunet_amd import train, ops
cfg = O_net.NetConfig(**SMALL)
DEV = 'cuda'
xs = [seedgen.seeded_volume((1, 1, 32, 32, 32), 41 + i).to(DEV) for i in range(2)]
ls = [seedgen.seeded_label((1, 1, 32, 32, 32), 51 + i).to(DEV) for i in range(2)]
w0 = O_step.dynamic_weights(0)
stash = {}
orig = train.deep_supervision_loss
def wrapped(predict, masks, *a, **k):
    stash['predict'], stash['masks'] = predict, masks
    return orig(predict, masks, *a, **k)
train.deep_supervision_loss = wrapped
NAMES = ['conv3d', 'conv3d_pair', 'upconv3d', 'instnorm_act', 'linear_attention', 'pos_conv', 'trilinear_up', 'roi_warp', 'roi_unwarp',
         'head_softmax', 'final_softmax', 'attention_gate']
rec = []
origs = {n: getattr(ops, n) for n in NAMES}
def wrap(n, f):
    def g(*a, **k):
        y = f(*a, **k)
        for i, o in enumerate(y if isinstance(y, tuple) else (y,)):
            rec.append((f'{n}[{i}]', o))
        return y
    return g
for n, f in origs.items():
    setattr(ops, n, wrap(n, f))

def snap():
    return [(n, t.detach().float().clone()) for n, t in rec]

def eager(j):
    rec.clear()
    mm = build(cfg, 100)
    rr = train.GradReducer(mm, bucket_mb=0.5, unused=train.UNUSED_PARAMETERS)
    rr.zero_grad()
    t, _ = train.train_step(mm, xs[j], ls[j], w0, reducer=rr)
    torch.cuda.synchronize()
    return snap(), [b.clone() for b in mm.last_boxes], [v.item() for v in t]

m = build(cfg, 100)
red = train.GradReducer(m, bucket_mb=0.5, unused=train.UNUSED_PARAMETERS)
rec.clear()
g = train.GraphedStep(m, xs[0], ls[0], w0, red)
# rec now holds warm-up (2x) + capture tensors: the capture's are the LAST third
n3 = len(rec) // 3
graph_rec = rec[2 * n3:]
graph_boxes = m.last_boxes
for j in (0, 1, 0):
    t, _ = g(xs[j], ls[j])
    torch.cuda.synchronize()
    got = [(n, tt.detach().float().clone()) for n, tt in graph_rec]
    gb = [b.clone() for b in graph_boxes]
    ref, rb, rt = eager(j)
    print('data', j, 'loss', ['%.6f' % v.item() for v in t], 'ref', ['%.6f' % v for v in rt], 'boxes equal', [torch.equal(a, b) for a, b in zip(gb, rb)])
    print('   graph boxes', [b.tolist() for b in gb]); print('   eager boxes', [b.tolist() for b in rb])
    shown = 0
    for i, ((n, a), (_, b)) in enumerate(zip(got, ref)):
        e = ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
        if e > 1e-4 and shown < 6:
            print(f'   first diffs: op {i} {n} {tuple(a.shape)} rel-L2 {e:.3e}')
            shown += 1

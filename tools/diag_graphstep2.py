import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd import train
from lintransunet_amd.model import get_model_dict
from oracle import seedgen, step as O_step
dev = torch.device('cuda', 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
x = seedgen.seeded_volume((B, 1, 32, 32, 32), 1).to(dev)
lab = seedgen.seeded_label((B, 1, 32, 32, 32), 2).to(dev)
w = O_step.dynamic_weights(0)
def build(dtype):
    torch.manual_seed(5)
    m = get_model_dict('MaskTransUnet')([8, 8, 8, 16, 32], [20, 12, 9, 10, 6], [False, True, True, True, True], 1, 2,
                                        dropout=0.0, act_dtype=dtype).to(dev).train()
    return m, train.GradReducer(m, bucket_mb=0.25, unused=train.UNUSED_PARAMETERS)
for dtype in (torch.float32, torch.bfloat16):
    m, red = build(dtype)
    red.zero_grad()
    t, _ = train.train_step(m, x, lab, w, reducer=red)
    torch.cuda.synchronize()
    ref = [f.clone() for f in red.flat]; rl = sum(v.item() for v in t); rb = [b.clone() for b in m.last_boxes]
    m, red = build(dtype)
    g = train.GraphedStep(m, x, lab, w, red)
    gb = m.last_boxes
    for r in range(4):
        t, _ = g(x, lab)
        torch.cuda.synchronize()
        errs = [((p - q).norm() / q.norm().clamp_min(1e-20)).item() for p, q in zip(red.flat, ref)]
        print(dtype, 'replay', r, 'loss', sum(v.item() for v in t), 'ref', rl, 'boxes eq', [torch.equal(a, b) for a, b in zip(gb, rb)], 'worst bucket', '%.2e' % max(errs))

import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd import train
from lintransunet_amd.model import get_model_dict
from oracle import seedgen, step as O_step
dev = torch.device('cuda', 0)
B = 2
x = seedgen.seeded_volume((B, 1, 32, 32, 32), 1).to(dev)
lab = seedgen.seeded_label((B, 1, 32, 32, 32), 2).to(dev)
w = O_step.dynamic_weights(0)
torch.manual_seed(5)
m = get_model_dict('MaskTransUnet')([8, 8, 8, 16, 32], [20, 12, 9, 10, 6], [False, True, True, True, True], 1, 2,
                                    dropout=0.0, act_dtype=torch.bfloat16).to(dev).train()
red = train.GradReducer(m, bucket_mb=0.25, unused=train.UNUSED_PARAMETERS)
mode = sys.argv[1] if len(sys.argv) > 1 else 'graph'
if mode == 'graph':
    g = train.GraphedStep(m, x, lab, w, red)
    step = lambda: g(x, lab)
else:
    def step():
        red.zero_grad()
        return train.train_step(m, x, lab, w, reducer=red)
step(); torch.cuda.synchronize()
ref = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
bad = {}
for r in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    step(); torch.cuda.synchronize()
    for n, p in m.named_parameters():
        if p.grad is None: continue
        e = ((p.grad - ref[n]).norm() / ref[n].norm().clamp_min(1e-20)).item()
        if e > 1e-4:
            bad.setdefault(n, []).append((r, '%.1e' % e))
for n, v in bad.items():
    print(n, tuple(dict(m.named_parameters())[n].shape), v[:6], len(v))
print('done', mode, len(bad))

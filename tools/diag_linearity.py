"""Which parameters break the gradient-linearity property of tests/test_gpu_model.py::test_full_size_properties (GPU box)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import seedgen
from oracle import step as O_step
from lintransunet_amd import train
from lintransunet_amd.model import get_model_dict
DEV = torch.device('cuda:0')
torch.manual_seed(7)
model = get_model_dict('MaskTransUnet')([16, 32, 64, 128, 256], [100, 65, 40, 25, 10], [False, True, True, True, True], 1, 2,
                                        dropout=0.0, act_dtype=torch.bfloat16).to(DEV).train()
x = seedgen.seeded_volume((2, 1, 128, 128, 128), 31).to(DEV)
label = seedgen.seeded_label((2, 1, 128, 128, 128), 32).to(DEV)
weights = O_step.dynamic_weights(0)

def step(scale):
    for p in model.parameters():
        p.grad = None
    predict, masks = model(x)
    totals, _ = train.deep_supervision_loss(predict, masks, label, weights, scale=scale)
    torch.autograd.backward(totals, [torch.ones_like(t) for t in totals])
    return {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}

g0 = step(1.0)
for name, sc in (('same scale again', 1.0), ('half scale', 0.5)):
    g1 = step(sc)
    rows = []
    from tests.test_gpu_model import exact_zero_grad
    for k in g0:
        if exact_zero_grad(k):
            continue
        n0 = g0[k].double().norm().item()
        rows.append(((g1[k].double() - sc * g0[k].double()).norm().item() / max(n0, 1e-9), k, n0))
    rows.sort(reverse=True)
    print(name)
    for r in rows[:12]:
        print('  %.3e  %-70s |g|=%.3e' % r)

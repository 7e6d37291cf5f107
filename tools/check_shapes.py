"""Robustness sweep: one bf16 training step (and one eval forward) at several patch shapes / batch sizes / label counts."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd import train
from lintransunet_amd.model import get_model_dict
from oracle import seedgen

dev = torch.device('cuda:0')
for C in (2, 3):
    model = get_model_dict('MaskTransUnet')([16, 32, 64, 128, 256], [100, 65, 40, 25, 10], [False, True, True, True, True], 1, C,
                                            dropout=0.3, act_dtype=torch.bfloat16).to(dev)
    specs = train.level_specs(5, ('CrossEntroLoss', 'DiceClassLoss', 'DiceClassLoss2'), criterion_weight=[10, 1, 2]) if C == 3 else None
    for B, size in ((1, (128, 128, 64)), (3, (96, 96, 96)), (2, (64, 64, 32)), (1, (96, 128, 48)), (2, (32, 32, 32)), (1, (160, 160, 32))):
        x = torch.randn((B, 1) + size).clamp_(-4, 4).to(dev)
        lab = seedgen.seeded_label((B, 1) + size, 5, n_classes=C).to(dev)
        model.train()
        totals, _ = train.train_step(model, x, lab, train.get_dynamic_weight(1)[0], specs=specs)
        tot = sum(t.item() for t in totals)
        ok = all(torch.isfinite(p.grad).all().item() for p in model.parameters() if p.grad is not None)
        model.eval()
        with torch.no_grad():
            oh = model(x)
        print(f'C={C} B={B} size={size}: loss {tot:.4f} grads finite {ok} eval {tuple(oh.shape)} sum-to-one {bool((oh.sum(1) == 1).all())}', flush=True)

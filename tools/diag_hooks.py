"""Diagnostic: eager hook path with a live 1-rank RCCL group vs the same step without collectives, per bucket."""
import os, sys, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd import train
from lintransunet_amd.model import get_model_dict
from oracle import seedgen, step as O_step
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
dist.init_process_group('nccl', device_id=dev)
x = seedgen.seeded_volume((2, 1, 32, 32, 32), 1).to(dev)
lab = seedgen.seeded_label((2, 1, 32, 32, 32), 2).to(dev)
w = O_step.dynamic_weights(0)
def run(collective, dtype=torch.bfloat16):
    torch.manual_seed(5)
    m = get_model_dict('MaskTransUnet')([8, 8, 8, 16, 32], [20, 12, 9, 10, 6], [False, True, True, True, True], 1, 2,
                                        dropout=0.0, act_dtype=dtype).to(dev).train()
    red = train.GradReducer(m, bucket_mb=0.25, unused=train.UNUSED_PARAMETERS)
    if collective:
        red.world = 2; red.avg = True
    red.zero_grad()
    train.train_step(m, x, lab, w, reducer=red)
    torch.cuda.synchronize()
    return [f.clone() for f in red.flat], red
for dtype in (torch.float32, torch.bfloat16):
    a, red = run(False, dtype)
    for trial in range(3):
        b, _ = run(True, dtype)
        errs = [((p - q).norm() / q.norm().clamp_min(1e-20)).item() for p, q in zip(b, a)]
        print(dtype, 'hooks+RCCL vs no collective, per bucket:', ['%.1e' % e for e in errs])
    c, _ = run(False, dtype)
    print(dtype, 'no collective twice:', ['%.1e' % ((p - q).norm() / q.norm().clamp_min(1e-20)).item() for p, q in zip(c, a)])
names = {id(p): n for n, p in red.buckets and [(n, p) for n, p in []]}
dist.destroy_process_group()

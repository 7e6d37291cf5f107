"""The CPU-baseline anchor of BASELINE.md section 4 (runs in the BUILD container only: it imports the reference from
/root/reference): the oracle's training step - what bench.py times as `cpu_baseline` on the GPU box - against the reference's own
modules replaying utils/utils_3D_embed_full.py:63-86, both fp32, dropout 0.3, 128^3, B = 1, 8 threads: the oracle has to land
within +-20 % of the reference before its GPU-box number is quoted.  1 warm-up + 2 timed steps each."""
import os, sys, time
import torch
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')
from model.trans_3DUnet import get_model_dict          # reference
from loss import criterions as R_loss                  # reference
from oracle import net as O_net, seedgen, step as O_step

torch.set_num_threads(int(sys.argv[1]) if len(sys.argv) > 1 else 8)
size = (128, 128, 128)
cfg = O_net.NetConfig(dropout=0.3)
shapes = O_net.param_shapes(cfg)
x = seedgen.seeded_volume((1, 1) + size, 8)
lab = seedgen.seeded_label((1, 1) + size, 9)
weights = O_step.dynamic_weights(0)
F = torch.nn.functional


def reference_step(model, crits):
    predict, masks = model(x)
    temp = F.max_pool3d(lab.float(), kernel_size=(2, 2, 1), stride=(2, 2, 1))
    loss_list = []
    for lvl in range(len(weights)):
        if lvl == 0:
            vals = [l(predict, lab.long()) for l in crits[-lvl - 1].values()]
        else:
            vals = [l(masks[-lvl], temp.long()) for l in crits[-lvl - 1].values()]
            k = 2 if lvl % 2 == 0 else (2, 2, 1)
            temp = F.max_pool3d(temp, kernel_size=k, stride=k)
        loss_list.append(vals)
    total = sum(sum(v) * w for v, w in zip(loss_list, weights))
    total.backward()


def timed(fn, tag):
    ts = []
    for i in range(3):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
        print(f'{tag} step {i}: {ts[-1]:.1f} s', flush=True)
    return sum(ts[1:]) / 2


P = seedgen.seeded_params(shapes, 7, requires_grad=True)
def oracle():
    for p in P.values():
        p.grad = None
    O_step.train_step(P, cfg, x, lab, weights)
t_or = timed(oracle, 'oracle')
model = get_model_dict('MaskTransUnet')(num_layers=cfg.num_layers, roi_size_list=cfg.roi_size_list, is_roi_list=cfg.is_roi_list,
                                        dim_input=1, dim_output=2, kernel_size=3, dropout=0.3)
model.load_state_dict(seedgen.seeded_params(shapes, 7), strict=True)
model.train()
crits = [R_loss.get_criterions(['CrossEntroLoss', 'BalanceDiceLoss'])] * 3 + [R_loss.get_criterions(['CrossEntroLoss', 'DiceClassLoss'])] * 2
def ref():
    model.zero_grad(set_to_none=True)
    reference_step(model, crits)
t_ref = timed(ref, 'reference')
print(f'threads {torch.get_num_threads()}: oracle {t_or:.1f} s/step ({1 / t_or:.4f} patches/s), reference {t_ref:.1f} s/step ({1 / t_ref:.4f} patches/s), '
      f'ratio {t_or / t_ref:.2f}')

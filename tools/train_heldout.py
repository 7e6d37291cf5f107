"""Trains the small configuration on synthetic ellipsoid CT patches on the GPU box (graph-replayed steps + fused AdamW, the
training loop of train3D.py in miniature) and writes the reference-loadable checkpoint gpurun_out/heldout_small.pt.
tests/golden/make_golden.py heldout then runs the REFERENCE on that checkpoint and a held-out volume to produce
tests/golden/heldout_small.npz (north_star: "Dice parity to the reference on a held-out synthetic volume").
    python tools/train_heldout.py [steps]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd import train, optim
from lintransunet_amd.model import get_model_dict
from lintransunet_amd import losses as L
from oracle import seedgen

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device('cuda', 0)
SIZE = (64, 64, 32)


def heldout_batch(batch, seed):
    """CT-like patches whose foreground (seeded ellipsoids) is brighter than the background, plus noise: learnable in a few
    hundred steps.  Pure function of the seed (CPU generators), so the golden script rebuilds the held-out volume."""
    lab = seedgen.seeded_label((batch, 1) + SIZE, seed, n_blobs=2)
    noise = seedgen.seeded_volume((batch, 1) + SIZE, seed + 1)
    x = 0.6 * noise + 1.2 * lab.float() - 0.3
    return x, lab


torch.manual_seed(2026)
model = get_model_dict('MaskTransUnet')([8, 8, 8, 16, 32], [20, 12, 9, 10, 6], [False, True, True, True, True], 1, 2,
                                        dropout=0.1, act_dtype=torch.float32).to(dev).train()
reducer = train.GradReducer(model, unused=train.UNUSED_PARAMETERS)
opt = optim.FusedAdamW(reducer, lr=2e-3, weight_decay=1e-2)
weights = train.get_dynamic_weight(800)
x0, l0 = heldout_batch(2, 1000)
step = train.GraphedStep(model, x0.to(dev), l0.to(dev), weights[0], reducer)
t0 = time.time()
for it in range(steps):
    x, lab = heldout_batch(2, 1000 + 2 * it)
    if it % 50 == 0:
        step.set_weights(weights[min(it // 10, 799)])        # "epochs" of 10 steps: exercises the per-epoch level weights
    totals, named = step(x.to(dev), lab.to(dev))
    opt.step()
    if it % 50 == 0 or it == steps - 1:
        print(f'step {it:4d} loss {sum(t.item() for t in totals):.4f} dice(level0) {named[0]["DiceClassLoss"].item():.4f} ({time.time() - t0:.0f} s)', flush=True)
os.makedirs('gpurun_out', exist_ok=True)
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
torch.save(sd, 'gpurun_out/heldout_small.pt')
# held-out volume (seed never seen in training), train-mode forward without dropout = the reference's probabilities
model.dropout = 0.0
xv, lv = heldout_batch(1, 999001)
with torch.no_grad():
    pass
predict, masks = model(xv.to(dev))
d = L.DiceClassLoss()(predict.detach(), lv.to(dev)).item()
print(f'held-out Dice loss (fp32, HIP path) {d:.6f}  -> foreground Dice {1 - d:.4f}')

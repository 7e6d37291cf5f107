"""Diagnostic: per-parameter gradient-norm error of the HIP model against a golden fixture (GPU box)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_model import run, SMALL
from oracle import net as O_net

for tag, kw, size, seed in (('small_wide', SMALL, (64, 96, 16), 200), ('full32', {}, (32, 32, 32), 300)):
    G = np.load(f'tests/golden/model_{tag}.npz')
    model, x, label, predict, masks, totals, named = run(O_net.NetConfig(**kw), size, 1, seed)
    sd = dict(model.named_parameters())
    rows = []
    for k, n in zip(G['grad_keys'], G['grad_norms']):
        got = sd[str(k)].grad.double().norm().item()
        rows.append((abs(got - n) / max(n, 1e-3), str(k), got, n))
    rows.sort(reverse=True)
    print(tag, 'total', sum(t.item() for t in totals), float(G['total']))
    for r in rows[:12]:
        print('  %.3e  %-75s got %.4e ref %.4e' % r)

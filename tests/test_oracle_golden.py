"""CPU tests: the oracle restatement against the vectors the reference itself produced
(tests/golden/make_golden.py).  No GPU, no /root/reference needed."""
import os

import numpy as np
import pytest
import torch

from oracle import losses as O_loss
from oracle import net as O_net
from oracle import roi as O_roi
from oracle import seedgen
from oracle import step as O_step

TOL = 1e-5


def _close(a, b, tol=TOL):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(np.asarray(b)).double()
    scale = max(1.0, b.abs().max().item())
    assert (a - b).abs().max().item() <= tol * scale


def test_state_dict_surface():
    shapes = O_net.param_shapes(O_net.NetConfig())
    assert len(shapes) == 614
    assert sum(int(np.prod(s)) for s in shapes.values()) == 20872836
    shapes3 = O_net.param_shapes(O_net.NetConfig(dim_output=3))
    assert sum(int(np.prod(s)) for s in shapes3.values()) == 20887532
    assert shapes['decode.bridge_list.1.transformer.down_embed.module_list.0.0.weight'] == (128, 32, 3, 3, 3)
    assert shapes['decode.bridge_list.4.transformer.pos_encoders.7.proj.weight'] == (256, 1, 3, 3, 3)


@pytest.mark.parametrize('tag', list(seedgen.LINATTN_CASES))
def test_linear_attention(golden_dir, tag):
    G = np.load(os.path.join(golden_dir, 'linattn.npz'))
    q, k, v, go = seedgen.linattn_case(tag)
    q, k, v = (t.requires_grad_(True) for t in (q, k, v))
    out = O_net.linear_attention(q, k, v)
    out.backward(go)
    _close(out.detach(), G[f'{tag}_out'])
    _close(q.grad, G[f'{tag}_dq'])
    _close(k.grad, G[f'{tag}_dk'])
    _close(v.grad, G[f'{tag}_dv'])


def test_roi_cases(golden_dir):
    G = np.load(os.path.join(golden_dir, 'roi.npz'))
    for name in G['names']:
        mask = torch.from_numpy(G[f'{name}_mask'])
        geo = O_roi.roi_geometry(int(G[f'{name}_roi_size']))
        box = O_roi.find_boxes(mask, geo['min_h'], geo['min_w'])
        _close(box, G[f'{name}_box'], 0)
        feat = torch.from_numpy(G[f'{name}_feat']).requires_grad_(True)
        roi = O_roi.warp_to_roi(feat, box, geo)
        _close(roi.detach(), G[f'{name}_roi'])
        roi.backward(torch.from_numpy(G[f'{name}_groi']))
        _close(feat.grad, G[f'{name}_dfeat'])
        roi_in = torch.from_numpy(G[f'{name}_roi_in']).requires_grad_(True)
        back = O_roi.warp_from_roi(feat.detach(), roi_in, box, geo)
        _close(back.detach(), G[f'{name}_back'])
        back.backward(torch.from_numpy(G[f'{name}_gback']))
        _close(roi_in.grad, G[f'{name}_droi_in'])


def test_losses(golden_dir):
    G = np.load(os.path.join(golden_dir, 'losses.npz'))
    p = torch.from_numpy(G['c2_p']).requires_grad_(True)
    lab = torch.from_numpy(G['c2_lab']).long()
    for name, fn in O_loss.BINARY.items():
        p.grad = None
        v = fn(p, lab)
        v.backward()
        _close(v.detach(), G[f'c2_{name}'])
        _close(p.grad, G[f'c2_{name}_dp'])
    p = torch.from_numpy(G['c3_p']).requires_grad_(True)
    lab = torch.from_numpy(G['c3_lab']).long()
    onehot = torch.nn.functional.one_hot(lab[:, 0], 3).permute(0, 4, 1, 2, 3)
    v = O_loss.weighted_ce(p, None, onehot=onehot)
    v.backward()
    _close(v.detach(), G['c3_CrossEntroLoss'])
    _close(p.grad, G['c3_CrossEntroLoss_dp'])
    _close(O_loss.dice_class_onehot(p.detach(), onehot, 2), G['c3_DiceClassLoss2'])
    _close(O_loss.dice_class0_onehot(p.detach(), onehot), G['c3_DiceClassLoss0'])
    from oracle import infer as O_infer
    p2, lab2 = torch.from_numpy(G['c2_p']), torch.from_numpy(G['c2_lab']).long()
    _close(1 - O_infer.recall(p2, lab2), G['c2_RecallLoss'])
    _close(1 - O_infer.precision(p2, lab2), G['c2_PrecisionLoss'])
    _close(O_infer.localization_loss(p2, lab2.float()), G['c2_LocalizationLoss'])


def _run_model(cfg, size, batch, wseed):
    P = seedgen.seeded_params(O_net.param_shapes(cfg), wseed, requires_grad=True)
    x = seedgen.seeded_volume((batch, 1) + size, wseed + 1)
    label = seedgen.seeded_label((batch, 1) + size, wseed + 2)
    boxes = []
    pred, masks = O_net.forward(P, cfg, x, True, boxes)
    total, levels = O_step.total_loss(pred, masks, label, O_step.dynamic_weights(0))
    total.backward()
    return P, x, label, pred, masks, boxes, total, levels


SMALL = dict(num_layers=[8, 8, 8, 16, 32], roi_size_list=[20, 12, 9, 10, 6])


def test_model_small(golden_dir):
    G = np.load(os.path.join(golden_dir, 'model_small.npz'))
    cfg = O_net.NetConfig(**SMALL)
    P, x, label, pred, masks, boxes, total, levels = _run_model(cfg, (32, 32, 32), 2, 100)
    _close(pred.detach(), G['out'])
    for i, m in enumerate(masks):
        _close(m.detach(), G[f'mask{i}'])
    for i, b in enumerate(boxes):
        _close(b, G[f'box{i}'], 0)
    _close(total.detach(), G['total'])
    _close(torch.tensor([[v.item() for v in lv] for lv in levels]), G['level_losses'])
    _close(O_loss.dice_class(pred.detach(), label.long()), G['dice'])
    norms = dict(zip(G['grad_keys'], G['grad_norms']))
    for k, p in P.items():
        if k in norms:
            assert abs(p.grad.double().norm().item() - norms[k]) <= 2e-4 * max(1.0, norms[k])
        else:
            assert p.grad is None and k in set(G['nograd_keys'])
    assert len(G['nograd_keys']) == 14
    for k in G.files:
        if k.startswith('grad::'):
            _close(P[k[6:]].grad, G[k], 2e-4)
    with torch.no_grad():
        onehot = O_net.forward(P, cfg, x, training=False)
    assert int(onehot[:, 1].sum().item()) == int(G['onehot_fg_count'])


def test_model_wide_and_full(golden_dir):
    for tag, cfg, size, wseed in (('small_wide', O_net.NetConfig(**SMALL), (64, 96, 16), 200),
                                  ('full32', O_net.NetConfig(), (32, 32, 32), 300)):
        G = np.load(os.path.join(golden_dir, f'model_{tag}.npz'))
        P, x, label, pred, masks, boxes, total, levels = _run_model(cfg, size, 1, wseed)
        _close(pred.detach().flatten()[torch.from_numpy(G['out_idx'])], G['out_sample'])
        _close(total.detach(), G['total'])
        for i, b in enumerate(boxes):
            _close(b, G[f'box{i}'], 0)
        norms = dict(zip(G['grad_keys'], G['grad_norms']))
        worst = max(abs(P[k].grad.double().norm().item() - n) / max(1.0, n) for k, n in norms.items())
        assert worst <= 2e-4


def test_label_pyramid_and_weights():
    lab = seedgen.seeded_label((1, 1, 32, 32, 16), 7)
    pyr = O_step.label_pyramid(lab, 5)
    assert [tuple(p.shape[2:]) for p in pyr] == [(32, 32, 16), (16, 16, 16), (8, 8, 16), (4, 4, 8), (2, 2, 8)]
    w = O_step.dynamic_weights(0)
    assert np.allclose(w, [0.15, 0.25, 0.4, 0.5, 1.0])
    assert O_step.dynamic_weights(799)[0] == 2.0


def test_multiclass_losses_golden(golden_dir):
    """the oracle's multi-class deep-supervision loss reproduces the reference's level losses and total on the
    reference's own predictions (tests/golden/model_multi_small.npz)"""
    from oracle import step as O_step
    from oracle import seedgen
    G = np.load(os.path.join(golden_dir, 'model_multi_small.npz'))
    predict = torch.from_numpy(G['out'])
    masks = [torch.from_numpy(G[f'mask{i}']) for i in range(4)]
    label = seedgen.seeded_label((2, 1, 32, 32, 32), 402, n_classes=3)
    total, per_level = O_step.total_loss_multi(predict, masks, label, tuple(G['weights']))
    assert abs(total.item() - float(G['total'])) <= 1e-5 * max(1.0, abs(float(G['total'])))
    got = np.array([[v.item() for v in vals] for vals in per_level])
    assert np.allclose(got, G['level_losses'], rtol=1e-5, atol=1e-6)

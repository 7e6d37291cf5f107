"""Oracle: one inner training step (forward, 5-level deep-supervision loss, backward) (test infrastructure).

Restates utils/utils_3D_embed_full.py:55-91 (single-class) without AMP/GradScaler
(the oracle is fp32), train3D.py:139-155 (which losses at which level) and
utils/utils_3D_embed_full.py:16-19 + train3D.py:122-137 (per-epoch level weights).
"""
import math

import torch
import torch.nn.functional as F

from . import losses as _losses
from . import net as _net


def level_weight(t, T, default_weight=0.2, initial_weight=1.0, final_weight=1.0):
    """utils_3D_embed_full.py:16-19."""
    t = max(t, 0)
    return min(initial_weight + default_weight * math.exp(t / (5 * T)), final_weight)


def dynamic_weights(epoch, T=12, warmup=10,
                    weight_list=(0.05, 0.05, 0.1, 0.1, 1.0),
                    initial_weight=(0.1, 0.2, 0.3, 0.4, 1.0),
                    final_weight=(2., 1.5, 1.0, 1., 1.0)):
    """Level weights of one epoch (train3D.py:122-137 with the defaults of 88-96, T=12 / warm-up 10 of 233-236)."""
    return tuple(level_weight(epoch - warmup, T, weight_list[i], initial_weight[i], final_weight[i])
                 for i in range(len(weight_list)))


def level_criterions(n_levels=5, criterion_list=('CrossEntroLoss', 'DiceClassLoss')):
    """train3D.py:139-155: CE+BalanceDice for the coarse levels, CE+DiceClass for the two finest."""
    out = []
    for i in range(n_levels):
        if i < n_levels - 2:
            out.append(_losses.get_criterions(['CrossEntroLoss', 'BalanceDiceLoss']))
        elif i == n_levels - 2:
            out.append(_losses.get_criterions(['CrossEntroLoss', 'DiceClassLoss']))
        else:
            out.append(_losses.get_criterions(list(criterion_list)))
    return out


def label_pyramid(label, n_levels=5):
    """Labels seen by level 0..n-1 (utils_3D_embed_full.py:64,73-76): level 0 = full res,
    level 1 = max-pool (2,2,1), then alternately (2,2,1) after odd / (2,2,2) after even levels."""
    out = [label]
    cur = F.max_pool3d(label.float(), kernel_size=(2, 2, 1), stride=(2, 2, 1))
    for lvl in range(1, n_levels):
        out.append(cur)
        k = 2 if lvl % 2 == 0 else (2, 2, 1)
        cur = F.max_pool3d(cur, kernel_size=k, stride=k)
    return out


def total_loss(predict, masks, label, weights, criterions=None):
    """Weighted deep-supervision loss (utils_3D_embed_full.py:66-82). Returns (total, [[per-loss values] per level])."""
    n_levels = len(weights)
    criterions = criterions or level_criterions(n_levels)
    pyramid = label_pyramid(label, n_levels)
    per_level = []
    for lvl in range(n_levels):
        pred = predict if lvl == 0 else masks[-lvl]
        tgt = pyramid[lvl].long()
        per_level.append([fn(pred, tgt) for fn in criterions[-lvl - 1].values()])
    total = sum(sum(vals) * w for vals, w in zip(per_level, weights))
    return total, per_level


def total_loss_multi(predict, masks, label, weights, num_classes=3,
                     criterion_list=('CrossEntroLoss', 'DiceClassLoss', 'DiceClassLoss2'), criterion_weight=(10, 1, 2)):
    """Multi-class step (utils/utils_3D_multi_class.py:68-102, train3D_multi_class.py:85-90,139-155): the same criterion list
    at every level, each loss scaled by `criterion_weight`, targets = one-hot of the (max-pooled) integer labels."""
    crit = _losses.get_multi_criterions(list(criterion_list))
    pyramid = label_pyramid(label, len(weights))
    per_level = []
    for lvl in range(len(weights)):
        pred = predict if lvl == 0 else masks[-lvl]
        tgt = pyramid[lvl].long().squeeze(1)                                   # [N, ...]
        onehot = F.one_hot(tgt, num_classes).movedim(-1, 1).float()            # [N, C, ...]
        per_level.append([w * fn(pred, onehot) for fn, w in zip(crit.values(), criterion_weight)])
    total = sum(sum(vals) * w for vals, w in zip(per_level, weights))
    return total, per_level


def train_step(P, cfg: _net.NetConfig, x, label, weights, step_times=1):
    """Forward + loss + backward on a dict of leaf parameters; grads land in P[k].grad."""
    predict, masks = _net.forward(P, cfg, x, training=True)
    total, per_level = total_loss(predict, masks, label, weights)
    (total / step_times).backward()
    return total.detach(), per_level, predict.detach(), [m.detach() for m in masks]

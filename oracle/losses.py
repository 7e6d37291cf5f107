"""Oracle: the deep-supervision losses of the single-class and multi-class paths (test infrastructure).

Restates (paths relative to the reference root):
  loss/criterions.py:696-735        class-mass-weighted focal-style CE on probabilities (binary labels)
  loss/criterions.py:35-70          per-class Dice loss (the parity metric)
  loss/criterions.py:416-442        generalised (balanced) Dice loss
  loss/multi_criterions.py:594-615  CE with one-hot targets
  loss/multi_criterions.py:30-110   per-class Dice with one-hot targets
  loss/criterions.py:773-782        name -> module registry
All take probabilities `predict [N,C,...]`; binary variants take integer labels
`target [N,1,...]`, multi-class variants take one-hot `target [N,C,...]`.
"""
import torch


def _rows(t):
    """[N,C,...] -> [N,S,C]"""
    return t.flatten(2).transpose(2, 1)


def _binary_onehot(target):
    t = _rows(target).squeeze(2)
    return torch.stack([1 - t, t], dim=-1)


def weighted_ce(predict, target, eps=1e-5, onehot=None):
    """mean( -w_c (1-p) t log(clamp(p,1e-6)) ),  w_c = (T - (sum_s p_c + eps)) / T,  T = sum(t) per sample."""
    p = _rows(predict)
    t = _binary_onehot(target) if onehot is None else _rows(onehot)
    logp = torch.log(torch.clamp(p, min=1e-6))
    mass = p.sum(dim=1, keepdim=True) + eps
    total = t.sum(dim=(1, 2), keepdim=True)
    w = (total - mass) / total
    return torch.mean(-w * (1 - p) * t * logp)


def dice_class(predict, target, class_index=1, eps=1e-9):
    """1 - mean_b (2 sum p_c t + eps) / (sum p_c + sum t + eps);  target = labels of that class as 0/1."""
    p = _rows(predict)[:, :, class_index]
    t = _rows(target).squeeze(2)
    inter = 2 * torch.sum(p * t, dim=-1) + eps
    denom = torch.sum(p + t, dim=-1) + eps
    return 1 - torch.mean(inter / denom)


def dice_class_onehot(predict, onehot, class_index, eps=1e-9):
    """multi_criterions variant: the class column of a one-hot target."""
    p = _rows(predict)[:, :, class_index]
    t = _rows(onehot)[:, :, class_index]
    inter = 2 * torch.sum(p * t, dim=-1) + eps
    denom = torch.sum(p + t, dim=-1) + eps
    return 1 - torch.mean(inter / denom)


def balanced_dice(predict, target, eps=1e-5):
    """Generalised Dice with class weights 1/(sum t_c + eps)^2."""
    p = _rows(predict)
    t = _binary_onehot(target)
    w = 1 / (t.sum(dim=1, keepdim=True) + eps) ** 2
    inter = 2 * torch.sum(p * t * w, dim=(1, 2)) + eps
    denom = torch.sum((p + t) * w, dim=(1, 2)) + eps
    return 1 - torch.mean(inter / denom)


BINARY = {
    'CrossEntroLoss': weighted_ce,
    'DiceClassLoss': dice_class,
    'BalanceDiceLoss': balanced_dice,
}


def _multi_ce(predict, onehot):
    return weighted_ce(predict, None, onehot=onehot)


def dice_class0_onehot(predict, onehot, eps=1e-9):
    """multi_criterions.py:30-56 (DiceClassLoss0): Dice of the foreground union, 1 - class 0 on both sides."""
    p = 1 - _rows(predict)[:, :, 0]
    t = 1 - _rows(onehot)[:, :, 0]
    inter = 2 * torch.sum(p * t, dim=-1) + eps
    denom = torch.sum(p + t, dim=-1) + eps
    return 1 - torch.mean(inter / denom)


# loss/multi_criterions.py registry (704-): callables (predict [N,C,...], one-hot target [N,C,...])
MULTI = {
    'CrossEntroLoss': _multi_ce,
    'DiceClassLoss': lambda p, t: dice_class_onehot(p, t, 1),
    'DiceClassLoss2': lambda p, t: dice_class_onehot(p, t, 2),
    'DiceClassLoss0': dice_class0_onehot,
}


def get_multi_criterions(names):
    return {n: MULTI[n] for n in names}


def get_criterions(names):
    """Same contract as loss/criterions.py:773-782: dict name -> callable(predict, target) -> 0-dim tensor."""
    return {n: BINARY[n] for n in names}

"""Loss surface of loss/criterions.py (binary labels) and loss/multi_criterions.py (class labels) over
the fused HIP loss kernels.

`get_criterions(name_list) -> {name: nn.Module}` keeps the reference contract (loss/criterions.py:773-782):
every module is `forward(predict[N,C,...], target) -> 0-dim tensor`.  `target` is an integer label
volume [N,1,...] (or a one-hot [N,C,...] tensor, as the multi-class scripts pass).  All losses of one
decoder level share one streaming reduction (`LevelCriterion`).
"""
import torch
import torch.nn as nn

from . import ops


def _channels_last(predict):
    """[N,C,...] (any strides) -> contiguous fp32 [N,...,C]"""
    nd = predict.dim()
    cl = predict.permute(0, *range(2, nd), 1)
    if cl.dtype != torch.float32:
        cl = cl.float()
    return cl.contiguous()


def _labels(target, n_class):
    """[N,1,...] integer labels, or one-hot [N,C,...], -> uint8 [N,...]"""
    if target.dim() >= 2 and target.shape[1] == n_class and n_class > 1 and target.shape[1] != 1:
        target = target.argmax(dim=1, keepdim=True)
    return target.reshape(target.shape[0], *target.shape[2:]).to(torch.uint8).contiguous()


class LevelCriterion(nn.Module):
    """Weighted sum of CE / balanced Dice / per-class Dice on one prediction, one kernel pass.

    spec: {'CrossEntroLoss': w, 'BalanceDiceLoss': w, 'DiceClassLoss': w (class 1), 'DiceClassLoss2': w (class 2),
           'DiceClassLoss0c': w (class 0), 'DiceClassLoss0': w (foreground union 1 - class 0, multi_criterions.py:30-56)}.  Returns (total, {name: w * value}) with values detached: what the reference
    scripts log (`criterions_w * l(...)`, utils_3D_multi_class.py:85; all weights are 1 in the single-class script).
    """
    _DICE = {'DiceClassLoss': 1, 'DiceClassLoss2': 2, 'DiceClassLoss0c': 0, 'DiceClassLoss0': 4}      # 4 = foreground union

    def __init__(self, spec: dict, scale: float = 1.0, scale_dev=None):
        super().__init__()
        unknown = set(spec) - {'CrossEntroLoss', 'BalanceDiceLoss', *self._DICE}
        if unknown:
            raise KeyError(f'no HIP kernel for losses {sorted(unknown)}')
        self.spec = dict(spec)
        self.scale = scale
        self.scale_dev = scale_dev          # 1-element fp32 device tensor: run-time factor on top of `scale` (captured graphs)

    def forward(self, predict, target):
        p = _channels_last(predict)
        C = p.shape[-1]
        lab = _labels(target, C)
        sc = self.scale
        wd = [0.0] * 5
        for name, cls in self._DICE.items():
            if name in self.spec:
                wd[cls] += self.spec[name] * sc
        total, values = ops.level_loss(p, lab, self.spec.get('CrossEntroLoss', 0.0) * sc,
                                       self.spec.get('BalanceDiceLoss', 0.0) * sc, wd, self.scale_dev)
        named = {}
        for name, w in self.spec.items():
            if name == 'CrossEntroLoss':
                v = values[1]
            elif name == 'BalanceDiceLoss':
                v = values[2]
            else:
                v = values[3 + self._DICE[name]]
            named[name] = v if w == 1.0 else v * w
        return total, named


class _Single(nn.Module):
    NAME = None

    def __init__(self):
        super().__init__()
        self.impl = LevelCriterion({self.NAME: 1.0})

    def forward(self, predict, target):
        return self.impl(predict, target)[0]


class CrossEntroLoss(_Single):
    """loss/criterions.py:696-735"""
    NAME = 'CrossEntroLoss'


class DiceClassLoss(_Single):
    """loss/criterions.py:35-70 (class 1)"""
    NAME = 'DiceClassLoss'


class DiceClassLoss2(_Single):
    """loss/multi_criterions.py:85-110 (class 2)"""
    NAME = 'DiceClassLoss2'


class BalanceDiceLoss(_Single):
    """loss/criterions.py:416-442"""
    NAME = 'BalanceDiceLoss'


class DiceClassLoss0(_Single):
    """loss/multi_criterions.py:30-56: Dice of the foreground union (1 - class 0 on both sides)"""
    NAME = 'DiceClassLoss0'


class _EvalMetric(nn.Module):
    """the evaluation losses train3D.py:143 requests besides the Dice losses (`eval_list`): computed on the un-thresholded class
    probabilities by the metric kernels of the inference driver (csrc/infer.hip); evaluation only, no gradient"""
    INDEX, COMPLEMENT = 0, False

    def forward(self, predict, target):
        from . import infer
        with torch.no_grad():
            v = infer.evaluate(predict, target, threshold=-1.0)[infer.METRIC_NAMES[self.INDEX]]
        return 1.0 - v if self.COMPLEMENT else v


class RecallLoss(_EvalMetric):
    """loss/criterions.py:314-345: 1 - mean_b (sum p t + 1e-5) / (sum t + 1e-5)"""
    INDEX, COMPLEMENT = 1, True


class PrecisionLoss(_EvalMetric):
    """loss/criterions.py:382-413: 1 - mean_b (sum p t + 1e-5) / (sum p + 1e-5)"""
    INDEX, COMPLEMENT = 2, True


class LocalizationLoss(_EvalMetric):
    """loss/criterions.py:179-241 (as written there: all three "axes" reduce to the H profile)"""
    INDEX = 3


Loss_Dict = {
    'CrossEntroLoss': CrossEntroLoss,
    'DiceClassLoss': DiceClassLoss,
    'DiceClassLoss0': DiceClassLoss0,
    'DiceClassLoss2': DiceClassLoss2,
    'BalanceDiceLoss': BalanceDiceLoss,
    'RecallLoss': RecallLoss,
    'PrecisionLoss': PrecisionLoss,
    'LocalizationLoss': LocalizationLoss,
}


def get_criterions(name_list):
    """loss/criterions.py:773-782"""
    return {name: Loss_Dict[name]() for name in name_list}

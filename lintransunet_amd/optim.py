"""Optimizer / schedule / checkpoint shell of train3D.py.

  train3D.py:193       optimizer = torch.optim.AdamW(model.parameters(), lr=1e-4)
  train3D.py:195-201   ReduceLROnPlateau(mode='min', factor=0.8, patience=5, threshold=1e-2, cooldown=1, min_lr=1e-7)
  train3D.py:226-291   dynamic level weights with a 10-epoch warm-up (train.get_dynamic_weight), best-checkpoint logic that
                       saves `model.state_dict()` as temp_model.pt

The AdamW step is one HIP launch per gradient bucket (`ltu_adamw`): parameters and moments live in flat fp32 buffers laid out
exactly like the reducer's gradient buckets (each `nn.Parameter` becomes a view into them; `state_dict()` is unaffected).
The schedule and checkpoint logic are host code with the same argument meaning as the torch classes the reference uses.
"""
import math
import os

import torch

from . import _lib
from .ops import _p, _s


class FusedAdamW:
    """torch.optim.AdamW semantics (defaults lr 1e-3, betas (0.9, 0.999), eps 1e-8, weight_decay 1e-2, no amsgrad) for the
    parameters held by a `train.GradReducer`.  Parameters that never receive a gradient are not updated (torch skips
    `p.grad is None` the same way)."""

    def __init__(self, reducer, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.reducer = reducer
        self.generation = reducer.generation          # the flat layout this optimizer's parameter / moment buffers mirror
        self.param_groups = [dict(lr=float(lr), betas=tuple(betas), eps=float(eps), weight_decay=float(weight_decay))]
        self.step_count = 0
        self.flat_p, self.m, self.v = [], [], []
        for params, flat_g in zip(reducer.buckets, reducer.flat):
            if not flat_g.is_cuda:
                raise _lib.LtuError('FusedAdamW runs on the GPU only (no CPU fallback)')
            flat = torch.empty_like(flat_g)
            off = 0
            for p in params:
                n = p.numel()
                flat[off:off + n].copy_(p.data.reshape(-1))
                p.data = flat[off:off + n].view_as(p)          # the model's weight store notices the move and rebuilds its table
                off += n
            self.flat_p.append(flat)
            self.m.append(torch.zeros_like(flat))
            self.v.append(torch.zeros_like(flat))

    def zero_grad(self):
        self.reducer.zero_grad()

    def step(self, grad_scale=1.0):
        if self.reducer.generation != self.generation:
            raise RuntimeError('the reducer re-assigned its gradient buckets (rebucket) after this optimizer was built: its flat '
                               'parameter / moment buffers no longer match; call rebucket() before constructing the optimizer')
        g = self.param_groups[0]
        self.step_count += 1
        for p, gr, m, v in zip(self.flat_p, self.reducer.flat, self.m, self.v):
            _lib.call('ltu_adamw', _p(p), _p(gr), _p(m), _p(v), p.numel(), g['lr'], g['betas'][0], g['betas'][1], g['eps'],
                      g['weight_decay'], self.step_count, float(grad_scale), _s())

    def state_dict(self):
        return dict(step=self.step_count, param_groups=[dict(g) for g in self.param_groups],
                    m=[t.clone() for t in self.m], v=[t.clone() for t in self.v])

    def load_state_dict(self, sd):
        self.step_count = int(sd['step'])
        self.param_groups = [dict(g) for g in sd['param_groups']]
        for dst, src in zip(self.m, sd['m']):
            dst.copy_(src)
        for dst, src in zip(self.v, sd['v']):
            dst.copy_(src)


class ReduceLROnPlateau:
    """torch.optim.lr_scheduler.ReduceLROnPlateau for any object with `param_groups` (same arguments, same state machine:
    relative/absolute threshold, patience counted in bad epochs, cooldown, per-group min_lr, eps)."""

    def __init__(self, optimizer, mode='min', factor=0.1, patience=10, threshold=1e-4, threshold_mode='rel', cooldown=0,
                 min_lr=0.0, eps=1e-8):
        if factor >= 1.0:
            raise ValueError('Factor should be < 1.0.')
        if mode not in ('min', 'max') or threshold_mode not in ('rel', 'abs'):
            raise ValueError('unknown mode')
        self.optimizer, self.mode, self.factor, self.patience = optimizer, mode, factor, patience
        self.threshold, self.threshold_mode, self.cooldown, self.eps = threshold, threshold_mode, cooldown, eps
        n = len(optimizer.param_groups)
        self.min_lrs = list(min_lr) if isinstance(min_lr, (list, tuple)) else [min_lr] * n
        self.best = math.inf if mode == 'min' else -math.inf
        self.num_bad_epochs = 0
        self.cooldown_counter = 0
        self.last_epoch = 0

    def _is_better(self, a):
        if self.mode == 'min':
            return a < (self.best * (1.0 - self.threshold) if self.threshold_mode == 'rel' else self.best - self.threshold)
        return a > (self.best * (self.threshold + 1.0) if self.threshold_mode == 'rel' else self.best + self.threshold)

    def step(self, metrics):
        current = float(metrics)
        self.last_epoch += 1
        if self._is_better(current):
            self.best = current
            self.num_bad_epochs = 0
        else:
            self.num_bad_epochs += 1
        if self.cooldown_counter > 0:
            self.cooldown_counter -= 1
            self.num_bad_epochs = 0
        if self.num_bad_epochs > self.patience:
            for i, g in enumerate(self.optimizer.param_groups):
                old = float(g['lr'])
                new = max(old * self.factor, self.min_lrs[i])
                if old - new > self.eps:
                    g['lr'] = new
            self.cooldown_counter = self.cooldown
            self.num_bad_epochs = 0


def reference_schedule(optimizer):
    """the scheduler of train3D.py:195-201"""
    return ReduceLROnPlateau(optimizer, mode='min', factor=0.8, patience=5, threshold=1e-2, cooldown=1, min_lr=1e-7)


class BestCheckpoint:
    """Best-checkpoint logic of train3D.py:254-268: whenever the eval loss does not exceed the best so far, write
    `model.state_dict()` to `<dir>/temp_model.pt` (loadable by the reference's get_model, train3D.py:104-120).  What the
    reference does not keep -- optimizer moments, step, learning rate, RNG state -- goes to a side file."""

    def __init__(self, model_dir, rank=None):
        self.dir = model_dir
        self.best_eval = math.inf
        self.best_train = math.inf
        if rank is None:       # data-parallel run: every rank holds the same weights, only rank 0 writes
            import torch.distributed as dist
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        self.rank = rank
        if self.rank == 0:
            os.makedirs(model_dir, exist_ok=True)

    def update(self, model, eval_loss, train_loss, optimizer=None):
        if eval_loss > self.best_eval:
            return False
        self.best_eval, self.best_train = eval_loss, train_loss
        if self.rank != 0:
            return True
        torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, os.path.join(self.dir, 'temp_model.pt'))
        if optimizer is not None:
            side = dict(optimizer={k: ([t.cpu() for t in v] if isinstance(v, list) and v and torch.is_tensor(v[0]) else v)
                                   for k, v in optimizer.state_dict().items()},
                        rng_cpu=torch.get_rng_state(), rng_cuda=torch.cuda.get_rng_state_all() if torch.cuda.is_available() else None)
            torch.save(side, os.path.join(self.dir, 'temp_model.extra.pt'))
        return True

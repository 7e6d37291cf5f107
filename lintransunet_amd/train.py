"""The inner training step (utils/utils_3D_embed_full.py:55-91) and its data-parallel form.

One process per GPU.  The batch is sharded over ranks; every normalisation, ROI box, softmax-over-tokens
and loss term of this network is per-sample, so averaging the per-rank gradients reproduces the
full-batch gradient of the reference's nn.DataParallel step.  The only exchange is
one bucketed all-reduce of the gradients over RCCL/xGMI (gloo on CPU tests), launched bucket by bucket
from autograd hooks so that it overlaps the rest of backward.
"""
import math
import os

import torch
import torch.distributed as dist

from . import ops
from .losses import LevelCriterion


def get_weight(t, T, default_weight=0.2, initial_weight=1.0, final_weight=1.0):
    """utils/utils_3D_embed_full.py:16-19"""
    t = max(t, 0)
    return min(initial_weight + default_weight * math.exp(t / (5 * T)), final_weight)


def get_dynamic_weight(epochs, T=12, warmup_step=10, weight_list=(0.05, 0.05, 0.1, 0.1, 1.0),
                       initial_weight=(0.1, 0.2, 0.3, 0.4, 1.0), final_weight=(2., 1.5, 1.0, 1., 1.0)):
    """train3D.py:122-137: list over epochs of per-level weight tuples"""
    cols = [[get_weight(j - warmup_step, T, weight_list[i], initial_weight[i], final_weight[i]) for j in range(epochs)]
            for i in range(len(weight_list))]
    return list(zip(*cols))


def level_specs(n_levels=5, criterion_list=('CrossEntroLoss', 'DiceClassLoss'), criterion_weight=None):
    """train3D.py:139-155 (single class: coarse levels CE+BalanceDice, two finest CE+DiceClass);
    with `criterion_weight` the multi-class weighting of utils/utils_3D_multi_class.py:85-102 applies."""
    cw = criterion_weight or [1.0] * len(criterion_list)
    final = {n: w for n, w in zip(criterion_list, cw)}
    if criterion_weight is not None:      # multi-class script: the same list at every level
        return [dict(final) for _ in range(n_levels)]
    specs = []
    for i in range(n_levels):
        if i < n_levels - 2:
            specs.append({'CrossEntroLoss': 1.0, 'BalanceDiceLoss': 1.0})
        elif i == n_levels - 2:
            specs.append({'CrossEntroLoss': 1.0, 'DiceClassLoss': 1.0})
        else:
            specs.append(final)
    return specs


def label_pyramid(label, n_levels=5):
    """uint8 [B,1,H,W,D] -> labels of level 0..n-1 (utils/utils_3D_embed_full.py:64,73-76)"""
    lab = label.reshape(label.shape[0], *label.shape[2:]).to(torch.uint8).contiguous()
    out = [lab]
    cur = ops.label_maxpool(lab, 1)
    for lvl in range(1, n_levels):
        out.append(cur)
        if lvl < n_levels - 1:
            cur = ops.label_maxpool(cur, 2 if lvl % 2 == 0 else 1)
    return out


def deep_supervision_loss(predict, masks, label, weights, specs=None, scale=1.0):
    """Per-level fused losses (utils/utils_3D_embed_full.py:66-82).  Returns (list of weighted level totals,
    list of {name: value}); `sum(totals)` is the reference's total_loss * scale."""
    n = len(weights)
    specs = specs or level_specs(n)
    pyr = label_pyramid(label, n)
    totals, named = [], []
    for lvl in range(n):
        pred = predict if lvl == 0 else masks[-lvl]
        crit = LevelCriterion(specs[-lvl - 1], scale=weights[lvl] * scale)
        tot, vals = crit(pred, pyr[lvl].unsqueeze(1))
        totals.append(tot)
        named.append(vals)
    return totals, named


def train_step(model, images, labels, weights, step_times=1, specs=None, reducer=None):
    """forward + 5-level loss + backward for one batch of patches (one `j` of utils_3D_embed_full.py:55-86).
    Returns the list of weighted level losses (device scalars, no host sync)."""
    ops.begin_step(images.device)
    predict, masks = model(images)
    totals, named = deep_supervision_loss(predict, masks, labels, weights, specs, scale=1.0 / step_times)
    if reducer is not None:
        reducer.prepare()
    torch.autograd.backward(totals, [torch.ones_like(t) for t in totals])
    ops.wgrad_branch_join()                 # weight-gradient branch (if one is installed) back into this stream
    ops.flush_deferred()                    # second stages of the two-stage reductions still queued by backward
    if reducer is not None:
        reducer.finish()
    return totals, named


class GradReducer:
    """Bucketed gradient all-reduce (mean over ranks) overlapped with backward.

    Parameters are bucketed in reverse registration order (decoder tail first = the order in which
    backward produces gradients).  Each parameter's `.grad` is a view into its bucket's flat buffer; a
    post-accumulate hook counts arrivals and launches `all_reduce(async_op=True)` when a bucket is full.
    Parameters that never receive a gradient (the 14 unused pos_encoders tensors) are left out.
    """

    def __init__(self, model, bucket_mb=16.0, unused=None, group=None, fused=True):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        unused = set(unused or [])
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad and n not in unused]
        named.reverse()
        cap = int(bucket_mb * 1024 * 1024 / 4)
        self.buckets, cur, cur_n = [], [], 0
        for n, p in named:
            cur.append(p)
            cur_n += p.numel()
            if cur_n >= cap:
                self.buckets.append(cur)
                cur, cur_n = [], 0
        if cur:
            self.buckets.append(cur)
        self.flat, self.pending, self.handles = [], [], []
        self.bucket_of = {}
        for bi, params in enumerate(self.buckets):
            flat = torch.zeros(sum(p.numel() for p in params), device=params[0].device, dtype=torch.float32)
            off = 0
            for p in params:
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
                self.bucket_of[p] = bi
                p.register_post_accumulate_grad_hook(self._hook)     # gradients that arrive through autograd
                if fused and p.is_cuda:
                    p._ltu_grad = p.grad                            # ... and those written by the wgrad kernels directly
                    p._ltu_hook = self._hook
            self.flat.append(flat)
        self.pending = [0] * len(self.buckets)
        self.active = False
        # RCCL averages inside the collective (ncclAvg); gloo (CPU tests) sums and the buckets are divided afterwards
        self.avg = self.world > 1 and dist.get_backend(group) == 'nccl'

    def _all_reduce(self, flat):
        return dist.all_reduce(flat, op=dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _finish_bucket(self, flat, handle):
        handle.wait()
        if not self.avg:
            flat.div_(self.world)

    def zero_grad(self):
        for f in self.flat:
            f.zero_()

    def prepare(self):
        self.pending = [len(b) for b in self.buckets]
        self.handles = []
        self.active = True

    def _hook(self, p):
        if not self.active:
            return
        bi = self.bucket_of[p]
        self.pending[bi] -= 1
        if self.pending[bi] == 0 and self.world > 1:
            ops.flush_deferred()            # pending second-stage reductions may still owe this bucket their sums
            self.handles.append((bi, self._all_reduce(self.flat[bi])))

    def reduce_all(self):
        """all-reduce every bucket now (used after a graph replay, where no hooks run)"""
        if self.world > 1:
            hs = [self._all_reduce(f) for f in self.flat]
            for f, h in zip(self.flat, hs):
                self._finish_bucket(f, h)

    def finish(self):
        ops.flush_deferred()                # safety net for callers that ran backward without train_step
        self.active = False
        if self.world > 1:
            launched = {bi for bi, _ in self.handles}
            for bi in range(len(self.buckets)):       # buckets holding a parameter that got no gradient this step
                if bi not in launched:
                    self.handles.append((bi, self._all_reduce(self.flat[bi])))
            for bi, h in self.handles:
                self._finish_bucket(self.flat[bi], h)
        self.handles = []


class GraphedStep:
    """One training step (arena reset, weight prep, forward, 5-level loss, backward into the reducer's flat gradient
    buffers) captured once into a HIP graph and replayed: ~1 400 kernel launches become one `hipGraphLaunch`.

    The batch lives in static device buffers (`copy_` new data in); dropout masks stay fresh across replays because the
    kernels mix a device-resident step counter, advanced inside the graph, into their Philox seeds.  With more than one rank
    the gradient all-reduce runs after the replay (bucket by bucket, asynchronously) instead of from autograd hooks.
    """

    def __init__(self, model, images, labels, weights, reducer, step_times=1, specs=None, warmup=2):
        self.model, self.reducer = model, reducer
        dev = images.device
        self.x, self.lab = images.clone(), labels.clone()
        self.counter = torch.zeros(1, device=dev, dtype=torch.int64)
        ops.set_step_counter(self.counter)

        # opt-in (LTU_WGRAD_BRANCH=1): measured +0.5 % only (21.09 vs 21.20 ms), see DESIGN.md section 7
        self.wg_stream = torch.cuda.Stream(device=dev) if os.environ.get('LTU_WGRAD_BRANCH', '0') == '1' else None

        def body():
            self.counter.add_(1)
            reducer.zero_grad()
            ops.wgrad_branch_install(self.wg_stream)       # projection weight gradients on a second graph branch
            try:
                return train_step(model, self.x, self.lab, weights, step_times=step_times, specs=specs, reducer=None)
            finally:
                ops.wgrad_branch_install(None)

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):        # warm-up on a side stream: allocator pools, arena size, weight store
                body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.barrier()                 # no collective in flight while the stream is capturing
            torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        # thread-local capture mode: the process group's watchdog thread may query events while this thread captures
        with torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
            self.totals, self.named = body()

    def __call__(self, images=None, labels=None):
        if images is not None:
            self.x.copy_(images, non_blocking=True)
            self.lab.copy_(labels, non_blocking=True)
        self.graph.replay()
        self.reducer.reduce_all()
        return self.totals, self.named


UNUSED_PARAMETERS = tuple(f'decode.bridge_list.4.transformer.pos_encoders.{n}.proj.{k}'
                          for n in range(1, 8) for k in ('weight', 'bias'))


def broadcast_parameters(model, src=0, group=None):
    """one-time parameter sync from rank `src` (replaces DataParallel's per-step replicate)"""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        for p in model.parameters():
            dist.broadcast(p.data, src, group=group)

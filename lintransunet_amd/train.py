"""The inner training step (utils/utils_3D_embed_full.py:55-91) and its data-parallel form.

One process per GPU.  The batch is sharded over ranks; every normalisation, ROI box, softmax-over-tokens
and loss term of this network is per-sample, so averaging the per-rank gradients reproduces the
full-batch gradient of the reference's nn.DataParallel step.  The only exchange is
one bucketed all-reduce of the gradients over RCCL/xGMI, launched bucket by bucket from autograd hooks so that it
overlaps the rest of backward.  The exchange goes through a communicator object (lintransunet_amd/comm.py): `RcclComm` = direct RCCL
calls behind the C-ABI on the GPU, `GlooComm` = the same interface on CPU tensors (host control plane and CPU tests).
"""
import math
import os

import torch
import torch.distributed as dist

from . import comm as _comm
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


def deep_supervision_loss(predict, masks, label, weights, specs=None, scale=1.0, level_scale=None, pyr=None):
    """Per-level fused losses (utils/utils_3D_embed_full.py:66-82).  Returns (list of weighted level totals,
    list of {name: value}); `sum(totals)` is the reference's total_loss * scale.
    level_scale: optional fp32 device tensor [n_levels] that REPLACES `weights[lvl] * scale` at run time (a captured graph
    then follows the per-epoch weights of train3D.py:122-137 and the accumulation count without re-capture)."""
    n = len(weights)
    specs = specs or level_specs(n)
    pyr = pyr or label_pyramid(label, n)      # pyr: the pyramid, already built (train_step builds it beside the encoder)
    totals, named = [], []
    for lvl in range(n):
        pred = predict if lvl == 0 else masks[-lvl]
        if level_scale is not None:
            crit = LevelCriterion(specs[-lvl - 1], scale=1.0, scale_dev=level_scale[lvl:lvl + 1])
        else:
            crit = LevelCriterion(specs[-lvl - 1], scale=weights[lvl] * scale)
        tot, vals = crit(pred, pyr[lvl].unsqueeze(1))
        totals.append(tot)
        named.append(vals)
    return totals, named


def train_step(model, images, labels, weights, step_times=1, specs=None, reducer=None, ctx=None, level_scale=None, reduce=True):
    """forward + 5-level loss + backward for one batch of patches (one `j` of utils_3D_embed_full.py:55-86).
    Returns the list of weighted level losses (device scalars, no host sync).
    Gradient accumulation (utils_3D_embed_full.py:85-91): call `step_times` times with reduce=False except on the last
    micro-step; the caller zeroes the gradients before the first one."""
    lc = ctx or ops.current()
    with ops.use(lc):
        lc.begin_step(images.device)
        pyr = []
        lc.side_run(lambda: pyr.extend(label_pyramid(labels, len(weights))))     # beside the encoder (joined inside the model's forward)
        predict, masks = model(images)
        lc.side_join()
        totals, named = deep_supervision_loss(predict, masks, labels, weights, specs, scale=1.0 / step_times, level_scale=level_scale,
                                              pyr=pyr)
        if reducer is not None:
            reducer.prepare(lc, reduce=reduce)
        one = lc.one(images.device)
        torch.autograd.backward(totals, [one] * len(totals))
        lc.wq_join()                         # weight-gradient queue (if one is installed): last batch, side stream back into this one
        lc.flush_deferred()                  # second stages of the two-stage reductions still queued by backward
        if reducer is not None:
            reducer.finish()
    return totals, named


def _default_comm(who):
    """comm=None: LocalComm in a single-rank process, a GlooComm over a gloo default group (CPU tensors: tests, host control);
    NEVER a silent LocalComm at world > 1 - a multi-rank process group on any other backend means the caller has GPU gradients to
    exchange and must say how (round 3 advice: with comm=None such a run reduced and broadcast nothing, without an error)"""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == 'gloo':
            return _comm.GlooComm()
        raise RuntimeError(f'{who}: torch.distributed is initialised with {dist.get_world_size()} ranks on the {dist.get_backend()!r} backend but '
                           'no communicator was given: pass comm=lintransunet_amd.comm.RcclComm(device, control=GlooComm(gloo_group)) '
                           '(the gradient exchange goes through direct RCCL calls, the control plane through gloo)')
    return _comm.LocalComm()


class GradReducer:
    """Bucketed gradient all-reduce (mean over ranks) overlapped with backward.

    Parameters are bucketed in reverse registration order (decoder tail first = the order in which
    backward produces gradients).  Each parameter's `.grad` is a view into its bucket's flat buffer; a
    hook counts arrivals (each parameter once) and issues the bucket's all-reduce when it is full, so the collective of one bucket
    runs while backward produces the next (comm.RcclComm: the RCCL kernel is enqueued on the communicator's stream by a plain C
    call; no process-group thread takes part).  GraphedStep replays the step as linear graph segments cut at exactly those points
    and issues the collectives between them (or captures them as side branches of one graph, or issues them after the replay).
    Parameters that never receive a gradient (the 14 unused pos_encoders tensors) are left out.
    With gradient accumulation only the last micro-step reduces (`prepare(reduce=False)` otherwise).
    """

    def __init__(self, model, bucket_mb=32.0, unused=None, comm=None, fused=True, tail_mb=0.5, force_collectives=False):
        """comm: a communicator of lintransunet_amd.comm (RcclComm on the GPU, GlooComm on CPU tensors); None = LocalComm, or -
        when torch.distributed is initialised with a gloo group - a GlooComm over the default group"""
        if comm is None:
            comm = _default_comm('GradReducer')
        self.comm = comm
        # force_collectives: take the multi-rank code path (bucket plan, hooks, collectives) on a 1-rank communicator - the one-GPU
        # rehearsal of what every rank does at N > 1 (tools/check_dist_graph.py, bench.py --rehearse-comm)
        self.world = max(comm.world, 2) if force_collectives else comm.world
        unused = set(unused or [])
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad and n not in unused]
        named.reverse()
        self.cap = int(bucket_mb * 1024 * 1024 / 4)
        # The gradients that become ready last (the encoder's first layers) close their bucket at the very end of backward, so that
        # bucket's all-reduce is exposed whatever its size: the last `tail_mb` of parameters get a bucket of their own (a
        # latency-sized collective) and the bulk of the former last bucket starts its all-reduce earlier.  With the step replayed
        # as linear graph segments (GraphedStep overlap='segments') an extra bucket costs nothing measurable; as a side branch of
        # ONE graph it cost +0.2 ms (profiles/r03_bucket_sweep.txt).
        self.tail_cap = int(tail_mb * 1024 * 1024 / 4)
        self.fused = fused
        self.ready_order = []           # parameters in the order their gradients became ready in the last backward
        self._seen = set()
        self.generation = 0             # bumped whenever the flat buffers are re-assigned (optim.FusedAdamW checks it)
        for n, p in named:
            p.register_post_accumulate_grad_hook(self._hook)     # gradients that arrive through autograd
        self._assign([p for n, p in named])
        self.active = False
        self.reduce_now = True
        self.ctx = None
        self.on_bucket = None           # GraphedStep(overlap='segments'): called instead of issuing a completed bucket's collective
        self.cut, self.cut_rest = [], []

    def _assign(self, params):
        """bucket `params` in the given order: flat fp32 buffers of ~cap elements, every .grad a view into its bucket"""
        self.buckets, cur, cur_n = [], [], 0
        for p in params:
            cur.append(p)
            cur_n += p.numel()
            if cur_n >= self.cap:
                self.buckets.append(cur)
                cur, cur_n = [], 0
        if cur:
            self.buckets.append(cur)
        if self.tail_cap > 0 and self.buckets and self.world > 1:
            last, tail, n = self.buckets[-1], [], 0
            while len(last) > 1 and n + last[-1].numel() <= self.tail_cap:
                n += last[-1].numel()
                tail.insert(0, last.pop())
            if tail:
                self.buckets.append(tail)
        self.flat, self.handles = [], []
        self.bucket_of = {}
        for bi, plist in enumerate(self.buckets):
            flat = torch.zeros(sum(p.numel() for p in plist), device=plist[0].device, dtype=torch.float32)
            off = 0
            for p in plist:
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
                self.bucket_of[p] = bi
                if self.fused and p.is_cuda:
                    p._ltu_grad = p.grad                            # written by the wgrad kernels directly
                    p._ltu_hook = self._hook
            self.flat.append(flat)
        self.pending = [0] * len(self.buckets)
        self.generation += 1

    def rebucket(self):
        """Re-assign the buckets in the order in which the gradients became ready during the last backward (as DDP does after its
        first iteration).  The registration order used at construction is only a guess: in this network the decoder's coarse
        stages and the bottleneck transformer are registered early in the decoder but finish late in backward, so buckets in
        registration order all complete near the end of the step and nothing overlaps.  Call between steps, after one backward
        with hooks armed (`prepare`), before building optimizers / step graphs on top of the buffers (a GraphedStep notices moved
        gradient storage and captures again; an optimizer built on the old buffers refuses to step: `generation` changed).  Every
        rank records the same order (it depends on the autograd graph only)."""
        if not self.ready_order:
            raise RuntimeError('rebucket() needs one backward pass with armed hooks first')
        seen = set(self.ready_order)
        rest = [p for b in self.buckets for p in b if p not in seen]
        self._assign(list(self.ready_order) + rest)

    def _all_reduce(self, flat, also=None):
        """also: a second stream whose work the bucket depends on (the weight-gradient queue's side stream)"""
        return self.comm.allreduce_avg(flat, also) if also is not None else self.comm.allreduce_avg(flat)

    def zero_grad(self):
        for f in self.flat:
            f.zero_()

    def prepare(self, ctx=None, reduce=True):
        """arm the hooks for one backward; reduce=False (a non-final accumulation micro-step) only accumulates"""
        self.pending = [len(b) for b in self.buckets]
        self.handles = []
        self.cut, self.cut_rest = [], []
        self.active = True
        self.ready_order, self._seen = [], set()
        self.reduce_now = bool(reduce) and self.world > 1
        self.ctx = ctx or ops.current()          # the hooks run on autograd's thread: they must not look the context up there

    def _hook(self, p):
        if not self.active:
            return
        # Each parameter counts ONCE per backward.  A fused parameter reports twice: from its weight-gradient op (`_ltu_hook`, when
        # the kernels that write the bucket have been enqueued) and from autograd's post-accumulate hook, which PyTorch also fires
        # when the op returned no gradient for it.  Counting both closed every bucket when half of its gradients were in
        # (found in round 3 by tools/rehearse_two_ranks.py, the first run with two real ranks: a 1-rank rehearsal cannot see an
        # all-reduce that comes too early).
        if p in self._seen:
            return
        q = self.ctx.wq if self.ctx is not None else None
        if q is not None and not q['running']:
            # the context holds weight-gradient kernels back (ops.Context.wq_*): the report is parked and repeated by the batch that
            # launches them, so a bucket closes - and is reduced - where its gradients really are enqueued, on the side stream
            q['after'].append((self._hook, p))
            return
        self._seen.add(p)
        self.ready_order.append(p)
        bi = self.bucket_of[p]
        self.pending[bi] -= 1
        if self.pending[bi] == 0 and self.reduce_now:
            self.ctx.flush_deferred()       # pending second-stage reductions may still owe this bucket their sums
            if self.on_bucket is not None:
                self.cut.append(bi)
                self.on_bucket(bi)
            else:
                self.handles.append((bi, self._all_reduce(self.flat[bi], self.ctx.wq_side())))

    def reduce_all(self):
        """all-reduce every bucket now (after a graph replay that did not capture the collectives)"""
        if self.world > 1:
            hs = [self._all_reduce(f) for f in self.flat]
            for h in hs:
                h.wait()

    def finish(self):
        (self.ctx or ops.current()).flush_deferred()      # safety net for callers that ran backward without train_step
        self.active = False
        if self.reduce_now:
            launched = {bi for bi, _ in self.handles} | set(self.cut)
            rest = [bi for bi in range(len(self.buckets)) if bi not in launched]       # buckets holding a parameter that got no gradient
            if self.on_bucket is not None:
                self.cut_rest = rest                  # the caller reduces them after its last segment
            else:
                for bi in rest:
                    self.handles.append((bi, self._all_reduce(self.flat[bi], (self.ctx or ops.current()).wq_side())))
            for bi, h in self.handles:
                h.wait()
        self.handles = []


class GraphedStep:
    """One training micro-step (arena reset, weight prep, forward, 5-level loss, backward into the reducer's flat gradient
    buffers, bucketed gradient all-reduce) captured once into a HIP graph and replayed: ~1 400 kernel launches become one
    `hipGraphLaunch`.

    * The batch lives in static device buffers (`copy_` new data in); dropout masks stay fresh across replays because the
      kernels mix a device-resident step counter, advanced inside the graph, into their seeds.
    * all-reduce, three ways (LTU_GRAPH_ALLREDUCE overrides):
      `overlap='segments'` cuts the capture where a bucket's last gradient has been enqueued: the step becomes K + 1 LINEAR graphs
      (K buckets) replayed back to back on the compute stream, and after segment k the bucket's collective is issued eagerly on the
      communicator's stream - it runs beside the next segments exactly as in the eager hook path, and no graph contains a fork
      (on ROCm 7.0 a fork inside a replayed graph costs 0.15 - 1 ms, profiles/r03_bucket_sweep.txt; a linear graph costs nothing);
      `overlap='graph'` captures the collectives inside ONE graph as side branches (forked where a bucket closes, joined at the end);
      `overlap='after'` issues them after the replay (fully exposed; the fallback if the other two cannot be captured).
    * accumulation (utils/utils_3D_embed_full.py:85-91, `step_times` micro-steps per optimizer step): `step(x, y, micro=j)`
      zeroes the buckets only for j == 0 and reduces only for j == step_times - 1; each (zero, reduce) combination in use is its
      own captured graph (they share one memory pool).
    * per-epoch level weights (train3D.py:122-137) live in a device tensor read by the loss kernels: `set_weights(w)` updates
      it in place, no re-capture.
    * every replay first checks that parameter and gradient storage is where it was at capture time (e.g. an optimizer built
      afterwards that re-homes `p.data`); if not, the step is captured again.
    * the step owns its `ops.Context` (scratch arena frozen after capture), so other graphs / eager steps cannot move it.
    """

    def __init__(self, model, images, labels, weights, reducer, step_times=1, specs=None, warmup=2, overlap=None):
        self.model, self.reducer = model, reducer
        self.step_times, self.specs, self.warmup = int(step_times), specs, warmup
        self.n_levels = len(weights)
        dev = images.device
        self.dev = dev
        self.x, self.lab = images.clone(), labels.clone()
        self.overlap = overlap or os.environ.get('LTU_GRAPH_ALLREDUCE', 'segments')
        if self.overlap not in ('segments', 'graph', 'after'):
            raise ValueError("overlap must be 'segments', 'graph' or 'after'")
        self.ctx = ops.Context()
        self.counter = torch.zeros(1, device=dev, dtype=torch.int64)
        self.ctx.set_step_counter(self.counter)
        self.level_scale = torch.empty(self.n_levels, device=dev, dtype=torch.float32)
        self.set_weights(weights)
        # weight-gradient queue (ops.Context.wq_install): with the step replayed as linear segments, the weight gradients of each
        # transformer / decoder level are captured as linear graphs of their own and replayed on a side stream beside the
        # data-gradient chain of the coarser, latency-bound levels (LTU_WQ=0: everything in line)
        self.wq_stream = None
        if self.overlap == 'segments' and os.environ.get('LTU_WQ', '1') != '0':
            # on a hardware queue of its own: beside the stream the step is replayed on (the current one) and beside the
            # communicator's stream (ops.concurrent_stream probes; streams that share a queue serialise)
            avoid = [torch.cuda.current_stream(dev)]
            cs = getattr(reducer.comm, 'stream', None)
            if isinstance(cs, torch.cuda.Stream):
                avoid.append(cs)
            self.wq_stream = ops.concurrent_stream(dev, avoid)
        self.graphs = {}
        self.pool = None
        self._capture((True, True))

    def set_weights(self, weights):
        """per-level deep-supervision weights of this epoch (divided by step_times as utils_3D_embed_full.py:85 does)"""
        if len(weights) != self.n_levels:
            raise ValueError('one weight per level')
        self.weights = tuple(float(w) for w in weights)
        self.level_scale.copy_(torch.tensor([w / self.step_times for w in self.weights], dtype=torch.float32))

    def _body(self, zero, reduce, on_flush=None, on_join=None):
        self.counter.add_(1)
        # weight gradients in batches on a side stream / as graphs of their own, at SIDE_BLOCKS workgroups per launch
        self.ctx.wq_install(self.wq_stream, on_flush, on_join, width=self.SIDE_BLOCKS)
        try:
            if zero:
                # nothing accumulates into the gradient buffers before backward: the fills run on the side stream, beside the
                # encoder (joined with the other forward-side work in front of the bottleneck transformer)
                self.ctx.side_run(self.reducer.zero_grad)
            return train_step(self.model, self.x, self.lab, self.weights, step_times=self.step_times, specs=self.specs,
                              reducer=self.reducer, ctx=self.ctx, level_scale=self.level_scale,
                              reduce=reduce and self.overlap in ('graph', 'segments'))
        finally:
            self.ctx.wq_install(None)

    def _signature(self):
        ps = list(self.model.parameters())
        return (tuple(p.data_ptr() for p in ps), tuple(0 if p.grad is None else p.grad.data_ptr() for p in ps),
                tuple(f.data_ptr() for f in self.reducer.flat))

    # Width of the weight-gradient kernels while they run on the side stream: half the machine.  At their stand-alone width (256
    # workgroups) they crowd the main chain's kernels out of the CUs (every main-chain kernel of backward ran 1.3-2x longer beside
    # them); at 128 the side stream is still far from being the critical path (64: it becomes it, 14.7 ms).  tools/sweep_side_grids.sh:
    # 13.58 -> 13.29 ms.  The width is an ARGUMENT of the launches that take one (ops.Context.side_width -> ltu_linear_wgrad_group /
    # ltu_upconv_wgrad: size query and launch get the same value), not a process-wide knob flipped around the capture (round 4).
    SIDE_BLOCKS = int(os.environ.get('LTU_SIDE_BLOCKS', '128'))

    def _capture(self, key):
        self._capture_inner(key)

    def _capture_inner(self, key):
        zero, reduce = key
        dev = self.dev
        if not self.graphs:            # first capture (or re-capture): warm up allocator pools, arena size, weight store, RCCL
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(self.warmup):
                    self._body(True, True)
            torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)          # every eager collective of the warm-up has completed on the communicator's stream
        # Nothing else needs to happen before a capture that contains collectives: the communicator (comm.RcclComm) enqueues
        # RCCL's kernels from this thread through a plain C call, so there is no watchdog or progress thread that could touch an
        # event of the capturing streams (round 2's ProcessGroupNCCL path needed a sleep here and could still abort).
        if self.overlap == 'segments' and ((reduce and self.reducer.world > 1) or self.wq_stream is not None):
            segs, totals, named, rest = self._capture_segments(zero, reduce)
        else:
            graph = torch.cuda.CUDAGraph()
            # thread-local capture mode: other threads of the process (data loading, logging) may use the HIP runtime meanwhile
            kw = {} if (self.pool is None or os.environ.get('LTU_GRAPH_SHARE_POOL', '1') == '0') else {'pool': self.pool}
            with torch.cuda.graph(graph, capture_error_mode='thread_local', **kw):
                totals, named = self._body(zero, reduce)
            if self.pool is None:
                self.pool = graph.pool()
            segs, rest = [(graph, 'main', None)], []
        self.ctx.freeze()                  # the graphs hold addresses inside the scratch arena
        self.graphs[key] = (segs, totals, named, rest)
        self.sig = self._signature()

    def _capture_segments(self, zero, reduce):
        """the step as a chain of linear graphs, cut at every point where a gradient bucket is complete (see the class docstring).
        The cut happens inside the reducer's hook, i.e. on autograd's thread in the middle of backward: the capture there is ended
        and the next one begun on the same stream and memory pool (relaxed capture mode: begin and end may come from different
        threads; nothing else of this process touches the GPU meanwhile)."""
        import gc
        import warnings
        dev, red = self.dev, self.reducer
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        gc.collect()
        torch.cuda.empty_cache()
        stream = torch.cuda.Stream(device=dev)
        stream.wait_stream(torch.cuda.current_stream(dev))
        segs, cur, kind = [], [None], ['main']

        def begin():
            cur[0] = torch.cuda.CUDAGraph()
            cur[0].capture_begin(pool=self.pool, capture_error_mode='relaxed')

        def end(bi=None):
            with warnings.catch_warnings(record=True) as w:          # a segment may be empty (two buckets closing at the same point,
                warnings.simplefilter('always')                      # or nothing left after the last one): torch warns about empty graphs
                cur[0].capture_end()
            empty = any('empty' in str(x.message).lower() for x in w)
            if not empty or bi is not None:
                segs.append((None if empty else cur[0], kind[0], bi))

        def cut(bi):
            end(bi)
            begin()

        def on_join():
            end()
            segs.append((None, 'join', None))
            begin()

        def on_flush(run):
            # a batch of weight gradients (ops.Context.wq_flush): the main chain's segment ends here, the batch becomes a linear graph
            # of its own (replayed on the side stream), the main chain goes on in a new segment
            end()
            kind[0] = 'side'
            begin()
            try:
                run()
            finally:
                end()
                kind[0] = 'main'
                begin()
        with torch.cuda.stream(stream):
            begin()
            red.on_bucket = cut
            try:
                totals, named = self._body(zero, reduce, *((on_flush, on_join) if self.wq_stream is not None else (None, None)))
            except BaseException:
                try:                       # leave the stream out of capture mode before the error travels on
                    cur[0].capture_end()
                except Exception:
                    pass
                raise
            finally:
                red.on_bucket = None
            end()
        torch.cuda.current_stream(dev).wait_stream(stream)
        return segs, totals, named, list(red.cut_rest)

    def __call__(self, images=None, labels=None, micro=0):
        """replay micro-step `micro` (0 .. step_times-1) of an optimizer step on a new batch"""
        if self._signature() != self.sig:          # parameter / gradient storage moved since capture: the graph reads stale memory
            if micro != 0:
                # a re-capture runs warm-up steps that zero the buckets: the micro-steps already accumulated would be lost
                raise RuntimeError('parameter / gradient storage moved in the middle of an accumulation cycle (micro > 0): '
                                   're-home parameters (optimizers, rebucket) only between optimizer steps')
            self.graphs = {}
            self.pool = None
            self.ctx.arena.frozen = False
        key = (micro == 0, micro == self.step_times - 1)
        if key not in self.graphs:
            self._capture(key)
        if images is not None:
            self.x.copy_(images, non_blocking=True)
            self.lab.copy_(labels, non_blocking=True)
        segs, totals, named, rest = self.graphs[key]
        handles = []
        main, side, used_side = torch.cuda.current_stream(self.dev), self.wq_stream, False
        for graph, kind, bi in segs:
            if kind == 'join':                 # the main chain needs what the side stream has produced so far
                main.wait_stream(side)
            elif graph is not None:
                if kind == 'side':             # a batch of weight gradients: beside the next segments of the main chain
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        graph.replay()
                    used_side = True
                else:
                    graph.replay()
            if bi is not None:             # this segment completed bucket bi: its all-reduce runs beside the next segments
                handles.append(self.reducer._all_reduce(self.reducer.flat[bi], side if used_side else None))
        for bi in rest:
            handles.append(self.reducer._all_reduce(self.reducer.flat[bi], side if used_side else None))
        if used_side:
            main.wait_stream(side)
        for h in handles:
            h.wait()
        if key[1] and self.overlap == 'after':
            self.reducer.reduce_all()
        self.totals, self.named = totals, named
        return totals, named

    def replay_local(self, images=None, labels=None):
        """the same micro-step WITHOUT its collectives (the (zero, no-reduce) graph, captured on first use): what the step costs when
        nothing is exchanged.  step-with-collectives minus this = the exposed communication time bench.py reports at N > 1."""
        if self.step_times != 1:
            raise RuntimeError('replay_local is defined for step_times == 1')
        key = (True, False)
        if key not in self.graphs:
            self._capture(key)
        if images is not None:
            self.x.copy_(images, non_blocking=True)
            self.lab.copy_(labels, non_blocking=True)
        segs, totals, named, _ = self.graphs[key]
        main, side, used_side = torch.cuda.current_stream(self.dev), self.wq_stream, False
        for graph, kind, _bi in segs:
            if kind == 'join':
                main.wait_stream(side)
            elif graph is not None and kind == 'side':
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    graph.replay()
                used_side = True
            elif graph is not None:
                graph.replay()
        if used_side:
            main.wait_stream(side)
        return totals, named

    def bucket_timeline(self):
        """one instrumented replay of the reducing step: [(bucket, MB, ms after the start of the step at which the segment that
        completes the bucket has finished - i.e. when its all-reduce can start), ...] and the step time; collectives are NOT issued
        (the timeline of the compute side).  Used by bench.py at N > 1 and by tools/bucket_timeline.py."""
        key = (True, True)
        if key not in self.graphs:
            self._capture(key)
        segs, _, _, rest = self.graphs[key]
        main, side, used_side = torch.cuda.current_stream(self.dev), self.wq_stream, False
        ev = lambda: torch.cuda.Event(enable_timing=True)
        e0, marks = ev(), []
        e0.record(main)
        for graph, kind, bi in segs:
            on = main
            if kind == 'join':
                main.wait_stream(side)
            elif graph is not None and kind == 'side':
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    graph.replay()
                used_side, on = True, side
            elif graph is not None:
                graph.replay()
            if bi is not None:
                e = ev()
                e.record(on)
                marks.append((bi, e))
        if used_side:
            main.wait_stream(side)
        e1 = ev()
        e1.record(main)
        e1.synchronize()
        total = e0.elapsed_time(e1)
        out = [(bi, self.reducer.flat[bi].numel() * 4 / 2 ** 20, e0.elapsed_time(e)) for bi, e in marks]
        out += [(bi, self.reducer.flat[bi].numel() * 4 / 2 ** 20, total) for bi in rest]
        return out, total


UNUSED_PARAMETERS = tuple(f'decode.bridge_list.4.transformer.pos_encoders.{n}.proj.{k}'
                          for n in range(1, 8) for k in ('weight', 'bias'))


def broadcast_parameters(model, comm=None, src=0):
    """one-time parameter sync from rank `src` (replaces DataParallel's per-step replicate)"""
    if comm is None:
        comm = _default_comm('broadcast_parameters')
    if comm.world > 1:
        for p in model.parameters():
            comm.broadcast(p.data, src)

"""torch.autograd.Function wrappers around the C-ABI kernels (include/ltu_hip.h).

Device tensors are channels-last: a reference tensor [B,C,H,W,D] is held as a contiguous
[B,H,W,D,C] tensor (fp32 or bf16).  PyTorch only provides memory, streams and the autograd
graph here; every arithmetic op is a HIP kernel from libltu_hip.so and there is no fallback.

Three host-side mechanisms keep the launch count down:
  * prepared weight operands (`prep=`): the model produces every cast / transposed / repacked weight of a
    step with ONE `ltu_weight_prep` launch; without `prep` an op prepares its own operands per call.
  * fused gradient accumulation: a parameter carrying `_ltu_grad` (an fp32 view into a flat gradient
    buffer, installed by `train.GradReducer`) receives its gradient straight from the weight-gradient
    kernels (`+=`), the Function returns None for it and calls `_ltu_hook(param)`.
  * `scratch_zeros`: small zero-initialised statistics buffers come from one arena that is cleared with
    a single fill per step instead of one `torch.zeros` launch each.

All of this host-side state lives in a `Context` (scratch arena, deferred second stages, weight-gradient branch, dropout step
counter, norm workspaces), never in module globals: every captured object (train.GraphedStep, infer.GraphedPredictor) and every
model's no-grad forward owns one, so two models, an evaluation between a training forward and its backward, or a graph
captured earlier cannot touch each other's buffers.  The context that is current when an op's forward runs (`ops.use(ctx)`,
thread-local) is remembered by the autograd node and used by its backward, which autograd runs on another thread.
"""
import contextlib
import ctypes
import threading

import torch

from . import _lib

F32, BF16 = 0, 1
ACT_NONE, ACT_LRELU = 0, 1
LRELU_SLOPE = 0.01
_c_void_p = ctypes.c_void_p


def _dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f'unsupported activation dtype {t.dtype}')


def _p(t):
    return 0 if t is None else t.data_ptr()


def _s():
    return torch.cuda.current_stream().cuda_stream


def _n(t):
    """capacity (elements) of a workspace tensor handed to the C-ABI beside its pointer; 0 for None"""
    return 0 if t is None else t.numel()


def _chk(t, name='tensor'):
    if not t.is_cuda:
        raise _lib.LtuError(f'{name} must live on the GPU: the HIP path has no CPU fallback')
    if not t.is_contiguous():
        raise _lib.LtuError(f'{name} must be contiguous')
    return t


def _ptr_array(tensors):
    arr = (_c_void_p * 3)()
    for i, t in enumerate(tensors):
        arr[i] = _p(t)
    return arr


# ---------------------------------------------------------------------------------------------- context

class _Arena:
    """Bump allocator over one fp32 buffer that is zero-filled once per step.  Once `frozen` (a HIP graph has captured addresses
    inside the buffer) the buffer is never replaced: requests that do not fit fall back to `torch.zeros`."""

    def __init__(self):
        self.buf = None
        self.off = 0
        self.need = 0
        self.frozen = False

    def begin_step(self, device):
        grow = self.need > 0 and (self.buf is None or self.buf.numel() < self.need or self.buf.device != device)
        if grow and not self.frozen:
            self.buf = torch.empty(int(self.need * 1.25) + 1024, device=device, dtype=torch.float32)
        if self.buf is not None:
            self.buf.zero_()
        self.off = 0
        self.need = 0

    def zeros(self, shape, device):
        n = 1
        for s in shape:
            n *= s
        n_al = (n + 63) // 64 * 64
        self.need += n_al
        if self.buf is not None and self.buf.device == device and self.off + n_al <= self.buf.numel():
            out = self.buf[self.off:self.off + n].view(shape)
            self.off += n_al
            return out
        return torch.zeros(shape, device=device, dtype=torch.float32)


class Context:
    """Host-side state of one stream of steps (see the module docstring)."""
    _DEFER_MAX = 8

    def __init__(self):
        self.arena = _Arena()
        self.step = None            # device-resident int64 step counter mixed into every dropout seed (common.h: make_drop)
        self.deferred = []          # (ReduceJob, workspace kept alive)
        self.wq = None              # weight-gradient queue (see wq_install)
        self.norm_ws_by_dev = {}
        self.ones = {}
        self.wg_group = []          # pending projection weight gradients of the current transformer layer(s) (see wgrad_group_push)
        self.wg_bytes = self.wg_layers = 0

    def one(self, device):
        """a cached fp32 scalar 1 (the seed gradient of each level loss: no fill launch per level and step)"""
        t = self.ones.get(device)
        if t is None:
            t = self.ones[device] = torch.ones((), device=device, dtype=torch.float32)
        return t

    # -- dropout step counter
    def set_step_counter(self, t):
        if t is not None and (t.dtype != torch.int64 or not t.is_cuda or t.numel() != 1):
            raise ValueError('step counter must be a 1-element int64 CUDA tensor')
        self.step = t

    def step_ptr(self):
        return 0 if self.step is None else self.step.data_ptr()

    # -- scratch arena
    def begin_step(self, device):
        """Recycles and clears the scratch arena: once per training step (train.train_step, inside a captured graph too) and at
        the start of a no-grad forward; never between a forward and its backward (saved statistics live in the arena)."""
        self.arena.begin_step(device)

    def scratch_zeros(self, shape, device):
        return self.arena.zeros(tuple(shape), device)

    def freeze(self):
        """a HIP graph now holds addresses inside the arena: it must never move again"""
        self.arena.frozen = True

    def norm_ws(self, device):
        """per-device scratch of the two-stage norm reductions (written and consumed inside one C call, so one buffer serves all)"""
        ws = self.norm_ws_by_dev.get(device)
        if ws is None:
            ws = self.norm_ws_by_dev[device] = torch.empty(_lib.load().ltu_norm_ws_floats(), device=device, dtype=torch.float32)
        return ws

    # -- deferred second stages: the two-stage reductions (projection weight gradients, LayerNorm gamma/beta gradients) may leave
    # their fold to `flush_deferred`, which folds up to 8 of them per launch.  Only gradients that live in a reducer's flat buffer
    # are deferred (nothing reads those before the end of backward / the bucket's all-reduce, both of which flush first).
    def defer_push(self, job, ws):
        if job.part:
            self.deferred.append((job, ws))
            if len(self.deferred) >= self._DEFER_MAX:
                self._fold_flush()

    # -- grouped projection weight gradients: the four projections of a transformer layer (qkv, out, ffn1, ffn2) hand their
    # weight-gradient jobs in here during the layer's backward; the layer's last one (qkv) flushes them as ONE launch + ONE fold
    # (ltu_linear_wgrad_group).  Only gradients that live in a reducer's flat buffer take part (nothing reads those before
    # flush_deferred, which also flushes a group left open).
    # At the small levels one layer's group is launch-latency-bound (2 048 tokens: 10 us + a 6 us fold for 13 MB of operands), so
    # the groups of consecutive layers of ONE transformer are kept back and launched together while their operands still fit
    # the last-level cache (WGRAD_DEFER_MB; the transformer's flush point, a closing gradient bucket and the end of backward
    # launch whatever is pending).
    def wgrad_group_push(self, g, x, dws, dbs, M, N, K, flush):
        if self.wg_group and self.wg_group[0][4] != M:
            self.wgrad_group_flush()                  # another transformer: a group shares one row split
        self.wg_group.append((g, x, dws, dbs, M, N, K))
        self.wg_bytes += 2 * M * (N + K)
        if len(self.wg_group) >= _lib.WGRAD_GROUP_MAX or (
                flush and (self.wg_bytes + self.wg_bytes // max(1, self.wg_layers + 1) > WGRAD_DEFER_MB * (1 << 20)
                           or len(self.wg_group) + 4 > _lib.WGRAD_GROUP_MAX)):
            self.wgrad_group_flush()
        elif flush:
            self.wg_layers += 1

    def wgrad_group_flush(self):
        jobs = self.wg_group
        if not jobs:
            return
        self.wg_group, self.wg_bytes, self.wg_layers = [], 0, 0
        self.wq_push(lambda keep: self._wgrad_group_launch(jobs, keep), [t for j in jobs for t in j[:2]], 'group')

    def side_width(self):
        """workgroup budget of the weight-gradient kernels that take one (ltu_linear_wgrad_group, ltu_upconv_wgrad): the width the
        owner of the weight-gradient queue asked for while a queue is installed (train.GraphedStep: half the machine, its side
        stream runs beside the main chain), 0 = the library's stand-alone default otherwise.  An ARGUMENT of the size query and of
        the launch - not a process-wide knob that the two could read at different values (round 4's workspace overrun)."""
        return 0 if self.wq is None else int(self.wq.get('width', 0))

    def _wgrad_group_launch(self, jobs, keep):
        arr = (_lib.WgradJob * len(jobs))()
        for r, (g, x, dws, dbs, M, N, K) in zip(arr, jobs):
            r.grad, r.a, r.ldg, r.lda, r.nw, r.M, r.N, r.K = g.data_ptr(), x.data_ptr(), N, K, len(dws), M, N, K
            for i, (dw, db) in enumerate(zip(dws, dbs)):
                r.dw[i], r.db[i] = dw.data_ptr(), db.data_ptr()
        lib = _lib.load()
        blocks = self.side_width()
        n = lib.ltu_linear_wgrad_group_ws_floats(ctypes.addressof(arr), len(jobs), blocks)
        if n > 0:
            ws = torch.empty(n, device=jobs[0][0].device, dtype=torch.float32)
            keep.append(ws)
            _lib.call('ltu_linear_wgrad_group', ctypes.addressof(arr), len(jobs), blocks, _p(ws), n, BF16, _s())
            return
        for g, x, dws, dbs, M, N, K in jobs:            # shapes the grouped kernel does not take: one call each
            wsb = _wgrad_ws(M, N, K, x)
            keep.append(wsb)
            _lib.call('ltu_linear_wgrad', _p(g), N, _p(x), K, _ptr_array(dws), _ptr_array(dbs), len(dws), M, N, K, _p(wsb), _n(wsb), 0,
                      _dt(x), _s())

    def _fold_flush(self):
        if not self.deferred:
            return
        jobs = list(self.deferred)
        self.deferred.clear()

        def launch(keep, jobs=jobs):
            arr = (_lib.ReduceJob * len(jobs))(*[j for j, _ in jobs])
            _lib.call('ltu_reduce_batch', ctypes.addressof(arr), len(jobs), _s())
        self.wq_push(launch, [ws for _, ws in jobs], 'fold')

    def flush_deferred(self):
        """fold every pending partial-sum workspace into its gradient and launch every weight gradient still held back
        (stream-ordered; call before gradients are consumed)"""
        self.wgrad_group_flush()
        self._fold_flush()
        self.wq_flush()

    # -- weight-gradient queue (round 4).  Weight gradients are consumed only at the end of the step (or by a bucket's all-reduce),
    # so they need not sit on the data-gradient chain: with a queue installed their launches are collected as closures and issued in
    # batches at the flush points (the input of every token transformer, a closing gradient bucket, the end of backward) on a SIDE
    # stream, where they run beside the latency-bound data-gradient kernels of the coarser levels.  train.GraphedStep captures each
    # batch as a linear graph of its own and replays it on the side stream between the linear segments of the main chain (`on_flush`;
    # no graph contains a fork: a fork inside a replayed graph costs more than it gains on ROCm 7).
    # Operands and workspaces of a batch stay alive until `wq_join` (the main stream has waited for the side stream): nothing
    # the main chain allocates meanwhile can land on memory a batch still reads.
    def wq_install(self, side_stream, on_flush=None, on_join=None, width=0):
        """side_stream None uninstalls.  on_flush(run): called instead of issuing a batch on the side stream (run() issues it on the
        current stream - the caller brackets it with its own capture); on_join(): called instead of making the current stream wait
        for the side stream (`side_join`); width: workgroup budget of the queued kernels that take one (`side_width`)"""
        self.wq = None if side_stream is None else {'side': side_stream, 'jobs': [], 'keep': [], 'on_flush': on_flush, 'on_join': on_join,
                                                    'running': False, 'batches': 0, 'side_pending': False, 'after': [], 'width': int(width)}

    # -- forward-side work on the same side stream: what the first kernels of the step do not need yet (the weight operands of
    # everything behind the encoder, the label pyramid) runs beside the encoder; `side_join` is placed in front of its first consumer
    def side_run(self, fn):
        q = self.wq
        if q is None or not WQ_SIDE_FWD:
            fn()
            return
        q['side_pending'] = True
        if q['on_flush'] is not None:
            q['on_flush'](fn)
            return
        side = q['side']
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()

    def side_join(self):
        q = self.wq
        if q is None or not q['side_pending']:
            return
        q['side_pending'] = False
        if q['on_join'] is not None:
            q['on_join']()
        else:
            torch.cuda.current_stream().wait_stream(q['side'])

    def wq_push(self, fn, tensors=(), fam=''):
        """fn(keep): launches weight-gradient kernels on the current stream, appending the workspaces it allocates to `keep`;
        fam: which family of weight gradients this is (WQ_INLINE lists families that are launched in line instead)"""
        q = self.wq
        if q is not None and not q['running'] and fam in WQ_INLINE:       # experiment: this family in line, at its stand-alone width
            w, q['width'] = q['width'], 0
            try:
                fn(q['keep'])
            finally:
                q['width'] = w
            return
        if q is None or q['running']:
            fn(q['keep'] if q is not None else [])
            return
        q['jobs'].append((fam, fn))
        q['keep'].extend(t for t in tensors if t is not None)
        if len(q['jobs']) >= WQ_MAX_JOBS:
            self.flush_deferred()

    def wq_flush(self):
        """issue the collected launches as one batch on the side stream, ordered behind everything issued so far on this stream"""
        q = self.wq
        if q is None or q['running'] or not (q['jobs'] or q['after']):
            return
        jobs, q['jobs'] = q['jobs'], []
        q['batches'] += 1
        if WQ_ORDER:       # experiment: families in a fixed order inside a batch (stable; e.g. the narrow grouped kernels first)
            jobs = sorted(jobs, key=lambda j: WQ_ORDER.index(j[0]) if j[0] in WQ_ORDER else len(WQ_ORDER))

        def run():
            q['running'] = True
            try:
                for _fam, fn in jobs:
                    fn(q['keep'])
                # parameters whose gradients were reported while the queue held their kernels back (train.GradReducer._hook parks
                # them here): NOW their gradients are enqueued - a bucket that this completes is reduced behind this batch
                after, q['after'] = q['after'], []
                for hook, p in after:
                    hook(p)
            finally:
                q['running'] = False
        if q['on_flush'] is not None:
            q['on_flush'](run)
            return
        side = q['side']
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            run()

    def wq_side(self):
        """the stream the queued weight gradients run on (None without a queue): a bucket's all-reduce has to wait for it too"""
        return None if self.wq is None else self.wq['side']

    def wq_join(self):
        """end of backward: launch what is left, make the current stream wait for the side stream, release the kept tensors"""
        q = self.wq
        if q is None:
            return
        self.flush_deferred()
        if q['on_flush'] is None:
            torch.cuda.current_stream().wait_stream(q['side'])
        q['keep'].clear()


_DEFAULT_CTX = Context()
_TLS = threading.local()


def current():
    """the context ops launched from this thread use (the process default unless inside `use`)"""
    return getattr(_TLS, 'ctx', None) or _DEFAULT_CTX


@contextlib.contextmanager
def use(ctx):
    prev = getattr(_TLS, 'ctx', None)
    _TLS.ctx = ctx
    try:
        yield ctx
    finally:
        _TLS.ctx = prev


def concurrent_stream(device, avoid, tries=8, cycles=1000000, priority=0):
    """A stream that REALLY runs beside every stream in `avoid`.  HIP maps its streams round-robin onto a few hardware queues
    (GPU_MAX_HW_QUEUES, 4 by default; 8 doubles the step time on this runtime) and two streams that share a queue serialise:
    every fourth stream of torch's pool lands on the current stream's queue (1.99x for those, 1.0x for the
    rest: profiles/HISTORY.md finding 34), which silently removed the whole benefit of the weight-gradient side stream whenever a communicator had taken a stream
    before it.  So candidates are probed: a spin kernel on each stream of `avoid` and on the candidate must take 1x, not 2x.
    Synchronises the device; call outside captures."""
    import time
    device = torch.device(device)

    def pair_ms(a, b):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        with torch.cuda.stream(a):
            torch.cuda._sleep(cycles)
        if b is not None:
            with torch.cuda.stream(b):
                torch.cuda._sleep(cycles)
        torch.cuda.synchronize(device)
        return time.perf_counter() - t0
    with torch.cuda.device(device):
        pair_ms(avoid[0], None)
        base = min(pair_ms(avoid[0], None) for _ in range(3))
        cands = []
        for _ in range(tries):
            c = torch.cuda.Stream(device=device, priority=priority)
            cands.append(c)
            if all(min(pair_ms(a, c) for _ in range(2)) < 1.5 * base for a in avoid):
                c.ltu_concurrent = True
                return c
    # no candidate ran beside every stream of `avoid` (a noisy probe on a shared GPU, or all hardware queues taken): the work on this
    # stream will serialise behind one of them - said loudly, and recorded on the stream for the caller's report
    import sys
    print(f'[lintransunet_amd] concurrent_stream: none of {tries} candidate streams ran beside the {len(avoid)} stream(s) to avoid '
          f'(a spin kernel pair took >= 1.5x one kernel): the side stream shares a hardware queue and will serialise', file=sys.stderr)
    cands[0].ltu_concurrent = False
    return cands[0]


# module-level conveniences over the current context
def set_step_counter(t):
    """Install (or clear with None) the device counter; advance it with `t.add_(1)` once per step, inside a captured graph too."""
    current().set_step_counter(t)


def begin_step(device):
    current().begin_step(device)


def scratch_zeros(shape, device):
    return current().scratch_zeros(shape, device)


def flush_deferred():
    current().flush_deferred()


# ---------------------------------------------------------------------------------------------- gradients of parameters

def _grad_buf(p):
    """fp32 buffer the weight-gradient kernels accumulate into, and whether it is the parameter's fused buffer"""
    tgt = getattr(p, '_ltu_grad', None)
    if tgt is not None:
        return tgt, True
    return torch.zeros(p.shape, device=p.device, dtype=torch.float32), False    # handed to autograd: must own its memory


def _grad_done(p, buf, fused):
    if fused:
        hook = getattr(p, '_ltu_hook', None)
        if hook is not None:
            hook(p)
        return None
    return buf


# ---------------------------------------------------------------------------------------------- weight operands

def _w_operand(w, dtype):
    """fp32 master weight -> GEMM operand in the activation dtype (same layout)"""
    if dtype == torch.float32:
        return w
    out = torch.empty(w.shape, device=w.device, dtype=dtype)
    _lib.call('ltu_cast_f32', _p(w), _p(out), w.numel(), BF16, _s())
    return out


def _w_transposed(ws, rows, cols, dtype):
    """cat(ws)^T as a GEMM operand: ws are [rows, cols] fp32 blocks -> [cols, len(ws)*rows] in the activation dtype"""
    n = rows * len(ws)
    wt = torch.empty((cols, n), device=ws[0].device, dtype=dtype)
    odt = F32 if dtype == torch.float32 else BF16
    for i, w in enumerate(ws):
        _lib.call('ltu_transpose_f32', _p(w), _p(wt), rows, cols, n, i * rows, odt, _s())
    return wt


DEFER_WGRAD = False
GROUP_WGRAD = True       # per-layer grouped projection weight gradients (ltu_linear_wgrad_group)
import os as _os
WGRAD_DEFER_MB = float(_os.environ.get('LTU_WGRAD_DEFER_MB', '400'))     # operand bytes of the layers' weight-gradient groups launched together
WQ_MAX_JOBS = int(_os.environ.get('LTU_WQ_JOBS', '1000000'))      # weight-gradient queue: a batch goes out when this many launches are queued
WQ_INLINE = frozenset(f for f in _os.environ.get('LTU_WQ_INLINE', '').split(',') if f)      # families of weight gradients kept off the queue
WQ_ORDER = [f for f in _os.environ.get('LTU_WQ_ORDER', '').split(',') if f]     # experiment: order of the families inside a batch
WQ_SIDE_FWD = _os.environ.get('LTU_WQ_FWD', '1') == '1'      # forward-side work (weight operands behind the encoder, label pyramid) on the side stream
WQ_FLUSH_IN_ENCODER = _os.environ.get('LTU_WQ_ENC', '1') == '1'      # ... and a batch per encoder block in the encoder's backward
WQ_SCHEDULE = _os.environ.get('LTU_WQ_SCHEDULE', 'end')     # weight-gradient queue: a batch where backward ENTERS ('start') / leaves ('end') a transformer
WGRAD_FLUSH_PER_LAYER = _os.environ.get('LTU_WGRAD_FLUSH', '') == 'layer'      # experiment: one edge per layer instead of per transformer
# tokens from which a transformer hands its weight gradients over LAYER BY LAYER (a batch behind every layer's backward instead of one
# behind the whole transformer): 0 = never
WGRAD_FLUSH_LAYER_TOKENS = int(_os.environ.get('LTU_WGRAD_FLUSH_TOKENS', '0'))


def flush_per_layer(tokens):
    return WGRAD_FLUSH_PER_LAYER or (WGRAD_FLUSH_LAYER_TOKENS > 0 and tokens >= WGRAD_FLUSH_LAYER_TOKENS)


def _defer_job():
    return _lib.ReduceJob()


class _WgradFlushPoint(torch.autograd.Function):
    """identity in forward; its backward runs when the gradient reaches this point, i.e. after everything downstream has
    run its backward.  kind 'in' (the input of a token transformer / an encoder block): the transformer's grouped weight gradients
    are handed over, and - schedule 'end' - the queue's batch goes out; kind 'out' (the OUTPUT of a token transformer, reached when
    backward ENTERS it) - schedule 'start' - the batch goes out there instead: everything queued so far (the previous, larger
    level's weight gradients) then runs beside this transformer's latency-bound layers rather than beside the machine-filling
    decoder kernels in front of it."""

    @staticmethod
    def forward(ctx, x, kind):
        ctx.lc, ctx.kind = current(), kind
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        lc = ctx.lc
        if ctx.kind == 'in':
            lc.wgrad_group_flush()
        if lc.wq is not None and (ctx.kind == 'enc' or WQ_SCHEDULE == 'both' or ctx.kind == ('in' if WQ_SCHEDULE == 'end' else 'out')):
            lc.wgrad_group_flush()
            lc._fold_flush()
            lc.wq_flush()
        return g, None


def wgrad_flush_point(x, kind='in'):
    if not x.requires_grad:
        return x
    if kind == 'out' and (current().wq is None or WQ_SCHEDULE == 'end'):
        return x
    return _WgradFlushPoint.apply(x, kind) if (current().wq is not None or WGRAD_DEFER_MB > 0) else x


def _wgrad_ws(M, N, K, like):
    """workspace for the two-stage (atomic-free) weight-gradient reduction of the bf16 path; None selects fp32 atomics"""
    if like.dtype != torch.bfloat16:
        return None
    n = _lib.load().ltu_wgrad_ws_floats(M, N, K)
    return torch.empty(n, device=like.device, dtype=torch.float32)


def _conv_ws(B, H, W, D, C, N, like):
    """workspace that lets a stride-1 bf16 conv over a small grid split its input channels (None: not needed)"""
    if like.dtype != torch.bfloat16:
        return None
    n = _lib.load().ltu_conv3d_ws_floats(B, H, W, D, C, N)
    return torch.empty(n, device=like.device, dtype=torch.float32) if n > 0 else None


class ConvPrep:
    """prepared operands of one 3x3x3 conv: wf [CoP][27][CiP], wd [CiP][27][CoP] (activation dtype), bias [CoP] fp32"""
    __slots__ = ('wf', 'wd', 'bias')

    def __init__(self, wf, wd, bias):
        self.wf, self.wd, self.bias = wf, wd, bias


class LinPrep:
    """prepared operands of a (group of) Linear / 1x1x1 conv weights: forward operands and cat(W)^T.
    group: None = the weight gradient is launched by the op's backward; 'collect' = handed to the context's weight-gradient group
    (a transformer layer's out / ffn projections); 'flush' = handed in and the group launched (the layer's qkv projection, whose
    backward is the last of the layer)."""
    __slots__ = ('w', 'wt', 'group', 'frag', 'fragT')

    def __init__(self, w, wt, group=None, frag=None, fragT=None):
        self.w, self.wt, self.group = w, wt, group
        self.frag = frag          # cat(W) in MFMA fragment order (weight-prep kind 8) for the row-block chain kernels, or None
        self.fragT = fragT        # W^T in fragment order (kind 9): the data-gradient operand of the backward chain kernel


# ---------------------------------------------------------------------------------------------- no-grad helpers

def window_embed(x, dtype):
    """[B,1,H,W,D] fp32 (reference layout) -> [B,H/2,W/2,D,8] channels-last, channels 4..7 zero."""
    _chk(x, 'x')
    B, _, H, W, D = x.shape
    y = torch.empty((B, H // 2, W // 2, D, 8), device=x.device, dtype=dtype)
    _lib.call('ltu_window_embed', _p(x), _p(y), _dt(y), B, H, W, D, _s())
    return y


def label_maxpool(lab, kd):
    """uint8 [B,H,W,D] -> max-pool (2,2,kd)."""
    B, H, W, D = lab.shape
    y = torch.empty((B, H // 2, W // 2, D // kd), device=lab.device, dtype=torch.uint8)
    _lib.call('ltu_label_maxpool', _p(lab), _p(y), B, H, W, D, kd, _s())
    return y


def onehot_argmax(p):
    """probabilities [..., C] fp32 -> one-hot of the arg-max class."""
    o = torch.empty_like(p)
    C = p.shape[-1]
    _lib.call('ltu_onehot_argmax', _p(p), _p(o), p.numel() // C, C, _s())
    return o


class RoiPlan:
    """Per-sample boxes and the separable sampling plans of one ROI bridge (device resident)."""

    def __init__(self, prob, roi_size, thr=0.5):
        _chk(prob, 'prob')
        B, H, W, D, C = prob.shape
        ni, nf, nh = ctypes.c_longlong(0), ctypes.c_longlong(0), ctypes.c_longlong(0)
        _lib.call('ltu_roi_plan_size', B, H, W, roi_size, ctypes.byref(ni), ctypes.byref(nf), ctypes.byref(nh))
        self.B, self.H, self.W, self.D, self.roi_size = B, H, W, D, roi_size
        self.eval_h = int(1.2 * roi_size)
        self.eval_w = int(self.eval_h * 0.6)
        self.up_h, self.up_w = 2 * ((self.eval_h + 1) // 2), 2 * ((self.eval_w + 1) // 2)
        self.ibuf = torch.empty(ni.value, device=prob.device, dtype=torch.int32)
        self.fbuf = torch.empty(nf.value, device=prob.device, dtype=torch.float32)
        self.box = torch.empty((B, 6), device=prob.device, dtype=torch.float32)
        hist = current().scratch_zeros((nh.value,), prob.device)          # zero bits: read as int32 by the kernels
        _lib.call('ltu_roi_plan', _p(prob), B, H, W, D, C, roi_size, float(thr), _p(self.box), _p(self.ibuf), _p(self.fbuf),
                  _p(hist), _s())


# ---------------------------------------------------------------------------------------------- conv / linear

class _Conv3d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0, x1, weight, bias, stride, ups, cop, prep):
        lc = ctx.lc = current()
        _chk(x0, 'x0')
        B, Hi, Wi, Di, C0 = x0.shape
        C1 = 0 if x1 is None else _chk(x1, 'x1').shape[-1]
        Co, Ci = weight.shape[0], weight.shape[1]
        CiP = C0 + C1
        assert CiP >= Ci and cop >= Co
        dev = x0.device
        if prep is None:
            wf = torch.empty((cop, 27, CiP), device=dev, dtype=x0.dtype)
            _lib.call('ltu_pack_conv_weight', _p(weight), _p(wf), 0, Co, Ci, cop, CiP, _dt(x0), _s())
            bias_p = bias
            if cop != Co:
                bias_p = torch.zeros(cop, device=dev, dtype=torch.float32)
                bias_p[:Co] = bias
        else:
            wf, bias_p = prep.wf, prep.bias
        sh, sw, sd = stride
        Hl, Wl, Dl = (2 * Hi, 2 * Wi, 2 * Di) if ups else (Hi, Wi, Di)
        Ho, Wo, Do = (Hl - 1) // sh + 1, (Wl - 1) // sw + 1, (Dl - 1) // sd + 1
        y = torch.empty((B, Ho, Wo, Do, cop), device=dev, dtype=x0.dtype)
        ws = _conv_ws(B, Ho, Wo, Do, CiP, cop, x0)
        _lib.call('ltu_conv3d_fwd', _p(x0), _p(x1), _p(wf), _p(bias_p), _p(y), B, Hi, Wi, Di, C0, C1, cop, sh, sw, sd,
                  int(ups), _p(ws), _n(ws), _dt(x0), _s())
        ctx.save_for_backward(x0, x1)
        ctx.params = (weight, bias)
        ctx.cfg = (stride, ups, cop, C0, C1, prep)
        return y

    @staticmethod
    def backward(ctx, g):
        lc = ctx.lc
        x0, x1 = ctx.saved_tensors
        weight, bias = ctx.params
        stride, ups, cop, C0, C1, prep = ctx.cfg
        g = g.contiguous()
        B, Hi, Wi, Di, _ = x0.shape
        Co, Ci = weight.shape[0], weight.shape[1]
        CiP = C0 + C1
        sh, sw, sd = stride
        dev = x0.device
        dt = _dt(x0)
        dx0 = dx1 = None
        need_dx = ctx.needs_input_grad[0] or (x1 is not None and ctx.needs_input_grad[1])
        if need_dx:
            if prep is None:
                wd = torch.empty((CiP, 27, cop), device=dev, dtype=x0.dtype)
                _lib.call('ltu_pack_conv_weight', _p(weight), 0, _p(wd), Co, Ci, cop, CiP, dt, _s())
            else:
                wd = prep.wd
            Hl, Wl, Dl = (2 * Hi, 2 * Wi, 2 * Di) if ups else (Hi, Wi, Di)
            d0 = torch.empty((B, Hl, Wl, Dl, C0), device=dev, dtype=x0.dtype)
            d1 = torch.empty((B, Hl, Wl, Dl, C1), device=dev, dtype=x0.dtype) if C1 else None
            ws = _conv_ws(B, Hl, Wl, Dl, cop, CiP, x0) if (sh, sw, sd) == (1, 1, 1) else None
            _lib.call('ltu_conv3d_dgrad', _p(g), _p(wd), _p(d0), _p(d1), B, Hl, Wl, Dl, C0, C1, cop, sh, sw, sd, _p(ws), _n(ws), dt, _s())
            if ups:
                dx0 = torch.empty_like(x0)
                _lib.call('ltu_sumpool2', _p(d0), _p(dx0), B, Hi, Wi, Di, C0, dt, _s())
            else:
                dx0 = d0
            dx1 = d1
        dw, fw = _grad_buf(weight)
        db, fb = _grad_buf(bias)
        # the weight gradient lands directly in the PyTorch layout [Co,Ci,3,3,3]; padded rows / channels are dropped
        def launch(keep):
            ws = _wgrad_ws(g.numel() // cop, cop, 27 * CiP, x0)
            keep.append(ws)
            _lib.call('ltu_conv3d_wgrad', _p(g), _p(x0), _p(x1), _p(dw), _p(db), B, Hi, Wi, Di, C0, C1, cop, sh, sw, sd, int(ups),
                      Co, Ci, _p(ws), _n(ws), dt, _s())
        if fw and fb:
            lc.wq_push(launch, (g, x0, x1), 'conv')          # fused gradient buffers: nothing reads them before the end of the step
        else:
            launch([])
        return dx0, dx1, _grad_done(weight, dw, fw), _grad_done(bias, db, fb), None, None, None, None


class PairPrep:
    """operands of two stride-1 convs fused over one input: wf [N0+N1][27][C], wd [C][27][N0+N1], bias [N0+N1] fp32"""
    __slots__ = ('wf', 'wd', 'bias', 'n0', 'n1', 'table')

    def __init__(self, wf, wd, bias, n0, n1, table=None):
        self.wf, self.wd, self.bias, self.n0, self.n1, self.table = wf, wd, bias, n0, n1, table


WPREP_DTYPE = [('src', '<u8'), ('dst', '<u8'), ('kind', '<i4'), ('R', '<i4'), ('C', '<i4'), ('p0', '<i4'), ('p1', '<i4'), ('pad', '<i4')]


def pair_records(wa, ba, wb, bb, n1, wf, wd, bias):
    """ltu_weight_prep records (src, dst, kind, R, C, p0, p1, pad) that fill the operands of a conv pair"""
    Ca, Ci = wa.shape[0], wa.shape[1]
    Cb = wb.shape[0]
    n0 = Ca
    esz = wf.element_size()
    return [
        (wa.data_ptr(), wf.data_ptr(), 2, Ca, Ci, n0, Ci, 0),
        (wb.data_ptr(), wf.data_ptr() + n0 * 27 * Ci * esz, 2, Cb, Ci, n1, Ci, 0),
        (wa.data_ptr(), wd.data_ptr(), 7, Ca, Ci, n0 + n1, Ci, (0 << 16) | n0),
        (wb.data_ptr(), wd.data_ptr(), 7, Cb, Ci, n0 + n1, Ci, (n0 << 16) | n1),
        (ba.data_ptr(), bias.data_ptr(), 4, 1, Ca, 0, 0, 0),
        (bb.data_ptr(), bias.data_ptr() + n0 * 4, 4, 1, Cb, 0, 0, 0),
    ]


def conv_pair_prep(wa, ba, wb, bb, n1, dtype):
    """stand-alone operand preparation of a conv pair (the model does this inside its one-launch weight store)"""
    import numpy as np
    Ca, Ci = wa.shape[0], wa.shape[1]
    dev = wa.device
    wf = torch.empty((Ca + n1, 27, Ci), device=dev, dtype=dtype)
    wd = torch.empty((Ci, 27, Ca + n1), device=dev, dtype=dtype)
    bias = torch.zeros(Ca + n1, device=dev, dtype=torch.float32)
    recs = pair_records(wa, ba, wb, bb, n1, wf, wd, bias)
    rec = np.zeros(len(recs), dtype=WPREP_DTYPE)
    for i, r in enumerate(recs):
        rec[i] = r
    table = torch.from_numpy(rec.view(np.uint8).copy()).to(dev)
    _lib.call('ltu_weight_prep', table.data_ptr(), len(recs), F32 if dtype == torch.float32 else BF16, _s())
    return PairPrep(wf, wd, bias, Ca, n1, table)


class _Conv3dPair(torch.autograd.Function):
    """two 3x3x3 stride-1 convs of the same input in one pass: (conv_a(x) [..,Ca], conv_b(x) padded to [..,n1])"""

    @staticmethod
    def forward(ctx, x, wa, ba, wb, bb, prep):
        lc = ctx.lc = current()
        _chk(x, 'x')
        B, H, W, D, C = x.shape
        n0, n1 = prep.n0, prep.n1
        y0 = torch.empty((B, H, W, D, n0), device=x.device, dtype=x.dtype)
        y1 = torch.empty((B, H, W, D, n1), device=x.device, dtype=x.dtype)
        ws = _conv_ws(B, H, W, D, C, n0 + n1, x)
        _lib.call('ltu_conv3d_pair_fwd', _p(x), _p(prep.wf), _p(prep.bias), _p(y0), _p(y1), B, H, W, D, C, n0, n1, _p(ws), _n(ws), _dt(x), _s())
        ctx.save_for_backward(x)
        ctx.params = (wa, ba, wb, bb)
        ctx.prep = prep
        return y0, y1

    @staticmethod
    def backward(ctx, g0, g1):
        lc = ctx.lc
        (x,) = ctx.saved_tensors
        wa, ba, wb, bb = ctx.params
        prep = ctx.prep
        B, H, W, D, C = x.shape
        n0, n1 = prep.n0, prep.n1
        dt = _dt(x)
        g0 = torch.zeros((B, H, W, D, n0), device=x.device, dtype=x.dtype) if g0 is None else g0.contiguous()
        g1 = torch.zeros((B, H, W, D, n1), device=x.device, dtype=x.dtype) if g1 is None else g1.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            ws = _conv_ws(B, H, W, D, n0 + n1, C, x)
            _lib.call('ltu_conv3d_pair_dgrad', _p(g0), _p(g1), _p(prep.wd), _p(dx), B, H, W, D, C, n0, n1, _p(ws), _n(ws), dt, _s())
        dwa, fwa = _grad_buf(wa)
        dba, fba = _grad_buf(ba)
        dwb, fwb = _grad_buf(wb)
        dbb, fbb = _grad_buf(bb)
        def launch(keep):
            ws = _wgrad_ws(g0.numel() // n0, n0 + n1, 27 * C, x)
            keep.append(ws)
            _lib.call('ltu_conv3d_pair_wgrad', _p(g0), _p(g1), _p(x), _p(dwa), _p(dba), _p(dwb), _p(dbb), B, H, W, D, C, n0, n1,
                      wa.shape[0], wb.shape[0], wa.shape[1], _p(ws), _n(ws), dt, _s())
        if fwa and fba and fwb and fbb:
            lc.wq_push(launch, (g0, g1, x), 'pair')
        else:
            launch([])
        outs = [_grad_done(wa, dwa, fwa), _grad_done(ba, dba, fba), _grad_done(wb, dwb, fwb), _grad_done(bb, dbb, fbb)]
        return (dx, *outs, None)


def conv3d_pair(x, wa, ba, wb, bb, prep):
    """(conv_a(x), conv_b(x) zero-padded to prep.n1 channels): both 3x3x3, stride 1, padding 1, computed in one pass"""
    return _Conv3dPair.apply(x, wa, ba, wb, bb, prep)


class UpConvPrep:
    """sub-pixel operands of a nearest-x2 + 3x3x3 conv: wf [8][Co][8][Ci], wd [Ci][64][Co] (activation dtype)"""
    __slots__ = ('wf', 'wd', 'table')

    def __init__(self, wf, wd, table=None):
        self.wf, self.wd, self.table = wf, wd, table


def upconv_prep(weight, dtype):
    """stand-alone operand preparation (the model prepares these in its one-launch weight store instead)"""
    import numpy as np
    Co, Ci = weight.shape[0], weight.shape[1]
    dev = weight.device
    wf = torch.empty((8, Co, 8, Ci), device=dev, dtype=dtype)
    wd = torch.empty((Ci, 64, Co), device=dev, dtype=dtype)
    rec = np.zeros(2, dtype=[('src', '<u8'), ('dst', '<u8'), ('kind', '<i4'), ('R', '<i4'), ('C', '<i4'), ('p0', '<i4'),
                             ('p1', '<i4'), ('pad', '<i4')])
    rec[0] = (weight.data_ptr(), wf.data_ptr(), 5, Co, Ci, Co, Ci, 0)
    rec[1] = (weight.data_ptr(), wd.data_ptr(), 6, Co, Ci, Co, Ci, 0)
    table = torch.from_numpy(rec.view(np.uint8).copy()).to(dev)
    _lib.call('ltu_weight_prep', table.data_ptr(), 2, F32 if dtype == torch.float32 else BF16, _s())
    return UpConvPrep(wf, wd, table)


class _UpConv3d(torch.autograd.Function):
    """conv3x3x3(nearest_upsample_x2(x)) as a sub-pixel convolution (csrc/upconv.hip)"""

    @staticmethod
    def forward(ctx, x, weight, bias, prep):
        lc = ctx.lc = current()
        _chk(x, 'x')
        B, H, W, D, Ci = x.shape
        Co = weight.shape[0]
        if prep is None:
            prep = upconv_prep(weight, x.dtype)
        y = torch.empty((B, 2 * H, 2 * W, 2 * D, Co), device=x.device, dtype=x.dtype)
        _lib.call('ltu_upconv_fwd', _p(x), _p(prep.wf), _p(bias), _p(y), B, H, W, D, Ci, Co, _dt(x), _s())
        ctx.save_for_backward(x)
        ctx.params = (weight, bias)
        ctx.prep = prep
        return y

    @staticmethod
    def backward(ctx, g):
        lc = ctx.lc
        (x,) = ctx.saved_tensors
        weight, bias = ctx.params
        prep = ctx.prep
        g = g.contiguous()
        B, H, W, D, Ci = x.shape
        Co = weight.shape[0]
        dt = _dt(x)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            nws = _lib.load().ltu_igemm_ws_floats(B * H * W * D, Ci, 64 * Co) if x.dtype == torch.bfloat16 else 0
            wsd = torch.empty(nws, device=x.device, dtype=torch.float32) if nws > 0 else None
            _lib.call('ltu_upconv_dgrad', _p(g), _p(prep.wd), _p(dx), B, H, W, D, Ci, Co, _p(wsd), _n(wsd), dt, _s())
        dw, fw = _grad_buf(weight)
        db, fb = _grad_buf(bias)
        dweff = lc.scratch_zeros((8, Co, 8, Ci), x.device)

        def launch(keep):
            ws, blocks = None, lc.side_width()
            if x.dtype == torch.bfloat16:
                ws = torch.empty(_lib.load().ltu_upconv_wgrad_ws_floats(B * H * W * D, Co, Ci, blocks), device=x.device, dtype=torch.float32)
                keep.append(ws)
            _lib.call('ltu_upconv_wgrad', _p(g), _p(x), _p(dweff), _p(db), _p(dw), Co, Ci, _p(ws), _n(ws), blocks, B, H, W, D, Ci, Co, dt,
                      _s())
        if fw and fb:
            lc.wq_push(launch, (g, x, dweff), 'upconv')      # dweff: read and written by the queued kernels (kept alive until the join)
        else:
            launch([])
        return dx, _grad_done(weight, dw, fw), _grad_done(bias, db, fb), None


def upconv3d(x, weight, bias, prep=None):
    """3x3x3 conv (padding 1) of the nearest-neighbour x2 upsampling of channels-last x, without materialising it."""
    return _UpConv3d.apply(x, weight, bias, prep)


def conv3d(x0, weight, bias, stride=(1, 1, 1), x1=None, ups=False, cop=None, prep=None):
    """3x3x3 conv, padding 1, on channels-last x0 (+ virtual concat x1).  Output has `cop` (>= Co, padded) channels."""
    return _Conv3d.apply(x0, x1, weight, bias, tuple(stride), bool(ups), cop or weight.shape[0], prep)


class _Linear(torch.autograd.Function):
    """y[M, sum N_i] = x[M,K] . cat(W_i)^T + cat(b_i);  W_i all [N/nw, K]."""

    @staticmethod
    def forward(ctx, x, prep, *wb):
        lc = ctx.lc = current()
        nw = len(wb) // 2
        ws, bs = wb[:nw], wb[nw:]
        _chk(x, 'x')
        M, K = x.shape
        Ns = ws[0].shape[0]
        N = Ns * nw
        y = torch.empty((M, N), device=x.device, dtype=x.dtype)
        wop = prep.w if prep is not None else [_w_operand(w, x.dtype) for w in ws]
        _lib.call('ltu_linear_fwd', _p(x), K, _ptr_array(wop), nw, _ptr_array(bs), _p(y), N, M, N, K, 0, _dt(x), _s())
        ctx.save_for_backward(x)
        ctx.params = (ws, bs)
        ctx.prep = prep
        return y

    @staticmethod
    def backward(ctx, g):
        lc = ctx.lc
        (x,) = ctx.saved_tensors
        ws, bs = ctx.params
        nw = len(ws)
        g = g.contiguous()
        M, K = x.shape
        Ns = ws[0].shape[0]
        N = Ns * nw
        dev, dt = x.device, _dt(x)
        dx = None
        if ctx.needs_input_grad[0]:
            wt = ctx.prep.wt if ctx.prep is not None else _w_transposed(ws, Ns, K, x.dtype)     # cat(W)^T
            dx = torch.empty((M, K), device=dev, dtype=x.dtype)
            _lib.call('ltu_linear_fwd', _p(g), N, _ptr_array([wt]), 1, _ptr_array([None]), _p(dx), K, M, K, N, 0, dt, _s())
        gw = [_grad_buf(w) for w in ws]
        gb = [_grad_buf(b) for b in bs]
        grp = ctx.prep.group if ctx.prep is not None else None
        if (grp is not None and GROUP_WGRAD and x.dtype == torch.bfloat16
                and all(f for _, f in gw) and all(f for _, f in gb)):
            lc.wgrad_group_push(g, x, [t for t, _ in gw], [t for t, _ in gb], M, N, K, grp == 'flush')
            dws = [_grad_done(w, t, f) for w, (t, f) in zip(ws, gw)]
            dbs = [_grad_done(b, t, f) for b, (t, f) in zip(bs, gb)]
            return (dx, None, *dws, *dbs)
        if lc.wq is not None and all(f for _, f in gw) and all(f for _, f in gb):
            def launch(keep):
                wsb = _wgrad_ws(M, N, K, x)
                keep.append(wsb)
                _lib.call('ltu_linear_wgrad', _p(g), N, _p(x), K, _ptr_array([t for t, _ in gw]), _ptr_array([t for t, _ in gb]), nw,
                          M, N, K, _p(wsb), _n(wsb), 0, dt, _s())
            lc.wq_push(launch, (g, x), 'lin')
            dws = [_grad_done(w, t, f) for w, (t, f) in zip(ws, gw)]
            dbs = [_grad_done(b, t, f) for b, (t, f) in zip(bs, gb)]
            return (dx, None, *dws, *dbs)
        wsb = _wgrad_ws(M, N, K, x)
        # Not deferred by default: 16 MB of partial tiles per projection are folded straight away while they still sit in
        # L2 / MALL (measured: a batched fold of 8 cold workspaces costs twice the 8 separate hot ones).
        job = _defer_job() if (DEFER_WGRAD and wsb is not None and all(f for _, f in gw) and all(f for _, f in gb)) else None
        _lib.call('ltu_linear_wgrad', _p(g), N, _p(x), K, _ptr_array([t for t, _ in gw]), _ptr_array([t for t, _ in gb]), nw,
                  M, N, K, _p(wsb), _n(wsb), ctypes.addressof(job) if job is not None else 0, dt, _s())
        if job is not None:
            lc.defer_push(job, wsb)
        dws = [_grad_done(w, t, f) for w, (t, f) in zip(ws, gw)]
        dbs = [_grad_done(b, t, f) for b, (t, f) in zip(bs, gb)]
        return (dx, None, *dws, *dbs)


def linear(x, weights, biases, prep=None):
    """x [M,K] (any leading dims flattened by the caller); weights: list of [N_i,K] (or [N_i,K,1,1,1])."""
    return _Linear.apply(x, prep, *weights, *biases)


class _LinearGelu(torch.autograd.Function):
    """h = dropout(gelu(x . W^T + b)): the FFN front half of a transformer layer (model/trans_block.py:203-208) in one launch
    where the projection kernel can carry GELU in its epilogue; backward = GELU backward, then the projection's backward."""

    @staticmethod
    def forward(ctx, x, prep, w, b, p, seed):
        lc = ctx.lc = current()
        _chk(x, 'x')
        M, K = x.shape
        N = w.shape[0]
        u = torch.empty((M, N), device=x.device, dtype=x.dtype)
        h = torch.empty((M, N), device=x.device, dtype=x.dtype)
        wop = prep.w[0] if prep is not None else _w_operand(w, x.dtype)
        _lib.call('ltu_linear_gelu_fwd', _p(x), K, _p(wop), _p(b), _p(u), _p(h), M, N, K, float(p), seed, lc.step_ptr(), _dt(x), _s())
        ctx.save_for_backward(x, u)
        ctx.params = (w, b)
        ctx.prep = prep
        ctx.cfg = (p, seed)
        return h

    @staticmethod
    def backward(ctx, gh):
        lc = ctx.lc
        x, u = ctx.saved_tensors
        w, b = ctx.params
        p, seed = ctx.cfg
        gh = gh.contiguous()
        M, K = x.shape
        N = w.shape[0]
        dev, dt = x.device, _dt(x)
        g = torch.empty_like(u)
        _lib.call('ltu_gelu_dropout_bwd', _p(gh), _p(u), _p(g), u.numel(), float(p), seed, lc.step_ptr(), dt, _s())
        dx = None
        if ctx.needs_input_grad[0]:
            wt = ctx.prep.wt if ctx.prep is not None else _w_transposed([w], N, K, x.dtype)
            dx = torch.empty((M, K), device=dev, dtype=x.dtype)
            _lib.call('ltu_linear_fwd', _p(g), N, _ptr_array([wt]), 1, _ptr_array([None]), _p(dx), K, M, K, N, 0, dt, _s())
        dw, fw = _grad_buf(w)
        db, fb = _grad_buf(b)

        def launch(keep):
            wsb = _wgrad_ws(M, N, K, x)
            keep.append(wsb)
            _lib.call('ltu_linear_wgrad', _p(g), N, _p(x), K, _ptr_array([dw]), _ptr_array([db]), 1, M, N, K, _p(wsb), _n(wsb), 0, dt, _s())
        grp = ctx.prep.group if ctx.prep is not None else None
        if grp is not None and GROUP_WGRAD and x.dtype == torch.bfloat16 and fw and fb:
            lc.wgrad_group_push(g, x, [dw], [db], M, N, K, grp == 'flush')
        elif fw and fb:
            lc.wq_push(launch, (g, x), 'lin')
        else:
            launch([])
        return dx, None, _grad_done(w, dw, fw), _grad_done(b, db, fb), None, None


def linear_gelu(x, weight, bias, p=0.0, seed=0, prep=None):
    """dropout(gelu(x . weight^T + bias)) for x [M,K]"""
    return _LinearGelu.apply(x, prep, weight, bias, p, seed)


# ---------------------------------------------------------------------------------------------- norms

def _ports(y, fork):
    """`fork` autograd outputs over one buffer: each consumer then delivers its own gradient tensor to backward, which hands all of
    them to ONE kernel that sums on load (no stand-alone add pass over an activation-sized tensor)"""
    return y if fork <= 1 else (y,) + tuple(y.view_as(y) for _ in range(fork - 1))


def _grads(gs):
    """the non-None incoming gradients (contiguous), padded with None to three"""
    out = [g.contiguous() for g in gs if g is not None]
    return out + [None] * (3 - len(out))


class _InstNormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, res_dup, act, p, seed, fork):
        lc = ctx.lc = current()
        _chk(x, 'x')
        B, C = x.shape[0], x.shape[-1]
        S = x.numel() // (B * C)
        sums = lc.scratch_zeros((B, C, 3), x.device)
        dt = _dt(x)
        y = torch.empty_like(x)
        _lib.call('ltu_instnorm_fwd', _p(x), _p(sums), _p(lc.norm_ws(x.device)), _n(lc.norm_ws(x.device)), _p(res), _p(y), B, S, C, act, LRELU_SLOPE, float(p), seed,
                  lc.step_ptr(), dt, _s())
        ctx.save_for_backward(x, sums)
        ctx.cfg = (act, p, seed, res is not None, res_dup is not None)
        return _ports(y, fork)

    @staticmethod
    def backward(ctx, *gs):
        lc = ctx.lc
        x, sums = ctx.saved_tensors
        act, p, seed, has_res, has_dup = ctx.cfg
        g, g2, g3 = _grads(gs)
        B, C = x.shape[0], x.shape[-1]
        S = x.numel() // (B * C)
        bsums = lc.scratch_zeros((B, C, 2), x.device)
        dx = torch.empty_like(x)
        _lib.call('ltu_instnorm_bwd', _p(g), _p(g2), _p(g3), _p(x), _p(sums), _p(bsums), _p(lc.norm_ws(x.device)), _n(lc.norm_ws(x.device)), _p(dx), B, S, C, act,
                  LRELU_SLOPE, float(p), seed, lc.step_ptr(), _dt(x), _s())
        # the residual passes the output gradient through.  With two output ports and a duplicate residual port the two gradient
        # tensors travel on separately (the producer of the residual sums them on load); otherwise they have to be added here.
        dres = ddup = None
        if has_res:
            if has_dup and g2 is not None and g3 is None:
                dres, ddup = g, g2
            else:
                dres = g if g2 is None else (g + g2 if g3 is None else g + g2 + g3)
        return dx, dres, ddup, None, None, None, None


def instnorm_act(x, res=None, act=ACT_LRELU, p=0.0, seed=0, fork=1, res_dup=None):
    """y = dropout(act(InstanceNorm(x))) + res over channels-last x [B,...,C].  fork > 1 returns that many ports of y (one per
    consumer); res_dup: a second port of the SAME residual tensor (values are read from `res`), which receives the gradient of
    the second output port."""
    return _InstNormAct.apply(x, res, res_dup, act, p, seed, fork)


class _ResLayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, r, gamma, beta, eps, p, seed, fork=False):
        lc = ctx.lc = current()
        _chk(x, 'x'); _chk(r, 'r')
        M, d = x.shape
        y = torch.empty_like(x)
        stat = torch.empty((M, 2), device=x.device, dtype=torch.float32)
        # r is overwritten with z = x + dropout(r): it is the producer's private output buffer
        _lib.call('ltu_layernorm_fwd', _p(x), _p(r), _p(gamma), _p(beta), _p(y), _p(stat), M, d, float(eps), float(p), seed,
                  lc.step_ptr(), _dt(x), _s())
        ctx.save_for_backward(r, stat)
        ctx.params = (gamma, beta)
        ctx.cfg = (p, seed)
        ctx.fork = fork
        if fork:
            # two autograd outputs over one buffer: the consumers' gradients then arrive separately and are summed inside the
            # backward kernel instead of by a stand-alone add pass
            return y, y.view_as(y)
        return y

    @staticmethod
    def backward(ctx, g, g2=None):
        lc = ctx.lc
        z, stat = ctx.saved_tensors
        gamma, beta = ctx.params
        p, seed = ctx.cfg
        if g is None:
            g, g2 = g2, None
        g = g.contiguous()
        if g2 is not None:
            g2 = g2.contiguous()
        M, d = z.shape
        dz = torch.empty_like(z)
        dr = torch.empty_like(z) if p > 0 else dz
        dgamma, fg = _grad_buf(gamma)
        dbeta, fb = _grad_buf(beta)
        job, ws = None, lc.norm_ws(g.device)
        if fg and fb:                      # gamma/beta gradients live in a reducer bucket: fold the partials later, batched
            job = _defer_job()
            ws = torch.empty(2048 * 2 * d, device=g.device, dtype=torch.float32)       # private: it outlives this call
        _lib.call('ltu_layernorm_bwd', _p(g), _p(g2), _p(z), _p(stat), _p(gamma), _p(dz), _p(dr), _p(dgamma), _p(dbeta),
                  _p(ws), _n(ws), ctypes.addressof(job) if job is not None else 0, M, d, float(p), seed, lc.step_ptr(), _dt(z), _s())
        if job is not None:
            lc.defer_push(job, ws)
        return dz, dr, _grad_done(gamma, dgamma, fg), _grad_done(beta, dbeta, fb), None, None, None, None


def res_layernorm(x, r, gamma, beta, eps=1e-6, p=0.0, seed=0, fork=False):
    """LayerNorm(x + dropout(r)) * gamma + beta over the last dim of [M,d]; consumes (overwrites) r.
    fork=True returns the result twice (for the projection and for the next residual)."""
    return _ResLayerNorm.apply(x, r, gamma, beta, eps, p, seed, fork)


class _GeluDropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, p, seed):
        lc = ctx.lc = current()
        _chk(u, 'u')
        h = torch.empty_like(u)
        _lib.call('ltu_gelu_dropout_fwd', _p(u), _p(h), u.numel(), float(p), seed, lc.step_ptr(), _dt(u), _s())
        ctx.save_for_backward(u)
        ctx.cfg = (p, seed)
        return h

    @staticmethod
    def backward(ctx, g):
        lc = ctx.lc
        (u,) = ctx.saved_tensors
        p, seed = ctx.cfg
        g = g.contiguous()
        du = torch.empty_like(u)
        _lib.call('ltu_gelu_dropout_bwd', _p(g), _p(u), _p(du), u.numel(), float(p), seed, lc.step_ptr(), _dt(u), _s())
        return du, None, None


def gelu_dropout(u, p=0.0, seed=0):
    return _GeluDropout.apply(u, p, seed)


# ---------------------------------------------------------------------------------------------- transformer layer tail
# the row-block chain kernel wins at every token level of the 128^3 step (micro-benchmark tools/bench_tail.py: 20 / 35 / 58 / 120 us
# against 27 / 51 / 72 / 130 us for the five launches at 1 024 / 8 640 / 21 504 / 114 816 tokens); LTU_TAIL_MAX_TOKENS caps it
USE_LAYER_TAIL = _os.environ.get('LTU_NO_LAYER_TAIL', '') == ''
USE_LAYER_TAIL_BWD = _os.environ.get('LTU_NO_LAYER_TAIL_BWD', '') == ''
TAIL_MAX_TOKENS = int(_os.environ.get('LTU_TAIL_MAX_TOKENS', '1000000000'))      # no cap by default: the chain kernels serve every level


def _wgrad_now_or_group(lc, g, x, ws, bs, M, N, K):
    """weight / bias gradients of y = x W^T + b given g = dL/dy: into the layer's group when the gradients are fused buffers,
    otherwise computed here; returns the values autograd expects (None for fused buffers)"""
    gw = [_grad_buf(w) for w in ws]
    gb = [_grad_buf(b) for b in bs]
    dws, dbs = [t for t, _ in gw], [t for t, _ in gb]
    if GROUP_WGRAD and all(f for _, f in gw) and all(f for _, f in gb):
        lc.wgrad_group_push(g, x, dws, dbs, M, N, K, False)
    else:
        wsb = _wgrad_ws(M, N, K, x)
        _lib.call('ltu_linear_wgrad', _p(g), N, _p(x), K, _ptr_array(dws), _ptr_array(dbs), len(ws), M, N, K,
                  _p(wsb), _n(wsb), 0, _dt(x), _s())
    return [_grad_done(w, t, f) for w, (t, f) in zip(ws, gw)], [_grad_done(b, t, f) for b, (t, f) in zip(bs, gb)]


FUSE_ATTN_APPLY = True      # phase B of the linear attention inside the forward chain kernel (tests flip it to compare)
FUSE_ATTN_MAX_TOKENS = int(_os.environ.get('LTU_FUSE_ATTN_MAX_TOKENS', '1000000000'))      # experiment: stand-alone phase B above this many tokens
FUSE_NEXT_QKV = _os.environ.get('LTU_NO_FUSE_QKV', '') == ''      # the next layer's q|k|v projection at the end of the forward chain kernel
FUSE_QKV_MAX_TOKENS = int(_os.environ.get('LTU_FUSE_QKV_MAX_TOKENS', '50000'))      # measured: pays at the d = 256 levels (-0.05 ms), costs as much at 114 816 x 128
FUSE_QKV_MIN_TOKENS = int(_os.environ.get('LTU_FUSE_QKV_MIN_TOKENS', '0'))


class _LayerTail(torch.autograd.Function):
    """the post-attention half of a transformer layer (model/trans_block.py:203-211) as one launch (csrc/tlayer.hip):
    y = LN2(t1 + drop(W2 drop(gelu(W1 t1)))),  t1 = LN1(x + drop(Wo a)).  Backward: the op-by-op kernels on the saved tensors."""

    @staticmethod
    def forward(ctx, a, x, wo, bo, w1, b1, w2, b2, g1, be1, g2, be2, preps, eps, p, seeds, fork, attn, nq0=None, nq1=None, nq2=None,
                nb0=None, nb1=None, nb2=None, nprep=None):
        lc = ctx.lc = current()
        _chk(a, 'a'); _chk(x, 'x')
        M, d = x.shape
        dev, dt = x.device, x.dtype
        po, p1, p2 = preps
        qkv = cx = colstats = qstat = None
        if attn is not None:
            # `a` is the fused projection output qkv [M, 3d]: phase A of the linear attention here, phase B inside the chain
            # kernel (its row blocks read their q rows and apply the merged context), which also writes the attention output
            B, N = attn
            qkv, H = a, d // 32
            nsplit = _lib.load().ltu_linattn_splits(B, N)
            cx = torch.empty((B * H, 32, 32), device=dev, dtype=torch.float32)
            colstats = torch.empty((B * H, 64), device=dev, dtype=torch.float32)
            qstat = torch.empty((M, H, 2), device=dev, dtype=torch.float32)
            ws = torch.empty(_lib.load().ltu_linattn_ws_floats(B, N, d), device=dev, dtype=torch.float32)
            _lib.call('ltu_linattn_ctx', _p(qkv), _p(cx), _p(colstats), _p(ws), _n(ws), B, N, d, _dt(qkv), _s())
            a = torch.empty((M, d), device=dev, dtype=dt)
        z1, t1, z2, y = (torch.empty((M, d), device=dev, dtype=dt) for _ in range(4))
        u, h = (torch.empty((M, 2 * d), device=dev, dtype=dt) for _ in range(2))
        stat1, stat2 = (torch.empty((M, 2), device=dev, dtype=torch.float32) for _ in range(2))
        # the NEXT layer's q | k | v projection inside this launch (nprep: its LinPrep with the fragment-ordered cat(Wq, Wk, Wv))
        qkv_next = torch.empty((M, 3 * d), device=dev, dtype=dt) if nprep is not None else None
        # u_mode 1: the kernel leaves dropout_mask * gelu'(u) in `u` - all the backward chain kernel needs of it; the op-by-op
        # backward (no transposed fragments, or switched off) wants the pre-activation itself
        u_mode = 1 if (USE_LAYER_TAIL_BWD and po.fragT is not None and p1.fragT is not None and p2.fragT is not None) else 0
        _lib.call('ltu_layer_tail_fwd', _p(a), _p(x), _p(po.frag), _p(p1.frag), _p(p2.frag), _p(bo), _p(b1), _p(b2), _p(g1), _p(be1),
                  _p(g2), _p(be2), _p(z1), _p(t1), _p(u), _p(h), _p(z2), _p(y), _p(stat1), _p(stat2), M, d, float(eps), float(p),
                  seeds[0], seeds[1], seeds[2], lc.step_ptr(), u_mode, _p(qkv), _p(cx), _p(qstat), attn[1] if attn else 0,
                  _p(nprep.frag) if nprep is not None else 0, _p(nb0), _p(nb1), _p(nb2), _p(qkv_next), _dt(a), _s())
        ctx.save_for_backward(a, z1, stat1, t1, u, h, z2, stat2, y if nprep is not None else None)
        ctx.params = (wo, bo, w1, b1, w2, b2, g1, be1, g2, be2)
        ctx.cfg = (preps, p, seeds)
        ctx.u_mode = u_mode
        ctx.attn = None if attn is None else (attn, qkv, cx, colstats, qstat)
        ctx.nxt = None if nprep is None else ((nq0, nq1, nq2), (nb0, nb1, nb2), nprep)
        if nprep is not None:
            return y, qkv_next              # y has one consumer left (the next layer's residual): no second port
        return (y, y.view_as(y)) if fork else y

    @staticmethod
    def backward(ctx, g, g2=None):
        lc = ctx.lc
        a, z1, stat1, t1, u, h, z2, stat2, y = ctx.saved_tensors
        wo, bo, w1, b1, w2, b2, gm1, be1, gm2, be2 = ctx.params
        (po, p1, p2), p, seeds = ctx.cfg
        M, d = a.shape
        dev, dt = a.device, _dt(a)
        nxt_grads = (None,) * 7
        if ctx.nxt is not None:
            # second output = the next layer's qkv: its gradient g2 = dqkv goes through that projection here - the data gradient
            # becomes the second gradient of y, the weight gradient closes the next layer's group (its last backward op)
            (nws, nbs, nprep), gq = ctx.nxt, g2
            g2 = None
            if gq is not None:
                gq = gq.contiguous()
                g2 = torch.empty((M, d), device=dev, dtype=a.dtype)
                _lib.call('ltu_linear_fwd', _p(gq), 3 * d, _ptr_array([nprep.wt]), 1, _ptr_array([None]), _p(g2), d, M, d, 3 * d, 0, dt, _s())
                gw = [_grad_buf(w) for w in nws]
                gb = [_grad_buf(b) for b in nbs]
                dws, dbs = [t for t, _ in gw], [t for t, _ in gb]
                if GROUP_WGRAD and all(f for _, f in gw) and all(f for _, f in gb):
                    lc.wgrad_group_push(gq, y, dws, dbs, M, 3 * d, d, True)
                else:
                    wsq = _wgrad_ws(M, 3 * d, d, y)
                    _lib.call('ltu_linear_wgrad', _p(gq), 3 * d, _p(y), d, _ptr_array(dws), _ptr_array(dbs), 3, M, 3 * d, d,
                              _p(wsq), _n(wsq), 0, dt, _s())
                nxt_grads = tuple(_grad_done(w, t, f) for w, (t, f) in zip(nws, gw)) + \
                    tuple(_grad_done(b, t, f) for b, (t, f) in zip(nbs, gb)) + (None,)
        if g is None:
            g, g2 = g2, None
        g = g.contiguous()
        if g2 is not None:
            g2 = g2.contiguous()

        def ln_bwd(gy, gy2, z, stat, gamma, beta, seed):
            dz = torch.empty_like(z)
            dr = torch.empty_like(z) if p > 0 else dz
            dg, fg = _grad_buf(gamma)
            db, fb = _grad_buf(beta)
            job, ws = None, lc.norm_ws(dev)
            if fg and fb:
                job = _defer_job()
                ws = torch.empty(2048 * 2 * d, device=dev, dtype=torch.float32)
            _lib.call('ltu_layernorm_bwd', _p(gy), _p(gy2), _p(z), _p(stat), _p(gamma), _p(dz), _p(dr), _p(dg), _p(db), _p(ws), _n(ws),
                      ctypes.addressof(job) if job is not None else 0, M, d, float(p), seed, lc.step_ptr(), dt, _s())
            if job is not None:
                lc.defer_push(job, ws)
            return dz, dr, _grad_done(gamma, dg, fg), _grad_done(beta, db, fb)

        def dgrad(gy, prep, w, N, K):        # gy [M,N] . W [N,K] -> [M,K]
            wt = prep.wt if prep is not None else _w_transposed([w], N, K, a.dtype)
            dx = torch.empty((M, K), device=dev, dtype=a.dtype)
            _lib.call('ltu_linear_fwd', _p(gy), N, _ptr_array([wt]), 1, _ptr_array([None]), _p(dx), K, M, K, N, 0, dt, _s())
            return dx

        if ctx.u_mode == 1:
            # the whole data-gradient chain as one launch (csrc/tlayer.hip: tail_bwd_kernel)
            dr2, dr1, dz1, da = (torch.empty_like(a) for _ in range(4))
            du = torch.empty_like(u)
            nblk = _lib.load().ltu_layer_tail_blocks(M)
            lnws = torch.empty((2, nblk, 2 * d), device=dev, dtype=torch.float32)
            _lib.call('ltu_layer_tail_bwd', _p(g), _p(g2), _p(z2), _p(z1), _p(u), _p(stat2), _p(stat1), _p(gm2), _p(gm1), _p(p2.fragT),
                      _p(p1.fragT), _p(po.fragT), _p(dr2), _p(du), _p(dr1), _p(dz1), _p(da), _p(lnws[0]), _p(lnws[1]), _n(lnws[0]), M, d, float(p),
                      seeds[0], seeds[1], seeds[2], lc.step_ptr(), 1, dt, _s())
            outs = []
            for k, (gamma, beta) in enumerate(((gm2, be2), (gm1, be1))):
                dg, fg = _grad_buf(gamma)
                db, fb = _grad_buf(beta)
                job = _defer_job()
                job.part, job.nsplit, job.n, job.k, job.nseg, job.mode = lnws[k].data_ptr(), nblk, 2 * d, 1, 1, 1
                job.out[0], job.out[1] = dg.data_ptr(), db.data_ptr()
                if fg and fb:
                    lc.defer_push(job, lnws)                  # folded with the other pending second stages, 8 per launch
                else:
                    _lib.call('ltu_reduce_batch', ctypes.addressof(job), 1, _s())
                outs.append((_grad_done(gamma, dg, fg), _grad_done(beta, db, fb)))
            (dgm2, dbe2), (dgm1, dbe1) = outs
            (dw2,), (db2,) = _wgrad_now_or_group(lc, dr2, h, [w2], [b2], M, d, 2 * d)
            (dw1,), (db1,) = _wgrad_now_or_group(lc, du, t1, [w1], [b1], M, 2 * d, d)
            (dwo,), (dbo,) = _wgrad_now_or_group(lc, dr1, a, [wo], [bo], M, d, d)
            return (_LayerTail._attn_bwd(ctx, da), dz1, dwo, dbo, dw1, db1, dw2, db2, dgm1, dbe1, dgm2, dbe2, None, None, None, None, None, None) + (nxt_grads if ctx.nxt is not None else ())
        dz2, dr2, dgm2, dbe2 = ln_bwd(g, g2, z2, stat2, gm2, be2, seeds[2])
        dh = dgrad(dr2, p2, w2, d, 2 * d)
        (dw2,), (db2,) = _wgrad_now_or_group(lc, dr2, h, [w2], [b2], M, d, 2 * d)
        du = torch.empty_like(u)
        _lib.call('ltu_gelu_dropout_bwd', _p(dh), _p(u), _p(du), u.numel(), float(p), seeds[1], lc.step_ptr(), dt, _s())
        dt1 = dgrad(du, p1, w1, 2 * d, d)
        (dw1,), (db1,) = _wgrad_now_or_group(lc, du, t1, [w1], [b1], M, 2 * d, d)
        dz1, dr1, dgm1, dbe1 = ln_bwd(dt1, dz2, z1, stat1, gm1, be1, seeds[0])
        da = dgrad(dr1, po, wo, d, d)
        (dwo,), (dbo,) = _wgrad_now_or_group(lc, dr1, a, [wo], [bo], M, d, d)
        return (_LayerTail._attn_bwd(ctx, da), dz1, dwo, dbo, dw1, db1, dw2, db2, dgm1, dbe1, dgm2, dbe2, None, None, None, None, None, None) + (nxt_grads if ctx.nxt is not None else ())

    @staticmethod
    def _attn_bwd(ctx, da):
        """gradient of the first input: da itself, or (fused attention) dqkv from the attention core's backward pass"""
        if ctx.attn is None:
            return da
        (B, N), qkv, cx, colstats, qstat = ctx.attn
        d = da.shape[1]
        H = d // 32
        nsplit = _lib.load().ltu_linattn_splits(B, N)
        dqkv = torch.empty_like(qkv)
        dctx = torch.empty_like(cx)
        ws = torch.empty(_lib.load().ltu_linattn_ws_floats(B, N, d), device=da.device, dtype=torch.float32)
        _lib.call('ltu_linattn_bwd', _p(qkv), _p(da), _p(cx), _p(colstats), _p(qstat), _p(dqkv), _p(dctx), 0, _p(ws), _n(ws), B, N, d,
                  _dt(qkv), _s())
        return dqkv


def layer_tail(a, x, lay_params, preps, eps, p, seeds, fork, attn=None, nxt=None):
    """a: attention output [M,d], x: layer input [M,d]; lay_params = (Wo, bo, W1, b1, W2, b2, g1, be1, g2, be2);
    preps = LinPrep of (out, linear1, linear2) with fragment-ordered operands; seeds = dropout sites (LN1, GELU, LN2).
    attn = (B, N): `a` is the fused q|k|v projection [M, 3d] instead and the linear attention runs in front of the chain (its
    phase B inside the chain kernel); the gradient returned for `a` is then dqkv.
    nxt = ([Wq, Wk, Wv], [bq, bk, bv], LinPrep with .frag) of the NEXT layer: its q|k|v projection runs at the end of this launch
    and the call returns (y, qkv_next)."""
    if nxt is not None:
        (q0, q1, q2), (c0, c1, c2), nprep = nxt
        return _LayerTail.apply(a, x, *lay_params, preps, eps, p, tuple(seeds), fork, attn, q0, q1, q2, c0, c1, c2, nprep)
    return _LayerTail.apply(a, x, *lay_params, preps, eps, p, tuple(seeds), fork, attn)


# ---------------------------------------------------------------------------------------------- linear attention

class _LinAttn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, B, N, d):
        lc = ctx.lc = current()
        _chk(qkv, 'qkv')
        H = d // 32
        dev = qkv.device
        nsplit = _lib.load().ltu_linattn_splits(B, N)
        out = torch.empty((B * N, d), device=dev, dtype=qkv.dtype)
        cx = torch.empty((B * H, 32, 32), device=dev, dtype=torch.float32)
        colstats = torch.empty((B * H, 64), device=dev, dtype=torch.float32)
        qstat = torch.empty((B * N, H, 2), device=dev, dtype=torch.float32)
        ws = torch.empty(_lib.load().ltu_linattn_ws_floats(B, N, d), device=dev, dtype=torch.float32)
        _lib.call('ltu_linattn_fwd', _p(qkv), _p(out), _p(cx), _p(colstats), _p(qstat), _p(ws), _n(ws), B, N, d, _dt(qkv), _s())
        ctx.save_for_backward(qkv, cx, colstats, qstat)
        ctx.cfg = (B, N, d, nsplit)
        return out

    @staticmethod
    def backward(ctx, g):
        lc = ctx.lc
        qkv, cx, colstats, qstat = ctx.saved_tensors
        B, N, d, nsplit = ctx.cfg
        H = d // 32
        g = g.contiguous()
        dev = qkv.device
        dqkv = torch.empty_like(qkv)
        dctx = torch.empty_like(cx)
        tvec = torch.empty((B * H, 32), device=dev, dtype=torch.float32)
        ws = torch.empty(_lib.load().ltu_linattn_ws_floats(B, N, d), device=dev, dtype=torch.float32)
        _lib.call('ltu_linattn_bwd', _p(qkv), _p(g), _p(cx), _p(colstats), _p(qstat), _p(dqkv), _p(dctx), _p(tvec), _p(ws), _n(ws),
                  B, N, d, _dt(qkv), _s())
        return dqkv, None, None, None


def linear_attention(qkv, B, N, d):
    """qkv [B*N, 3d] (q|k|v, 32-wide heads) -> attention output [B*N, d]."""
    return _LinAttn.apply(qkv, B, N, d)


# ---------------------------------------------------------------------------------------------- stencils / resampling

class _PosConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, p, seed, fork):
        lc = ctx.lc = current()
        _chk(x, 'x')
        B, H, W, D, C = x.shape
        y = torch.empty_like(x)
        _lib.call('ltu_dwconv_fwd', _p(x), _p(w), _p(b), _p(y), B, H, W, D, C, float(p), seed, lc.step_ptr(), _dt(x), _s())
        ctx.save_for_backward(x)
        ctx.params = (w, b)
        ctx.cfg = (p, seed)
        return _ports(y, fork)

    @staticmethod
    def backward(ctx, *gs):
        lc = ctx.lc
        (x,) = ctx.saved_tensors
        w, b = ctx.params
        p, seed = ctx.cfg
        g, g2, _ = _grads(gs)
        B, H, W, D, C = x.shape
        dx = torch.empty_like(x)
        dw, fw = _grad_buf(w)
        db, fb = _grad_buf(b)
        _lib.call('ltu_dwconv_bwd', _p(g), _p(g2), _p(x), _p(w), _p(dx), 0, 0, 0, 0, B, H, W, D, C, float(p), seed,
                  lc.step_ptr(), _dt(x), _s())

        def launch(keep):                    # the weight / bias gradient: off the data-gradient chain
            ws = torch.empty(_lib.load().ltu_dwconv_bwd_ws_floats(B, H, W, D, C, _dt(x)), device=x.device, dtype=torch.float32)
            keep.append(ws)
            _lib.call('ltu_dwconv_bwd', _p(g), _p(g2), _p(x), _p(w), 0, _p(dw), _p(db), _p(ws), _n(ws), B, H, W, D, C, float(p), seed,
                      lc.step_ptr(), _dt(x), _s())
        if fw and fb:
            lc.wq_push(launch, (g, g2, x), 'dw')
        else:
            launch([])
        return dx, _grad_done(w, dw, fw), _grad_done(b, db, fb), None, None, None


def pos_conv(x, w, b, p=0.0, seed=0, fork=1):
    """chan_dropout(x + depthwise3x3x3(x) + b) on channels-last x; w [C,1,3,3,3] in the reference's (D,H,W) kernel order.
    fork = 2 returns two ports of the result (one per consumer; their gradients are summed inside the backward kernel)."""
    return _PosConv.apply(x, w, b, p, seed, fork)


SEPARABLE_TRILINEAR_ADJOINT = True      # three 1-D passes instead of the 64-tap gather (tests flip it to compare)


class _Trilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, sd, fork):
        lc = ctx.lc = current()
        _chk(x, 'x')
        B, H, W, D, C = x.shape
        y = torch.empty((B, 2 * H, 2 * W, sd * D, C), device=x.device, dtype=x.dtype)
        _lib.call('ltu_trilinear_up', _p(x), 0, _p(y), 0, B, H, W, D, C, sd, _dt(x), _s())
        ctx.cfg = (B, H, W, D, C, sd)
        return _ports(y, fork)

    @staticmethod
    def backward(ctx, *gs):
        lc = ctx.lc
        B, H, W, D, C, sd = ctx.cfg
        g, g2, _ = _grads(gs)
        dx = torch.empty((B, H, W, D, C), device=g.device, dtype=g.dtype)
        if SEPARABLE_TRILINEAR_ADJOINT and (C % 8 == 0 or g.dtype == torch.float32):
            ws = torch.empty(_lib.load().ltu_trilinear_adjoint_ws_elems(B, H, W, D, C, sd), device=g.device, dtype=g.dtype)
            _lib.call('ltu_trilinear_adjoint', _p(g), _p(g2), _p(dx), _p(ws), _n(ws), B, H, W, D, C, sd, _dt(g), _s())
        else:
            _lib.call('ltu_trilinear_up', _p(g), _p(g2), _p(dx), 1, B, H, W, D, C, sd, _dt(g), _s())
        return dx, None, None


def trilinear_up(x, sd, fork=1):
    """Trilinear x(2,2,sd) upsampling, align_corners=True; fork = 2 returns two ports (one per consumer)."""
    return _Trilinear.apply(x, sd, fork)


class _RoiResample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, plan, which):
        lc = ctx.lc = current()
        _chk(x, 'x')
        B, C = x.shape[0], x.shape[-1]
        D = x.shape[3]
        if which == 0:
            shape = (B, plan.eval_h, plan.eval_w, D, C)
        else:
            shape = (B, plan.H, plan.W, D, C)
        y = torch.empty(shape, device=x.device, dtype=x.dtype)
        _lib.call('ltu_roi_resample', _p(x), _p(y), _p(plan.ibuf), _p(plan.fbuf), which, 0, B, plan.H, plan.W, D, C,
                  plan.roi_size, _dt(x), _s())
        ctx.plan, ctx.which, ctx.in_shape = plan, which, tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, g):
        lc = ctx.lc
        plan, which = ctx.plan, ctx.which
        g = g.contiguous()
        B, _, _, D, C = ctx.in_shape
        dx = torch.empty(ctx.in_shape, device=g.device, dtype=g.dtype)
        _lib.call('ltu_roi_resample', _p(g), _p(dx), _p(plan.ibuf), _p(plan.fbuf), which, 1, B, plan.H, plan.W, D, C,
                  plan.roi_size, _dt(g), _s())
        return dx, None, None


def roi_warp(x, plan):
    """image lattice -> fixed ROI grid (roi_alignment2)."""
    return _RoiResample.apply(x, plan, 0)


def roi_unwarp(x, plan):
    """ROI grid -> image lattice (post_processing2)."""
    return _RoiResample.apply(x, plan, 1)


# ---------------------------------------------------------------------------------------------- heads

class _HeadSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, C):
        lc = ctx.lc = current()
        _chk(z, 'z')
        CP = z.shape[-1]
        M = z.numel() // CP
        p = torch.empty(z.shape[:-1] + (C,), device=z.device, dtype=torch.float32)
        _lib.call('ltu_head_softmax_fwd', _p(z), _p(p), M, C, CP, _dt(z), _s())
        ctx.save_for_backward(p)
        ctx.cfg = (C, CP, z.dtype)
        return p

    @staticmethod
    def backward(ctx, g):
        lc = ctx.lc
        (p,) = ctx.saved_tensors
        C, CP, zdt = ctx.cfg
        g = g.contiguous()
        dz = torch.empty(p.shape[:-1] + (CP,), device=p.device, dtype=zdt)
        _lib.call('ltu_head_softmax_bwd', _p(g), _p(p), _p(dz), p.numel() // C, C, CP, _dt(dz), _s())
        return dz, None


def head_softmax(z, C):
    """class softmax over the first C of the (padded) channels of z -> fp32 probabilities [..., C]."""
    return _HeadSoftmax.apply(z, C)


class _FinalSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, C):
        lc = ctx.lc = current()
        _chk(z, 'z')
        B, h, w, D, CP = z.shape
        p = torch.empty((B, 2 * h, 2 * w, D, C), device=z.device, dtype=torch.float32)
        _lib.call('ltu_final_softmax_fwd', _p(z), _p(p), B, h, w, D, C, CP, _dt(z), _s())
        ctx.save_for_backward(p)
        ctx.cfg = (B, h, w, D, C, CP, z.dtype)
        return p

    @staticmethod
    def backward(ctx, g):
        lc = ctx.lc
        (p,) = ctx.saved_tensors
        B, h, w, D, C, CP, zdt = ctx.cfg
        g = g.contiguous()
        dz = torch.empty((B, h, w, D, CP), device=p.device, dtype=zdt)
        _lib.call('ltu_final_softmax_bwd', _p(g), _p(p), _p(dz), B, h, w, D, C, CP, _dt(dz), _s())
        return dz, None


def final_softmax(z, C):
    """window un-embedding + class softmax: z [B,h,w,D,CP >= 4C] (conv output, possibly padded) -> fp32 probabilities [B,2h,2w,D,C]."""
    return _FinalSoftmax.apply(z, C)


class _Gate(torch.autograd.Function):
    """skip * sigmoid(psi . relu(IN(Wx skip) + IN(Wg up)) + b)   (model/Unet_3Dblock.py:217-221, 1385)."""

    @staticmethod
    def forward(ctx, skip, up, wx, bx, wg, bg, pw, pb, px, pg):
        lc = ctx.lc = current()
        _chk(skip, 'skip'); _chk(up, 'up')
        B, C = skip.shape[0], skip.shape[-1]
        Cg = up.shape[-1]
        M = skip.numel() // C
        S = M // B
        dev, dt = skip.device, _dt(skip)
        u1 = torch.empty((M, C), device=dev, dtype=skip.dtype)
        u2 = torch.empty((M, C), device=dev, dtype=skip.dtype)
        wxo = px.w[0] if px is not None else _w_operand(wx, skip.dtype)
        wgo = pg.w[0] if pg is not None else _w_operand(wg, skip.dtype)
        _lib.call('ltu_linear_fwd', _p(skip), C, _ptr_array([wxo]), 1, _ptr_array([bx]), _p(u1), C, M, C, C, 0, dt, _s())
        _lib.call('ltu_linear_fwd', _p(up), Cg, _ptr_array([wgo]), 1, _ptr_array([bg]), _p(u2), C, M, C, Cg, 0, dt, _s())
        s1 = lc.scratch_zeros((B, C, 3), dev)
        s2 = lc.scratch_zeros((B, C, 3), dev)
        _lib.call('ltu_instnorm_stats', _p(u1), _p(s1), _p(lc.norm_ws(dev)), _n(lc.norm_ws(dev)), B, S, C, dt, _s())
        _lib.call('ltu_instnorm_stats', _p(u2), _p(s2), _p(lc.norm_ws(dev)), _n(lc.norm_ws(dev)), B, S, C, dt, _s())
        a = torch.empty(M, device=dev, dtype=torch.float32)
        out = torch.empty_like(skip)
        _lib.call('ltu_gate_fwd', _p(u1), _p(u2), _p(s1), _p(s2), _p(pw), _p(pb), _p(skip), _p(a), _p(out), B, S, C, dt, _s())
        ctx.save_for_backward(skip, up, u1, u2, s1, s2, a)
        ctx.params = (wx, bx, wg, bg, pw, pb)
        ctx.prep = (px, pg)
        return out

    @staticmethod
    def backward(ctx, g):
        lc = ctx.lc
        skip, up, u1, u2, s1, s2, a = ctx.saved_tensors
        wx, bx, wg, bg, pw, pb = ctx.params
        px, pg = ctx.prep
        g = g.contiguous()
        B, C = skip.shape[0], skip.shape[-1]
        Cg = up.shape[-1]
        M = skip.numel() // C
        S = M // B
        dev, dt = skip.device, _dt(skip)
        dskip = torch.empty_like(skip)
        ds = torch.empty(M, device=dev, dtype=torch.float32)
        dpw, fpw = _grad_buf(pw)
        dpb, fpb = _grad_buf(pb)
        bs1 = lc.scratch_zeros((B, C, 2), dev)
        bs2 = lc.scratch_zeros((B, C, 2), dev)
        du1 = torch.empty_like(u1)
        du2 = torch.empty_like(u2)
        _lib.call('ltu_gate_bwd', _p(g), _p(u1), _p(u2), _p(s1), _p(s2), _p(pw), _p(skip), _p(a), _p(dskip), _p(ds), _p(dpw),
                  _p(dpb), _p(bs1), _p(bs2), _p(lc.norm_ws(dev)), _n(lc.norm_ws(dev)), _p(du1), _p(du2), B, S, C, dt, _s())
        # through the two 1x1x1 convs
        wxt = px.wt if px is not None else _w_transposed([wx], C, C, skip.dtype)
        wgt = pg.wt if pg is not None else _w_transposed([wg], C, Cg, skip.dtype)
        dup = torch.empty_like(up)
        # dskip += du1 . Wx  (accumulating epilogue), dup = du2 . Wg
        _lib.call('ltu_linear_fwd', _p(du1), C, _ptr_array([wxt]), 1, _ptr_array([None]), _p(dskip), C, M, C, C, 1, dt, _s())
        _lib.call('ltu_linear_fwd', _p(du2), C, _ptr_array([wgt]), 1, _ptr_array([None]), _p(dup), Cg, M, Cg, C, 0, dt, _s())
        dwx, f1 = _grad_buf(wx)
        dbx, f2 = _grad_buf(bx)
        dwg, f3 = _grad_buf(wg)
        dbg, f4 = _grad_buf(bg)
        def launch(keep):
            w1, w2 = _wgrad_ws(M, C, C, skip), _wgrad_ws(M, C, Cg, skip)
            keep.extend((w1, w2))
            _lib.call('ltu_linear_wgrad', _p(du1), C, _p(skip), C, _ptr_array([dwx]), _ptr_array([dbx]), 1, M, C, C, _p(w1), _n(w1), 0, dt, _s())
            _lib.call('ltu_linear_wgrad', _p(du2), C, _p(up), Cg, _ptr_array([dwg]), _ptr_array([dbg]), 1, M, C, Cg, _p(w2), _n(w2), 0, dt, _s())
        if f1 and f2 and f3 and f4:
            lc.wq_push(launch, (du1, du2, skip, up), 'gate')
        else:
            launch([])
        return (dskip, dup, _grad_done(wx, dwx, f1), _grad_done(bx, dbx, f2), _grad_done(wg, dwg, f3), _grad_done(bg, dbg, f4),
                _grad_done(pw, dpw, fpw), _grad_done(pb, dpb, fpb), None, None)


def attention_gate(skip, up, wx, bx, wg, bg, pw, pb, prep_x=None, prep_g=None):
    return _Gate.apply(skip, up, wx, bx, wg, bg, pw, pb, prep_x, prep_g)


# ---------------------------------------------------------------------------------------------- loss

class _LevelLoss(torch.autograd.Function):
    """Weighted sum of the losses of one decoder level; returns (total, values[1+2+C]) with values detached."""

    @staticmethod
    def forward(ctx, p, label, w_ce, w_bal, w_dice, scale_dev):
        lc = ctx.lc = current()
        _chk(p, 'p'); _chk(label, 'label')
        B, C = p.shape[0], p.shape[-1]
        S = p.numel() // (B * C)
        dev = p.device
        sums = torch.empty(_lib.load().ltu_loss_ws_floats(B, S, C), device=dev, dtype=torch.float32)      # partials + sums: no zero fill
        buf = torch.empty(9, device=dev, dtype=torch.float32)
        values = buf[:8]                 # the report (non-differentiable); buf[8] repeats the total as the differentiable output, so
        coef = torch.empty((B, C, 3), device=dev, dtype=torch.float32)          # no copy kernel is needed to separate the two
        wd = (ctypes.c_float * 5)(*[float(w_dice[c]) if c < len(w_dice) else 0.0 for c in range(5)])
        _lib.call('ltu_loss_fwd', _p(p), _p(label), _p(sums), _n(sums), _p(values), _p(coef), B, S, C, float(w_ce), float(w_bal), wd, _p(scale_dev), _s())
        ctx.save_for_backward(p, label, coef)
        ctx.mark_non_differentiable(values)
        ctx.set_materialize_grads(False)         # no zero-filled gradient tensor for the report output
        return buf[8], values

    @staticmethod
    def backward(ctx, g, _gv):
        lc = ctx.lc
        p, label, coef = ctx.saved_tensors
        B, C = p.shape[0], p.shape[-1]
        S = p.numel() // (B * C)
        if g is None:
            return None, None, None, None, None, None
        g = g.contiguous().to(torch.float32)
        dp = torch.empty_like(p)
        _lib.call('ltu_loss_bwd', _p(p), _p(label), _p(coef), _p(g), _p(dp), B, S, C, _s())
        return dp, None, None, None, None, None


def level_loss(p, label, w_ce=0.0, w_bal=0.0, w_dice=(), scale_dev=None):
    """p fp32 [B,...,C] channels-last probabilities, label uint8 [B,...]: weighted CE + balanced Dice + per-class Dice
    (w_dice[c], c < 4) + Dice of the foreground union (w_dice[4]).
    scale_dev: optional 1-element fp32 device tensor multiplying all weights at run time."""
    return _LevelLoss.apply(p, label, w_ce, w_bal, tuple(w_dice), scale_dev)

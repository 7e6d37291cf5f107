#!/usr/bin/env python3
"""Benchmark of the LinTransUNet hot path on MI355X: 128^3 CT patches/s, forward + 5-level loss + backward.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One process per GPU, weak scaling (2 patches per GPU, BASELINE.json config 3): every rank draws its own
synthetic patches, gradients are all-reduced (mean) over RCCL in buckets overlapped with backward (direct RCCL calls behind the
C-ABI, issued between the linear graph segments the step is replayed as; the host control plane - rendezvous, barriers, max over
ranks - is gloo).  A "step" is one inner training step of
utils/utils_3D_embed_full.py:55-86 (dropout 0.3 as in training, no optimizer step, inputs resident in HBM).
Rank 0 prints ONE JSON line.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment launches itself: the parent starts N
child processes (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1) BEFORE
anything touches the GPU, relays their output and exits non-zero if any child fails.  `--dry-run` exercises exactly
that launcher + rendezvous + JSON plumbing on the CPU (gloo), without the model.
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Roofline object: the chain kernels that carry the post-attention half of every transformer layer (out projection, LayerNorm,
# FFN, LayerNorm: model/trans_block.py:203-211) are the largest kernel family by time.  Their GEMMs have AI ~ 98 flop/B < ridge
# ~ 310: HBM-bound.  Algorithmic bytes of a layer with M tokens: forward 10 M d, backward 13 M d
# elements of activations (each tensor read or written once) + 16 d^2 of weights, bf16.
HBM_PEAK_GBS = 8000.0


# Whole-step mixed roofline of SURVEY.md section 8d: T_roof(fwd+bwd) per 128^3 patch = 3 * (394.1 GFLOP of 3x3x3 convs / 2.5 PFLOP/s
# + 5.11 GB of everything else / 8 TB/s) = 2.39 ms; other patch sizes scale with the measured per-size work of the same table.
ROOF_MS_PER_PATCH = {128: 3 * (394.1e9 / 2.5e15 + (5.797e9 - 0.686e9) / 8e12) * 1e3,
                     96: 3 * (254.4e9 / 2.5e15 + (4.023e9 - 0.396e9) / 8e12) * 1e3}


def synthetic_batch(batch, size, seed, device, n_classes=2):
    """N(0,1) volume clipped to the dataset's normalised HU range + ellipsoid labels (3 label
    values for the multi-class configuration 4)."""
    from lintransunet_amd import data
    return data.synthetic_patches(batch, size, seed, device, n_classes=n_classes)


def cpu_baseline(size=(128, 128, 128), threads=None, timed=2):
    """The oracle (CPU restatement of the reference, fp32, dropout on): 1 warm-up + `timed` timed steps at B=1, mean."""
    import torch
    from oracle import net as O_net, seedgen, step as O_step
    if threads is None:        # the box exposes every host core but grants a 16-core share per GPU
        try:
            threads = len(os.sched_getaffinity(0))
        except AttributeError:
            threads = os.cpu_count() or 1
        threads = max(1, min(threads, 16))
    torch.set_num_threads(threads)
    print(f'[bench] cpu_baseline: oracle, 1 warm-up + {timed} timed steps on {threads} host threads ...', file=sys.stderr, flush=True)
    cfg = O_net.NetConfig(dropout=0.3)
    P = seedgen.seeded_params(O_net.param_shapes(cfg), 7, requires_grad=True)
    x = seedgen.seeded_volume((1, 1) + size, 8)
    lab = seedgen.seeded_label((1, 1) + size, 9)
    times = []
    for i in range(1 + timed):
        for p in P.values():
            p.grad = None
        t0 = time.perf_counter()
        O_step.train_step(P, cfg, x, lab, O_step.dynamic_weights(0))
        times.append(time.perf_counter() - t0)
        print(f'[bench] cpu_baseline step {i}: {times[-1]:.1f} s', file=sys.stderr, flush=True)
    dt = sum(times[1:]) / timed
    return {'value': 1.0 / dt, 'unit': 'patches/s', 'cores': threads, 'kind': 'port',
            'sample': f'{timed} fwd+bwd steps after 1 warm-up, {size[0]}^3 B=1, fp32, dropout 0.3 (warm-up {times[0]:.1f} s, timed '
                      + ', '.join(f'{t:.1f}' for t in times[1:]) + ' s)'}


class KernelTimer:
    """Timing of one op family with HIP events on the stream it is launched on (torch's current stream).

    Eagerly launched kernels cannot be bracketed tightly (the host needs ~35 us per launch, more than the kernel runs), so the
    calls of one eager step are recorded (arguments kept alive) and then replayed back to back from a captured HIP graph of
    exactly those launches, bracketed by one event pair: sum of the kernels' durations plus the ~1 us node-to-node gaps."""

    def __init__(self):
        self.calls = []
        self.work = 0.0         # accumulated algorithmic bytes of the recorded launches
        self.on = False

    def wrap(self, fn, work_of, tag=None):
        def wrapped(*a, **k):
            if self.on:
                self.calls.append((tag,) + tuple(a[1:]))          # drop the autograd ctx
                self.work += work_of(*a, **k)
            return fn(*a, **k)
        return wrapped

    def measure(self, replay_fn, reps=5):
        """replay_fn(args) re-issues one recorded launch; returns (total ms per pass over all recorded launches, launches)"""
        import torch
        if not self.calls:
            return 0.0, 0
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.no_grad():
            for c in self.calls:
                replay_fn(c)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode='thread_local'), torch.no_grad():
            for c in self.calls:
                replay_fn(c)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps, len(self.calls)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--size', type=int, default=128)
    ap.add_argument('--batch', type=int, default=2, help='patches per GPU')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--classes', type=int, default=2, choices=[2, 3],
                    help='model outputs: 2 = single-class pancreas (BASELINE configs 2/3), 3 = multi-class path (config 4)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-families', action='store_true', help='skip the per-family timing table of the roofline object')
    ap.add_argument('--no-graph', action='store_true', help='launch every kernel eagerly instead of replaying a captured HIP graph')
    ap.add_argument('--allreduce', default=None, choices=['segments', 'graph', 'after'],
                    help="gradient all-reduce of the graph-replayed step: between linear graph segments (overlapped, default), captured as "
                         "side branches of one graph, or after the replay")
    ap.add_argument('--rehearse-comm', action='store_true',
                    help='1 GPU only: run the step with a live 1-rank RCCL communicator and the collective path forced on (what every '
                         'rank does at N > 1 minus the link time): prices the captured fork / join edges and RCCL launches')
    ap.add_argument('--test-comm', default=None, choices=['staged'],
                    help='rehearsal only: several ranks on ONE GPU with the gradient exchange staged through the host and gloo (RCCL refuses '
                         'two ranks per device); exercises launcher, rendezvous, broadcast, hooks, segments and the JSON line - not a measurement')
    ap.add_argument('--bucket-mb', type=float, default=32.0)
    ap.add_argument('--tail-mb', type=float, default=0.5)
    ap.add_argument('--dry-run', action='store_true',
                    help='CPU/gloo rehearsal of the launcher, rendezvous, barrier / max-over-ranks timing and the JSON line; no GPU, no model')
    return ap.parse_args(argv)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children.  The parent never initialises the GPU (it
    imports nothing that does) and never replaces itself; rank 0's JSON line reaches stdout through the inherited descriptor."""
    import subprocess
    port = os.environ.get('MASTER_PORT') or str(_free_port())
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=port)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = set(range(len(procs)))
        while pending:
            for i in list(pending):
                code = procs[i].poll()
                if code is None:
                    continue
                pending.discard(i)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print(f'[bench] rank {i} exited with {code}; stopping the other ranks', file=sys.stderr)
                    for k in pending:
                        procs[k].terminate()
            time.sleep(0.05)
    except KeyboardInterrupt:
        rc = 130
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def largest_kernel(group, pmc_json, stats_csv):
    """second `roofline` object: wgrad_group_ring_bf16_kernel (+ its fold).  group = (ms alone per step, launches, operand bytes) from
    family_timer.FamilyTimer.group_wgrad (live, one HIP event pair around a graph of exactly these launches at the side-stream width);
    in-step time from the committed kernel statistics, HBM traffic from the committed PMC passes"""
    if group is None:
        return None
    ms, n, byts = group
    out = {'kernel': 'wgrad_group_ring_bf16_kernel + wgroup_fold_kernel (grouped projection weight gradients, side stream, 128 workgroups)',
           'bound': 'hbm', 'launches': n, 'algorithmic_bytes_per_step': byts, 'ms_alone': ms,
           'achieved': byts / (ms * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           'ms_in_step': None, 'frac_in_step': None, 'traffic': None}
    if stats_csv:
        import csv
        t = 0.0
        for r in csv.DictReader(open(stats_csv)):
            if 'wgrad_group_ring' in r['Name'] or 'wgroup_fold' in r['Name']:
                t += int(r['TotalDurationNs']) / 8 / 1e6
        if t > 0:
            out.update(ms_in_step=t, frac_in_step=byts / (t * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       in_step_source=f'profiles/{os.path.basename(stats_csv)}')
    if pmc_json:
        fam = json.load(open(pmc_json))['families'].get('projection weight gradients (grouped ring kernel + fold)')
        if fam:
            out.update(traffic=(fam['hbm_read_GB_per_step'] + fam['hbm_write_GB_per_step']) * 1e9,
                       traffic_source=f'profiles/{os.path.basename(pmc_json)} (bytes per step)')
    return out


def quiet_stdout():
    """gloo and RCCL print connection / version banners to stdout (file descriptor 1) from C++; the benchmark's stdout carries exactly
    ONE JSON line, so they are sent to stderr while the process group and the communicator come up"""
    from lintransunet_amd.comm import _stdout_to_stderr
    return _stdout_to_stderr()


def dry_run(args, rank, world):
    """the multi-process plumbing of main() on the CPU: gloo rendezvous, barrier-bracketed timed region, MAX over ranks, one line"""
    import torch
    import torch.distributed as dist
    if os.environ.get('LTU_BENCH_FAIL_RANK') == str(rank):       # test hook: a rank that dies before the rendezvous
        raise RuntimeError('simulated rank failure')
    if world > 1:
        with quiet_stdout():
            dist.init_process_group('gloo')
            dist.barrier()
    buf = torch.zeros(1 << 16)
    for _ in range(args.warmup):
        if world > 1:
            dist.all_reduce(buf)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        buf.add_(1.0)
        if world > 1:
            dist.all_reduce(buf)
    if world > 1:
        dist.barrier()
    tmax = torch.tensor([time.perf_counter() - t0])
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({'metric': f'{args.size}^3 CT patches/sec (fwd+bwd)', 'value': None, 'unit': 'patches/s', 'n_gpus': world,
                          'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': tmax.item() / args.steps * 1e3,
                          'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'none',
                          'config': {'workload': 'dry run: launcher + gloo rendezvous only, no GPU work', 'parallelism': f'dp{world}'}}))
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args))         # before torch / HIP are touched in this process

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        print(f'[bench] --gpus {args.gpus} but the launcher set WORLD_SIZE={world}: using {world} ranks', file=sys.stderr)
    if args.dry_run:
        return dry_run(args, rank, world)
    if args.test_comm:
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    from lintransunet_amd.model import get_model_dict
    from lintransunet_amd import train, ops
    from lintransunet_amd import comm as C

    comm = None
    if world > 1:
        # host control plane (unique id, barriers, max over ranks): gloo on 127.0.0.1; gradient exchange: direct RCCL calls behind
        # the C-ABI (lintransunet_amd/comm.py) - no ProcessGroupNCCL, hence no watchdog thread next to the graph captures
        with quiet_stdout():
            dist.init_process_group('gloo')
            dist.barrier()                  # gloo connects lazily: its banner comes with the first collective
            comm = C.HostStagedComm(C.GlooComm()) if args.test_comm else C.RcclComm(dev, control=C.GlooComm())
    elif args.rehearse_comm:
        comm = C.RcclComm(dev)

    torch.manual_seed(1234)          # same initial weights on every rank (then broadcast for good measure)
    act = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    model = get_model_dict('MaskTransUnet')([16, 32, 64, 128, 256], [100, 65, 40, 25, 10], [False, True, True, True, True],
                                            1, args.classes, dropout=0.3, act_dtype=act).to(dev).train()
    train.broadcast_parameters(model, comm)
    torch.manual_seed(1234 + rank)   # dropout streams differ per rank
    reducer = train.GradReducer(model, bucket_mb=args.bucket_mb, unused=train.UNUSED_PARAMETERS, comm=comm, tail_mb=args.tail_mb,
                                force_collectives=args.rehearse_comm and world == 1)
    size = (args.size,) * 3
    weights = train.get_dynamic_weight(1)[0]
    batches = [synthetic_batch(args.batch, size, 100 + 10 * rank + i, dev, args.classes) for i in range(2)]
    # multi-class script (train3D_multi_class.py / utils_3D_multi_class.py:85-102): CE + Dice(class 1) + Dice(class 2), weights 10/1/2
    specs = train.level_specs(5, ('CrossEntroLoss', 'DiceClassLoss', 'DiceClassLoss2'), criterion_weight=[10, 1, 2]) if args.classes == 3 else None

    # live timing of the dominant kernel family: the transformer layer chain kernels (csrc/tlayer.hip: tail_fwd_kernel and
    # tail_bwd_kernel carry the post-attention half of every transformer layer, forward and backward; together the largest family by
    # time).  Their (tokens, d, second-gradient) shapes are recorded over one eager step; the same launches are then re-issued on
    # buffers of those shapes from a small captured graph and bracketed by ONE HIP event pair on the launching stream.
    timer = KernelTimer()
    import numpy as np
    from lintransunet_amd import _lib
    from lintransunet_amd.ops import _p, _s

    def tail_bytes(ctx, a, x, *rest):
        # forward: reads q (the q third of the qkv rows: the attention's phase B runs inside the kernel), x; writes a, z1, t1, z2, y
        # (d wide) and u, h (2d wide); backward: reads dy, dy2, z2, z1, u; writes dr2, dr1, dz1, da, du; weights (8 d^2 forward +
        # 8 d^2 backward, bf16) are read once per launch from HBM.  A launch that also forms the next layer's q|k|v projection
        # (round 3) writes 3 M d more and reads 3 d^2 more weights
        M, d = x.shape
        nxt = len(rest) > 16 and rest[-1] is not None
        return float((11 + 13 + (3 if nxt else 0)) * M * d + (16 + (3 if nxt else 0)) * d * d) * 2.0
    ops._LayerTail.forward = staticmethod(timer.wrap(ops._LayerTail.forward, tail_bytes, 'tail'))
    chain_cache = {}

    def chain_buffers(M, d):
        if (M, d) not in chain_cache:
            bf = lambda *sh: (torch.randn(*sh, device=dev) * 0.5).bfloat16()
            def fragw(n, k, kind):
                w = torch.randn(n, k, device=dev) / k ** 0.5
                out = torch.empty(n * k, device=dev, dtype=torch.bfloat16)
                rec = np.zeros(1, dtype=ops.WPREP_DTYPE)
                rec[0] = (w.data_ptr(), out.data_ptr(), kind, n, k, 0, 0, 0)
                table = torch.from_numpy(rec.view(np.uint8).copy()).to(dev)
                _lib.call('ltu_weight_prep', table.data_ptr(), 1, 1, _s())
                torch.cuda.synchronize()
                return out
            nblk = _lib.load().ltu_layer_tail_blocks(M)
            chain_cache[(M, d)] = dict(
                a=bf(M, d), x=bf(M, d), dy=bf(M, d), dy2=bf(M, d), w=[fragw(d, d, 8), fragw(2 * d, d, 8), fragw(d, 2 * d, 8)],
                qkv=bf(M, 3 * d), ctx=torch.randn(args.batch * (d // 32), 32, 32, device=dev) * 0.05, qstat=torch.empty(M, d // 32, 2, device=dev),
                wq=torch.cat([fragw(d, d, 8) for _ in range(3)]), qkv_next=torch.empty(M, 3 * d, device=dev, dtype=torch.bfloat16),
                wt=[fragw(d, 2 * d, 9), fragw(2 * d, d, 9), fragw(d, d, 9)], bias=torch.zeros(2 * d, device=dev),
                gamma=torch.ones(d, device=dev), md=[torch.empty(M, d, device=dev, dtype=torch.bfloat16) for _ in range(8)],
                m2d=[torch.empty(M, 2 * d, device=dev, dtype=torch.bfloat16) for _ in range(3)],
                stat=[torch.empty(M, 2, device=dev) for _ in range(2)], lnws=torch.empty(2, nblk, 2 * d, device=dev))
        return chain_cache[(M, d)]

    def replay(c):
        M, d = c[2].shape                  # x [M, d] (c[1] is a [M, d] or, with the fused attention phase B, qkv [M, 3d])
        fused = c[1].shape[1] == 3 * d
        nxt = len(c) > 19 and c[-1] is not None        # the launch also forms the next layer's q|k|v projection
        q = chain_buffers(M, d)
        z1, t1, z2, y, dr2, dr1, dz1, da = q['md']
        u, h, du = q['m2d']
        _lib.call('ltu_layer_tail_fwd', _p(q['a']), _p(q['x']), _p(q['w'][0]), _p(q['w'][1]), _p(q['w'][2]), _p(q['bias']), _p(q['bias']),
                  _p(q['bias']), _p(q['gamma']), _p(q['bias']), _p(q['gamma']), _p(q['bias']), _p(z1), _p(t1), _p(u), _p(h), _p(z2), _p(y),
                  _p(q['stat'][0]), _p(q['stat'][1]), M, d, 1e-6, 0.3, 11, 12, 13, 0, 1, _p(q['qkv']) if fused else 0,
                  _p(q['ctx']) if fused else 0, _p(q['qstat']) if fused else 0, M // args.batch if fused else 0,
                  _p(q['wq']) if nxt else 0, _p(q['bias']) if nxt else 0, _p(q['bias']) if nxt else 0, _p(q['bias']) if nxt else 0,
                  _p(q['qkv_next']) if nxt else 0, 1, _s())
        _lib.call('ltu_layer_tail_bwd', _p(q['dy']), _p(q['dy2']), _p(z2), _p(z1), _p(u), _p(q['stat'][1]), _p(q['stat'][0]), _p(q['gamma']),
                  _p(q['gamma']), _p(q['wt'][0]), _p(q['wt'][1]), _p(q['wt'][2]), _p(dr2), _p(du), _p(dr1), _p(dz1), _p(da), _p(q['lnws'][0]),
                  _p(q['lnws'][1]), q['lnws'][0].numel(), M, d, 0.3, 11, 12, 13, 0, 1, 1, _s())

    def eager_step(i):
        reducer.zero_grad()
        x, lab = batches[i % 2]
        return train.train_step(model, x, lab, weights, specs=specs, reducer=reducer)

    # The dominant kernel family is recorded over one eager step and re-timed below from a graph of exactly those launches (forward
    # and backward chain of every transformer layer); the timed region of the headline number replays the whole step from a captured
    # HIP graph.
    eager_step(0)
    torch.cuda.synchronize()
    timer.on = True
    eager_step(1)
    torch.cuda.synchronize()
    timer.on = False
    reducer.rebucket()                        # buckets in gradient-ready order (recorded by the eager step): each bucket's all-reduce
                                              # starts while backward still produces the next one
    # per-family device time of the step (lintransunet_amd/family_timer.py): the C-ABI calls that go into the captured step graph are
    # logged; AFTER the timed region each family's launches are replayed alone from a graph of their own between one HIP event pair
    families, ft, group = None, None, None
    if not args.no_families and not args.no_graph:
        from lintransunet_amd.family_timer import FamilyTimer
        ft = FamilyTimer()
    for c in timer.calls:
        chain_buffers(*c[1].shape)            # allocate / prepare outside the timed replay
    ms_lin, n_lin = timer.measure(replay)
    n_lin *= 2                                # one forward and one backward launch per recorded layer
    timer.calls = []

    launch, graph_mode, graphed = 'eager', None, None
    allreduce = 'hooks (overlapped with backward)' if (world > 1 or args.rehearse_comm) else 'none (1 rank)'
    step = eager_step
    if not args.no_graph:
        # ladder: step as linear graph segments with the collectives between them (overlapped) -> collectives captured as side branches of
        # one graph -> collectives after the replay -> eager launches.
        # A failed capture must not cost the measurement; every rung runs the same kernels.
        multi = world > 1 or args.rehearse_comm
        modes = [args.allreduce] if args.allreduce else (['segments', 'graph', 'after'] if multi else ['segments', 'graph'])
        for mode in modes:
            try:
                if ft is not None:
                    ft.calls.clear()
                    ft.attach()
                try:
                    graphed = train.GraphedStep(model, batches[0][0], batches[0][1], weights, reducer, specs=specs, overlap=mode)
                finally:
                    if ft is not None:
                        ft.detach()
                step = lambda i: graphed(*batches[i % 2])
                launch = 'hip-graph replay'
                if multi:
                    allreduce = {'segments': 'eager RCCL calls between linear graph segments cut where a bucket closes (overlapped with backward)',
                                 'graph': 'captured in the step graph (side branches, overlapped with backward)',
                                 'after': 'after the replay (exposed)'}[mode]
                break
            except Exception as e:
                print(f'[bench] rank {rank}: HIP-graph capture with all-reduce mode {mode!r} failed ({type(e).__name__}: {e})',
                      file=sys.stderr)
                torch.cuda.synchronize()
                # At N > 1 a silent step down the ladder would read as a scaling defect (the collectives of 'after' are fully
                # exposed): the first multi-GPU run fails loudly instead, unless the caller asked for a mode or allows the fallback
                if world > 1 and not args.allreduce and os.environ.get('LTU_ALLOW_FALLBACK', '0') != '1':
                    print(f'[bench] rank {rank}: not falling back at world {world} (LTU_ALLOW_FALLBACK=1 allows it)', file=sys.stderr)
                    raise
        graph_mode = mode if launch == 'hip-graph replay' else None

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    diag = {}
    if world > 1:
        # self-diagnosis of the multi-rank step (the first N > 1 RCCL execution happens on the driver's node): after the warm-up
        # steps every rank must hold the SAME averaged gradients - per-bucket checksums, MAX - MIN over ranks through the gloo
        # control plane - or the run fails non-zero
        sums = torch.stack([torch.stack([f.double().sum(), f.double().abs().sum()]) for f in reducer.flat]).cpu()
        hi, lo = sums.clone(), sums.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        spread = ((hi - lo).abs() / hi.abs().clamp_min(1e-30)).max().item()
        diag['bucket_checksum_spread_over_ranks'] = spread
        # NaN / all-zero buckets / a gross mismatch (> 1e-6 relative) mean the exchange is broken: fail.  A last-bit spread (the ranks
        # of a ring all-reduce may add in different orders) is reported and the run goes on.
        if not (spread <= 1e-6) or not bool(torch.isfinite(sums).all()) or float(sums[:, 1].min()) == 0.0:
            print(f'[bench] rank {rank}: gradient buckets differ between ranks after the all-reduce (relative spread {spread:.3e}, '
                  f'checksums {sums.tolist()}): the exchange is broken', file=sys.stderr)
            sys.exit(3)
        if spread != 0.0 and rank == 0:
            print(f'[bench] per-bucket checksums differ between ranks in the last bits (relative spread {spread:.3e}): continuing', file=sys.stderr)
    if world > 1:
        comm.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        comm.barrier()
    dt = time.perf_counter() - t0
    timer.on = False
    if world > 1:
        dt = comm.max_float(dt)
    if (world > 1 or args.rehearse_comm or os.environ.get('LTU_BENCH_LOCAL') == '1') and launch == 'hip-graph replay':
        # what the exchange costs: the same replayed step without its collectives, and where in the step each bucket closes
        for i in range(2):
            graphed.replay_local(*batches[i % 2])
        torch.cuda.synchronize()
        if world > 1:
            comm.barrier()
        t1 = time.perf_counter()
        for i in range(args.steps):
            graphed.replay_local(*batches[i % 2])
        torch.cuda.synchronize()
        if world > 1:
            comm.barrier()
        dt_local = time.perf_counter() - t1
        if world > 1:
            dt_local = comm.max_float(dt_local)
        marks, total = graphed.bucket_timeline()
        diag['ms_per_step_without_collectives'] = dt_local / args.steps * 1e3
        diag['exposed_collective_ms'] = (dt - dt_local) / args.steps * 1e3
        diag['buckets'] = [{'bucket': bi, 'MB': round(mb, 2), 'closes_at_ms': round(t, 3)} for bi, mb, t in marks]
        diag['compute_ms_of_that_replay'] = round(total, 3)
    if ft is not None and ft.calls and launch == 'hip-graph replay':
        # replays the recorded C-ABI calls (widths and workspace capacities are among their arguments) on stale buffers: after the
        # timed region
        stats = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_bench_kernel_stats.csv')))
        from lintransunet_amd.family_timer import in_step_ms
        in_step = in_step_ms(stats[-1]) if stats and args.size == 128 and args.batch == 2 and args.classes == 2 else None
        families = ft.table(ft.measure(), args.size, args.batch, in_step=in_step)
        if in_step is not None:
            for row in families:
                row['in_step_source'] = f'profiles/{os.path.basename(stats[-1])} (rocprofv3 kernel trace of the same command, committed)'
        group = ft.group_wgrad()
        reducer.zero_grad()                                            # ... and the weight-gradient kernels among them accumulate
        torch.cuda.synchronize()

    if rank == 0:
        patches = args.batch * world * args.steps
        achieved = timer.work / (ms_lin * 1e-3) / 1e9 if ms_lin > 0 else 0.0       # GB/s
        # HBM traffic of the same kernel family from the PMC counters (FETCH_SIZE x2 on gfx950 + WRITE_SIZE): counters need their
        # own rocprofv3 --pmc passes, so this is the committed offline collection of tools/pmc_step.sh, not a live number
        traffic, traffic_src = None, None
        pmcs = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_step.json')))
        pmc = pmcs[-1] if pmcs else ''
        if pmc:
            fam = json.load(open(pmc))['families'].get('transformer layer chain kernels (forward + backward)')
            if fam and fam['dispatches_per_step']:
                traffic = (fam['hbm_read_GB_per_step'] + fam['hbm_write_GB_per_step']) * 1e9 / fam['dispatches_per_step']
                traffic_src = f'profiles/{os.path.basename(pmc)} (offline rocprofv3 --pmc passes of the same step)'
        ms_step = dt / args.steps * 1e3
        roof_ms = ROOF_MS_PER_PATCH.get(args.size)
        stats_all = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_bench_kernel_stats.csv')))
        stats_csv = stats_all[-1] if stats_all and args.size == 128 and args.batch == 2 and args.classes == 2 else None
        out = {
            'metric': f'{args.size}^3 CT patches/sec (fwd+bwd)', 'value': patches / dt, 'unit': 'patches/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': f'MaskTransUnet train step (fwd + 5-level loss + bwd), {args.size}^3 single-channel patches, '
                                   f'{args.batch} per GPU, dropout 0.3, random-init weights' + (', 3 labels (multi-class losses)' if args.classes == 3 else ''), 'global_batch': args.batch * world,
                       'patch': [args.size] * 3, 'parallelism': f'dp{world}',
                       'launch': launch + (f' ({graph_mode}: linear graph segments, weight gradients as graphs of their own on a side stream)'
                                           if graph_mode == 'segments' and graphed.wq_stream is not None else
                                           f' ({graph_mode})' if graph_mode else ''),
                       # whether the probed side stream really runs beside the compute (and communicator) stream: False = it shares
                       # a hardware queue and the weight gradients serialise (ops.concurrent_stream warns on stderr)
                       **({'side_stream_concurrent': bool(getattr(graphed.wq_stream, 'ltu_concurrent', False))}
                          if graphed is not None and graphed.wq_stream is not None else {}),
                       'peak_memory_gb': round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2),
                       **({'multi_rank_diagnosis': diag} if diag else {}),
                       'allreduce': allreduce + (' [REHEARSAL: host-staged test communicator, ranks share one GPU - not a measurement]' if args.test_comm else '')},
            'roofline': {'bound': 'hbm', 'kernel': 'tail_fwd_kernel + tail_bwd_kernel (row-block chain kernels: post-attention half of every transformer layer - with the next layer\'s q|k|v projection where the layers are adjacent - forward and backward)',
                         'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'launches': n_lin, 'avg_launch_ms': ms_lin / max(n_lin, 1),
                         'algorithmic_bytes_per_launch': timer.work / max(n_lin, 1), 'traffic': traffic,
                         'traffic_source': traffic_src,
                         'timing': 'one HIP event pair around a graph replay of exactly these launches',
                         # the whole step against SURVEY 8d's mixed roofline (convs at the MFMA peak + everything else at the HBM peak)
                         'step_roofline_ms': roof_ms * args.batch if roof_ms else None,
                         'step_frac': (roof_ms * args.batch / ms_step) if roof_ms else None,
                         # every op family of the step, replayed alone (graph of exactly its launches, one HIP event pair), against the
                         # algorithmic work SURVEY 8d assigns it (fwd + bwd = 3 x forward) and the peak that bounds it
                         'families': families,
                         # the largest SINGLE kernel of the step (11-12 % of all kernel time), on the side stream: the grouped
                         # projection weight gradient; algorithmic bytes = its operands read once ((M K + M N) 2 B per job)
                         'largest_kernel': largest_kernel(group, pmcs[-1] if pmcs else None, stats_csv)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(size)
        print(json.dumps(out))
    if world > 1:
        comm.close()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

"""CPU tests of the host side: C-ABI surface, state_dict surface, loud failure without a GPU, and the
world-size-2 gradient reducer over gloo."""
import os
import re
import socket

import pytest
import torch
import torch.multiprocessing as mp

from oracle import net as O_net

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_exports_every_declared_symbol():
    from lintransunet_amd import _lib
    header = open(os.path.join(ROOT, 'include', 'ltu_hip.h')).read()
    # what an experiments build adds (#ifdef LTU_EXPERIMENTS ... #endif) is not part of the product surface
    exp = ''.join(re.findall(r'#ifdef LTU_EXPERIMENTS.*?#endif', header, flags=re.S))
    assert set(re.findall(r'^(?:int|long long)\s+(ltu_\w+)\s*\(', exp, flags=re.M)) == set(_lib.EXPERIMENT_SIGNATURES)
    header = re.sub(r'#ifdef LTU_EXPERIMENTS.*?#endif', '', header, flags=re.S)
    declared = set(re.findall(r'^(?:int|long long)\s+(ltu_\w+)\s*\(', header, flags=re.M))
    assert len(declared) >= 35
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ltu_version() >= 1
    # the documents quote the same number of entry points as the header declares
    for doc in ('DESIGN.md', 'INTEGRATION.md'):
        text = open(os.path.join(ROOT, doc)).read()
        quoted = set(int(m) for m in re.findall(r'(\d+) (?:`extern "C"` )?(?:entry points|functions)', text))
        assert quoted == {len(declared)}, (doc, quoted, len(declared))
    # argument counts of the binding table match the C prototypes
    protos = re.findall(r'^(?:int|long long)\s+(ltu_\w+)\s*\(([^;]*)\);', header, flags=re.M | re.S)
    for name, args in protos:
        n = 0 if args.strip() == 'void' else len(args.split(','))
        assert n == len(_lib.SIGNATURES[name]), name
    # the workspace contract (SURVEY 8b error behaviour; round 4's GPU fault was a launch writing past a workspace sized under
    # another geometry): every caller-owned workspace / scratch pointer is followed by its capacity, as a `long long` argument
    cap_after = {'ws': ('ws_floats', 'ws_elems'), 'part_ws': ('ws_floats',), 'lnws1': ('lnws_floats',)}
    with_ws = 0
    for name, args in protos:
        params = [re.sub(r'/\*.*?\*/', '', a, flags=re.S).strip() for a in args.split(',')]
        names = [p.split()[-1].lstrip('*') if p else '' for p in params]
        for i, n in enumerate(names):
            if n in cap_after:
                with_ws += 1
                assert i + 1 < len(names) and names[i + 1] in cap_after[n] and params[i + 1].startswith('long long'), (name, n)
                assert _lib.SIGNATURES[name][i + 1] is _lib.L, name
        if name == 'ltu_loss_fwd':
            assert names[names.index('sums') + 1] == 'sums_floats'
    assert with_ws == 21, with_ws
    # the two launches whose width the caller chooses take it in the size query AND in the launch
    for q, l in (('ltu_linear_wgrad_group_ws_floats', 'ltu_linear_wgrad_group'), ('ltu_upconv_wgrad_ws_floats', 'ltu_upconv_wgrad')):
        for fn in (q, l):
            args = dict(protos)[fn]
            assert re.search(r'\bint blocks\b', args), fn


@pytest.mark.parametrize('dim_output', [2, 3])
def test_state_dict_surface_matches_reference(dim_output):
    from lintransunet_amd.model import get_model_dict
    cfg = O_net.NetConfig(dim_output=dim_output)
    model = get_model_dict('MaskTransUnet')(cfg.num_layers, cfg.roi_size_list, cfg.is_roi_list, 1, dim_output)
    got = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    want = O_net.param_shapes(cfg)
    assert got == want
    assert len(got) == 614
    from oracle import seedgen
    model.load_state_dict(seedgen.seeded_params(want, 1), strict=True)


def test_no_cpu_fallback():
    from lintransunet_amd.model import get_model_dict
    from lintransunet_amd import ops, _lib
    m = get_model_dict('MaskTransUnet')([8, 8, 8, 16, 32], [20, 12, 9, 10, 6], [False, True, True, True, True], 1, 2)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 1, 32, 32, 32))
    with pytest.raises(_lib.LtuError):
        ops.linear(torch.zeros(4, 32), [torch.zeros(32, 32)], [torch.zeros(32)])


def test_schedule_helpers():
    from lintransunet_amd import train
    from oracle import step as O_step
    dw = train.get_dynamic_weight(30)
    for e in (0, 9, 10, 29):
        assert dw[e] == O_step.dynamic_weights(e)
    specs = train.level_specs(5)
    assert list(specs[0]) == ['CrossEntroLoss', 'BalanceDiceLoss'] and list(specs[3]) == ['CrossEntroLoss', 'DiceClassLoss']
    assert len(train.UNUSED_PARAMETERS) == 14


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _reducer_worker(rank, world, port, out):
    import torch.distributed as dist
    from lintransunet_amd.train import GradReducer, broadcast_parameters
    from lintransunet_amd.comm import GlooComm
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    comm = GlooComm()                            # the communicator interface of the RCCL path, on CPU tensors (comm.py)
    assert (comm.world, comm.rank) == (world, rank)
    assert comm.max_float(float(rank)) == world - 1
    torch.manual_seed(rank)                      # different init per rank: broadcast must fix it
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16), torch.nn.Tanh(),
                              torch.nn.Linear(16, 1))
    unused = torch.nn.Linear(3, 3)               # registered, never used in forward (like pos_encoders.1..7)
    net.add_module('unused', unused)
    broadcast_parameters(net, comm)
    red = GradReducer(net, bucket_mb=0.0005, unused=['unused.weight', 'unused.bias'], comm=comm)
    assert len(red.buckets) >= 2
    g = torch.Generator().manual_seed(123)
    x = torch.randn(8, 6, generator=g)
    y = torch.randn(8, 1, generator=g)
    shard = slice(rank * 4, rank * 4 + 4)
    for _ in range(2):                           # two steps: buffers are reused
        red.zero_grad()
        red.prepare()
        loss = ((net[:5](x[shard]) - y[shard]) ** 2).mean()      # per-sample loss, mean over the shard
        loss.backward()
        red.finish()
    # the path bench.py takes at N > 1: backward without hooks (as inside a replayed graph), then reduce_all()
    red.zero_grad()
    ((net[:5](x[shard]) - y[shard]) ** 2).mean().backward()
    red.reduce_all()
    after_replay = [p.grad.clone() for p in list(net.parameters())[:6]]
    red.zero_grad()
    red.prepare()
    ((net[:5](x[shard]) - y[shard]) ** 2).mean().backward()
    red.finish()
    assert all(torch.allclose(a, p.grad, atol=1e-7) for a, p in zip(after_replay, list(net.parameters())[:6]))
    # gradient accumulation (utils/utils_3D_embed_full.py:85-91): two micro-steps of half a shard each, every loss divided by
    # step_times = 2; the first only accumulates (no collective may start), the last one reduces
    red.zero_grad()
    for j in range(2):
        red.prepare(reduce=(j == 1))
        sl = slice(rank * 4 + 2 * j, rank * 4 + 2 * j + 2)
        (((net[:5](x[sl]) - y[sl]) ** 2).mean() / 2).backward()
        if j == 0:
            assert not red.handles
        red.finish()
    assert all(torch.allclose(a, p.grad, atol=1e-6) for a, p in zip(after_replay, list(net.parameters())[:6]))
    # buckets re-assigned in gradient-ready order (last layer first): same gradients, first bucket = the head of the network's tail
    red.rebucket()
    assert red.buckets[0][0] is net[4].bias or red.buckets[0][0] is net[4].weight
    red.zero_grad()
    red.prepare()
    ((net[:5](x[shard]) - y[shard]) ** 2).mean().backward()
    red.finish()
    assert all(torch.allclose(a, p.grad, atol=1e-7) for a, p in zip(after_replay, list(net.parameters())[:6]))
    if rank == 0:
        ref = torch.nn.Sequential(*[m for m in list(net.children())[:5]])
        grads = [p.grad.clone() for p in list(net.parameters())[:6]]
        for p in ref.parameters():
            p.grad = None
        ((ref(x) - y) ** 2).mean().backward()
        err = max((a - p.grad).abs().max().item() for a, p in zip(grads, ref.parameters()))
        out.put((err, unused.weight.grad is None))
    dist.destroy_process_group()


def test_grad_reducer_world2_gloo():
    """sharded batch + mean all-reduce == full-batch gradient; unused parameters are skipped"""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reducer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    err, unused_none = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert err < 1e-6
    assert unused_none


def _run_bench(args, env_extra=None, timeout=180):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_bench_self_launch_dry_run_world2():
    """`python bench.py --gpus 2` with no launcher environment (the form of the driver's command): the parent starts one child per
    rank before touching the GPU, the ranks rendezvous on 127.0.0.1 (gloo stub of the RCCL path), rank 0 prints ONE JSON line"""
    import json
    r = _run_bench(['--gpus', '2', '--steps', '3', '--warmup', '1', '--dry-run'])
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    assert len(r.stdout.strip().splitlines()) == 1, r.stdout          # nothing else on stdout (gloo's connection banner goes to stderr)
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 3 and out['warmup'] == 1 and out['scaling'] == 'weak'
    assert out['config']['parallelism'] == 'dp2'


def test_bench_self_launch_propagates_failure():
    """a rank that dies takes the launch down with a non-zero exit code (no hang on the survivors' barrier)"""
    r = _run_bench(['--gpus', '2', '--dry-run', '--steps', '-1'], env_extra={'LTU_BENCH_FAIL_RANK': '1'}, timeout=120)
    assert r.returncode != 0


def test_bench_parent_does_not_import_torch_before_launch():
    """the self-launching parent must not initialise HIP: it forks its children before importing torch at all"""
    src = open(os.path.join(ROOT, 'bench.py')).read()
    head = src[:src.index('def main():')]
    assert not re.search(r'^import torch|^from torch', head, flags=re.M)
    body = src[src.index('def main():'):]
    assert body.index('self_launch(args)') < body.index('import torch')


def test_comm_surface_without_gpu():
    """LocalComm is a no-op communicator; the RCCL entry points refuse to work before ltu_comm_load (no compute, no GPU)"""
    import ctypes
    from lintransunet_amd import _lib
    from lintransunet_amd import comm as C
    c = C.LocalComm()
    t = torch.ones(4)
    c.allreduce_avg(t).wait()
    c.broadcast(t)
    c.barrier()
    assert c.max_float(2.5) == 2.5 and torch.equal(t, torch.ones(4))
    lib = _lib.load()
    assert lib.ltu_comm_load(None) == -4                       # LTU_E_ARG
    assert lib.ltu_comm_load(b'/nonexistent/librccl.so') == -100        # LTU_E_COMM: not loadable
    buf = (ctypes.c_char * 128)()
    assert lib.ltu_comm_unique_id(ctypes.addressof(buf)) == -100        # not loaded
    with pytest.raises(RuntimeError):
        C.GlooComm()                                           # needs an initialised gloo group


def test_reducer_counts_each_parameter_once():
    """a fused parameter reports from its weight-gradient op AND from autograd's post-accumulate hook (PyTorch fires it even when
    the op returned None for the parameter): a bucket must close when ALL its parameters have reported, not after that many calls"""
    from lintransunet_amd.train import GradReducer

    class FakeComm:
        world, rank = 2, 0

        def __init__(self):
            self.log = []

        def allreduce_avg(self, flat):
            self.log.append((flat.data_ptr(), set(reported)))

            class H:
                def wait(self):
                    pass
            return H()

        def broadcast(self, t, src=0):
            pass
    net = torch.nn.Sequential(*[torch.nn.Linear(4, 4) for _ in range(6)])
    comm = FakeComm()
    red = GradReducer(net, bucket_mb=0.0001, comm=comm, tail_mb=0.0)          # 40 floats per bucket: two layers each
    assert len(red.buckets) == 3
    reported = set()
    red.zero_grad()
    red.prepare()
    for b in red.buckets:
        for p in b:
            reported.add(id(p))
            red._hook(p)              # the op's own report ...
            red._hook(p)              # ... and the post-accumulate hook
    red.finish()
    assert len(comm.log) == 3
    assert all(v == 0 for v in red.pending)
    for (ptr, seen), b, flat in zip(comm.log, red.buckets, red.flat):
        assert ptr == flat.data_ptr() and all(id(p) in seen for p in b)
    # the real thing: a custom Function that writes the gradient itself and returns None still triggers the post-accumulate hook
    calls = []

    class F(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.w = w
            return x * w

        @staticmethod
        def backward(ctx, g):
            calls.append('op')
            return g * ctx.w, None
    w = torch.nn.Parameter(torch.ones(1))
    w.grad = torch.zeros(1)
    w.register_post_accumulate_grad_hook(lambda p: calls.append('post_accumulate'))
    F.apply(torch.ones(3, requires_grad=True), w).sum().backward()
    assert calls in (['op', 'post_accumulate'], ['op'])          # torch 2.10: both - the behaviour the reducer has to be robust against


def test_reducer_tail_bucket_plan():
    """multi-rank bucket plan: buckets of ~bucket_mb in the given order, and the last `tail_mb` of parameters (the gradients that
    arrive last) in a latency-sized bucket of their own; every parameter exactly once, order preserved, also after rebucket()"""
    from lintransunet_amd.train import GradReducer

    class FakeComm:
        world, rank = 2, 0

        def allreduce_avg(self, flat):
            class H:
                def wait(self):
                    pass
            return H()
    net = torch.nn.Sequential(*[torch.nn.Linear(16, 16) for _ in range(8)])          # 8 x (256 + 16) parameters
    red = GradReducer(net, bucket_mb=1024 * 4 / 2 ** 20, tail_mb=300 * 4 / 2 ** 20, comm=FakeComm())      # 1 024-float buckets, 300-float tail
    order = [p for _, p in reversed(list(net.named_parameters()))]
    flat_order = [p for b in red.buckets for p in b]
    assert len(flat_order) == len(order) and all(a is b for a, b in zip(flat_order, order))
    sizes = [sum(p.numel() for p in b) for b in red.buckets]
    assert sizes[-1] <= 300 and len(red.buckets[-1]) >= 1 and all(s >= 1024 for s in sizes[:-2])
    assert sum(sizes) == sum(p.numel() for p in net.parameters())
    for b, f in zip(red.buckets, red.flat):
        assert f.numel() == sum(p.numel() for p in b)
        assert all(p.grad.data_ptr() >= f.data_ptr() and p.grad.data_ptr() < f.data_ptr() + 4 * f.numel() for p in b)
    # a 1-rank reducer keeps the plain plan (no tail bucket)
    red1 = GradReducer(torch.nn.Sequential(*[torch.nn.Linear(16, 16) for _ in range(8)]), bucket_mb=1024 * 4 / 2 ** 20, tail_mb=300 * 4 / 2 ** 20)
    assert sum(p.numel() for p in red1.buckets[-1]) > 300
    # ready order recorded by a backward, then re-bucketing keeps the invariants and bumps the generation
    gen = red.generation
    red.zero_grad()
    red.prepare()
    x = torch.randn(4, 16)
    net(x).sum().backward()
    red.finish()
    assert all(v == 0 for v in red.pending)
    red.rebucket()
    assert red.generation == gen + 1
    assert sorted(id(p) for b in red.buckets for p in b) == sorted(id(p) for p in net.parameters())
    assert red.buckets[0][0] is net[7].bias or red.buckets[0][0] is net[7].weight          # the last layer's gradients arrive first


def test_bench_report_helpers_read_the_committed_profiles():
    """bench.py's second roofline object and the in-step family times are computed from the committed kernel statistics / PMC passes
    (profiles/rNN_*): the helpers parse the newest committed files and the numbers are consistent with each other"""
    import glob
    import importlib.util
    from lintransunet_amd import family_timer
    stats = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_bench_kernel_stats.csv')))
    pmcs = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_step.json')))
    assert stats and pmcs
    fam = family_timer.in_step_ms(stats[-1])
    assert set(fam) >= {'transformer', 'conv3', 'instnorm', 'resample', 'dwconv', 'other'}
    assert 12.0 < sum(fam.values()) < 25.0 and fam['transformer'] > fam['conv3'] > fam['instnorm'] > fam['dwconv']
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    byts = 4353687552                      # 24 M d bytes per layer over the 32 layers of the 128^3 x 2 step
    lk = bench.largest_kernel((1.8, 14, byts), pmcs[-1], stats[-1])
    assert lk['launches'] == 14 and abs(lk['achieved'] - byts / 1.8e-3 / 1e9) < 1e-6
    assert 1.5 < lk['ms_in_step'] < 3.0 and 0.1 < lk['frac_in_step'] < lk['frac'] < 0.5
    assert 1.0 <= lk['traffic'] / byts < 2.0          # PMC traffic of the kernel and its fold against the algorithmic operand bytes
    assert bench.largest_kernel(None, pmcs[-1], stats[-1]) is None

"""The data-parallel path on ONE GPU: a live 1-rank RCCL communicator driven through the C-ABI (lintransunet_amd/comm.py ->
csrc/comm.hip: ltu_comm_load / _unique_id / _init / _allreduce_avg / _broadcast / _destroy), i.e. what every rank does at N > 1
minus the links.  RCCL's kernels launch and are captured; the mean over one rank is the identity, so gradients must come out
unchanged - eagerly, from autograd hooks, and replayed from a step graph with the collectives captured as side branches."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import net as O_net          # noqa: E402
from oracle import seedgen               # noqa: E402
from oracle import step as O_step        # noqa: E402

DEV = 'cuda'
SMALL = dict(num_layers=[8, 8, 8, 16, 32], roi_size_list=[20, 12, 9, 10, 6])


@pytest.fixture(scope='module')
def comm():
    from lintransunet_amd import comm as C
    c = C.RcclComm(torch.device('cuda', 0))
    yield c
    c.close()


def test_allreduce_and_broadcast_one_rank(comm):
    assert (comm.world, comm.rank) == (1, 0)
    x = torch.randn(1 << 20, device=DEV)
    ref = x.clone()
    n0 = comm.calls
    comm.allreduce_avg(x).wait()
    comm.broadcast(x, 0)
    torch.cuda.synchronize()
    assert comm.calls == n0 + 1 and torch.equal(x, ref)
    from lintransunet_amd import _lib
    with pytest.raises(_lib.LtuError):
        comm.allreduce_avg(x.cpu())                      # no silent staging through the host
    with pytest.raises(_lib.LtuError):
        comm.allreduce_avg(x.bfloat16())


def test_captured_allreduce_replays(comm):
    """the collective is a plain enqueue on the communicator's stream: fork / join are stream dependencies, so a capture takes it
    as a side branch and replays it"""
    x = torch.zeros(1 << 16, device=DEV)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        x.add_(1.0)
        comm.allreduce_avg(x).wait()
    torch.cuda.synchronize()
    x.zero_()
    g = torch.cuda.CUDAGraph()
    n0 = comm.calls
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        x.add_(1.0)
        comm.allreduce_avg(x).wait()
        x.mul_(2.0)
    assert comm.calls == n0 + 1
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    assert comm.calls == n0 + 1                          # nothing is re-issued from the host
    assert torch.equal(x, torch.full_like(x, 62.0))      # ((((0+1)*2+1)*2+1)*2+1)*2+1)*2


def test_graphed_step_with_captured_collectives_matches_plain_step(comm):
    from lintransunet_amd import train
    from lintransunet_amd.model import get_model_dict
    cfg = O_net.NetConfig(**SMALL)
    x = seedgen.seeded_volume((2, 1, 32, 32, 32), 1).to(DEV)
    lab = seedgen.seeded_label((2, 1, 32, 32, 32), 2).to(DEV)
    w = O_step.dynamic_weights(0)

    def build(c):
        m = get_model_dict('MaskTransUnet')(cfg.num_layers, cfg.roi_size_list, cfg.is_roi_list, 1, 2, dropout=0.0,
                                            act_dtype=torch.bfloat16)      # the bf16 path has no atomics: runs are bit-reproducible
        m.load_state_dict(seedgen.seeded_params(O_net.param_shapes(cfg), 100), strict=True)
        m = m.to(DEV).train()
        red = train.GradReducer(m, bucket_mb=0.25, unused=train.UNUSED_PARAMETERS, comm=c, force_collectives=c is not None)
        return m, red

    def whole(m):
        return torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None])
    m0, r0 = build(None)
    r0.zero_grad()
    train.train_step(m0, x, lab, w, reducer=r0)
    torch.cuda.synchronize()
    ref = whole(m0)
    m1, r1 = build(comm)
    assert len(r1.flat) >= 3 and r1.world == 2
    n0 = comm.calls
    step = train.GraphedStep(m1, x, lab, w, r1, overlap='graph')
    built = comm.calls - n0
    assert built >= 3 * len(r1.flat)                     # two warm-up steps and the capture each enqueue every bucket once
    for _ in range(3):
        step(x, lab)
    torch.cuda.synchronize()
    assert comm.calls - n0 == built                      # replays re-issue nothing
    err = ((whole(m1) - ref).norm() / ref.norm()).item()
    assert err <= 1e-6, err
    # the same step as linear segments with eager collectives in between
    seg = train.GraphedStep(m1, x, lab, w, r1, overlap='segments')
    n1 = comm.calls
    for _ in range(2):
        seg(x, lab)
    torch.cuda.synchronize()
    assert comm.calls - n1 == 2 * len(r1.flat)
    assert ((whole(m1) - ref).norm() / ref.norm()).item() <= 1e-6
    # eager hook path and after-replay path through the same communicator
    r1.zero_grad()
    train.train_step(m1, x, lab, w, reducer=r1)
    torch.cuda.synchronize()
    assert ((whole(m1) - ref).norm() / ref.norm()).item() <= 1e-6


def test_collectives_see_complete_buckets():
    """WHERE a bucket's collective is issued (eagerly from the gradient hooks, or as a captured side branch): a stand-in communicator
    snapshots the bucket on its own stream at that point; after the step every snapshot must equal the final bucket, i.e. no gradient
    of the bucket was written after its all-reduce was forked.  (Round 3: with every fused parameter counted twice the buckets closed
    half-way - invisible to a 1-rank all-reduce, which is the identity.)"""
    from lintransunet_amd import train
    from lintransunet_amd.model import get_model_dict

    class SnapshotComm:
        world, rank = 2, 0

        def __init__(self):
            self.stream = torch.cuda.Stream()
            self.snaps = {}

        def allreduce_avg(self, flat, also=None):
            self.stream.wait_stream(torch.cuda.current_stream())
            if also is not None:                 # the weight-gradient queue's side stream contributed to this bucket too
                self.stream.wait_stream(also)
            snap = self.snaps.setdefault(flat.data_ptr(), torch.empty_like(flat))
            with torch.cuda.stream(self.stream):
                snap.copy_(flat)
            stream = self.stream

            class H:
                def wait(self):
                    torch.cuda.current_stream().wait_stream(stream)
            return H()

        def broadcast(self, t, src=0):
            pass
    cfg = O_net.NetConfig(**SMALL)
    x = seedgen.seeded_volume((2, 1, 32, 32, 32), 1).to(DEV)
    lab = seedgen.seeded_label((2, 1, 32, 32, 32), 2).to(DEV)
    w = O_step.dynamic_weights(0)
    m = get_model_dict('MaskTransUnet')(cfg.num_layers, cfg.roi_size_list, cfg.is_roi_list, 1, 2, dropout=0.0, act_dtype=torch.bfloat16)
    m.load_state_dict(seedgen.seeded_params(O_net.param_shapes(cfg), 100), strict=True)
    m = m.to(DEV).train()
    comm = SnapshotComm()
    red = train.GradReducer(m, bucket_mb=0.25, unused=train.UNUSED_PARAMETERS, comm=comm)
    assert len(red.flat) >= 5

    def check(tag):
        torch.cuda.synchronize()
        for bi, f in enumerate(red.flat):
            assert f.abs().sum().item() > 0
            assert torch.equal(comm.snaps[f.data_ptr()], f), f'{tag}: bucket {bi} was reduced before its last gradient arrived'
        assert all(v == 0 for v in red.pending), red.pending          # every parameter counted exactly once
    red.zero_grad()
    train.train_step(m, x, lab, w, reducer=red)                        # eager hooks
    check('hooks')
    red.rebucket()                                                     # gradient-ready order, as bench.py uses it
    comm.snaps.clear()
    red.zero_grad()
    train.train_step(m, x, lab, w, reducer=red)
    check('hooks, ready order')
    for mode in ('graph', 'segments'):                                 # captured side branches; linear segments + eager collectives
        comm.snaps.clear()
        step = train.GraphedStep(m, x, lab, w, red, overlap=mode)
        if mode == 'segments':
            closing = [bi for _, _, bi in step.graphs[(True, True)][0] if bi is not None]      # segments that complete a bucket
            assert len(closing) + len(step.graphs[(True, True)][3]) == len(red.flat), closing
            assert any(kind == 'side' for _, kind, _ in step.graphs[(True, True)][0])        # weight gradients as graphs of their own
        for _ in range(2):
            for s in comm.snaps.values():
                s.zero_()
            step(x, lab)
            check(mode)


def test_two_ranks_on_one_gpu_average_to_the_full_batch_gradient():
    """tools/rehearse_two_ranks.py: two real data-parallel processes on this GPU (the gradient exchange staged through gloo by a
    test communicator, RCCL refuses two ranks on one device): ranks end up with identical gradients, and their mean equals the
    gradient of the full batch of four on one process - for the hook path, the graph + after-replay path and accumulation"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'rehearse_two_ranks.py')], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'ok' in r.stdout.splitlines()[-1]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs: RCCL refuses two ranks on one device')
def test_two_gpus_through_rccl():
    """tools/rehearse_two_ranks.py --rccl: the first real N > 1 execution of the product communicator wherever two devices are visible -
    RcclComm over a gloo control plane in two fresh processes: known-answer all-reduce (ranks contribute 1 and 3, every rank reads
    2) and broadcast, then hooks / graph + after / segments (with the weight-gradient side stream) / accumulation: identical
    gradients on both ranks and mean of the rank gradients = full-batch gradient of one process"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'rehearse_two_ranks.py'), '--rccl'], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'ok' in r.stdout.splitlines()[-1]


def test_bench_with_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2 --test-comm staged`: the driver-facing multi-GPU flow (self-launch, gloo rendezvous, parameter
    broadcast, gradient hooks, re-bucketing, the step as graph segments with the collectives between them, barrier / max-over-ranks
    timing, ONE JSON line from rank 0) with two processes sharing this GPU and the exchange staged through the host"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--test-comm', 'staged', '--size', '32', '--steps', '2',
                        '--warmup', '1', '--no-cpu-baseline', '--no-families'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1 and len(r.stdout.strip().splitlines()) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['global_batch'] == 4 and out['config']['parallelism'] == 'dp2'
    assert out['config']['launch'].startswith('hip-graph replay') and 'segments' in out['config']['allreduce']


def test_bench_default_path_with_family_table():
    """the driver's invocation shape of bench.py on one GPU (HIP-graph replay as linear segments + side-stream weight gradients, the
    chain-kernel roofline object AND the per-family table, whose replay re-issues every recorded C-ABI call of the step: it runs
    under the launch-geometry knobs of the capture - sized differently, the replayed weight-gradient kernels overran their
    workspaces and the benchmark died of a GPU fault in round 4), small patch, no CPU baseline"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--size', '64', '--steps', '2', '--warmup', '1', '--no-cpu-baseline'],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['config']['launch'].startswith('hip-graph replay (segments')
    fams = {f['family']: f for f in out['roofline']['families']}
    assert {'transformer', 'conv3', 'instnorm'} <= set(fams) and all(f['ms_per_step'] > 0 for f in fams.values())

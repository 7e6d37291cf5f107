"""Rehearsal of bench.py's multi-GPU code path on ONE GPU with a live 1-rank RCCL communicator driven through the C-ABI
(lintransunet_amd/comm.py: RcclComm -> ltu_comm_init / ltu_comm_allreduce_avg; what every rank does at N > 1):
  0. GraphedStep(overlap='segments'): the step as linear graph segments cut where a bucket closes, collectives issued eagerly between them;
  1. GraphedStep(overlap='graph'): the bucket all-reduces are CAPTURED inside the step graph (side branches on the communicator's
     stream, forked where a bucket's last gradient is produced);
  2. GraphedStep(overlap='after'): the collectives are issued after the replay (`reduce_all`);
  3. the eager hook path;
  4. a reducer without any communicator (`none`): the price of the fork / join edges and of the collectives themselves.
The first three must leave identical gradients in the flat buckets (dropout off, same batch).  Also rehearses accumulation over
2 micro-steps (reduce on the last only), re-bucketing, and CYCLES build / capture / replay cycles back to back (there is no
watchdog thread any more, so no sleep separates them).
run: python tools/check_dist_graph.py [bf16]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd import train
from lintransunet_amd import comm as C
from lintransunet_amd.model import get_model_dict
from oracle import seedgen, step as O_step

torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
# fp32 storage: run-to-run differences are 1e-6
DT = torch.bfloat16 if 'bf16' in sys.argv else torch.float32
CYCLES = int(os.environ.get('CYCLES', '3'))
comm = C.RcclComm(dev)             # 1 rank: RCCL's kernels still launch (and are captured); the average of one is the identity


def build(with_comm=True):
    torch.manual_seed(5)
    m = get_model_dict('MaskTransUnet')([8, 8, 8, 16, 32], [20, 12, 9, 10, 6], [False, True, True, True, True], 1, 2,
                                        dropout=0.0, act_dtype=DT).to(dev).train()
    red = train.GradReducer(m, bucket_mb=0.25, tail_mb=0.01, unused=train.UNUSED_PARAMETERS, comm=comm if with_comm else None,
                            force_collectives=with_comm)
    if with_comm:
        train.broadcast_parameters(m, comm)
    return m, red


def grads_of(m):
    """the whole gradient in registration order as one vector (bucket plans differ between the runs; single tensors whose gradient
    is mathematically zero - conv biases in front of InstanceNorm, the key bias - hold only rounding noise)"""
    return [torch.cat([p.grad.detach().flatten() for p in m.parameters() if p.grad is not None]).clone()]


x = seedgen.seeded_volume((2, 1, 32, 32, 32), 1).to(dev)
lab = seedgen.seeded_label((2, 1, 32, 32, 32), 2).to(dev)
w = O_step.dynamic_weights(0)
results = {}
for mode in ('segments', 'graph', 'after', 'none'):
    m, red = build(mode != 'none')
    n0 = comm.calls
    step = train.GraphedStep(m, x, lab, w, red, overlap='after' if mode == 'none' else mode)
    built = comm.calls - n0
    n0 = comm.calls
    torch.cuda.synchronize()
    for _ in range(3):
        step(x, lab)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        totals, _ = step(x, lab)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20 * 1e3
    per = (comm.calls - n0) / 23
    results[mode] = grads_of(m)
    print(f'{mode:6s}: {len(red.flat)} buckets, collectives enqueued while building (warm-up + capture) {built}, per replay {per:.1f}, '
          f'{dt:.3f} ms/step, loss {sum(t.item() for t in totals):.6f}', flush=True)
    if mode == 'graph':
        assert per == 0, 'captured collectives must not be re-issued from the host'
        assert built >= 3 * len(red.flat)
    elif mode == 'segments':
        assert per == len(red.flat) and len(step.graphs[(True, True)][0]) >= 2        # eager collectives between linear segments
    elif mode == 'after':
        assert per == len(red.flat)
m, red = build()
red.zero_grad()
train.train_step(m, x, lab, w, reducer=red)
torch.cuda.synchronize()
results['eager'] = grads_of(m)
# buckets in gradient-ready order, then the captured step again: same gradients (summed over all parameters)
tot_before = sum(f.double().sum().item() for f in red.flat)
names = {id(p): n for n, p in m.named_parameters()}
red.rebucket()
print('ready-order buckets:', [(names[id(b[0])].split('.')[0:3], len(b), sum(p.numel() for p in b)) for b in red.buckets][:4], '...')
step = train.GraphedStep(m, x, lab, w, red, overlap='graph')
step(x, lab)
torch.cuda.synchronize()
tot_after = sum(f.double().sum().item() for f in red.flat)
assert abs(tot_after - tot_before) <= 1e-3 * abs(tot_before) + 1e-6, (tot_before, tot_after)
for mode in ('segments', 'after', 'eager', 'none'):
    worst = max(((a - b).norm() / b.norm().clamp_min(1e-20)).item() for a, b in zip(results['graph'], results[mode]))
    print(f'gradients graph vs {mode}: whole-gradient rel-L2 {worst:.2e}')
    assert worst < (5e-2 if DT == torch.bfloat16 else 1e-4)
# accumulation: 2 micro-steps, collectives only inside the last one's graph
m, red = build()
step = train.GraphedStep(m, x, lab, w, red, step_times=2, overlap='graph')
for j in range(2):
    step(x, lab, micro=j)
torch.cuda.synchronize()
acc = grads_of(m)
worst = max(((a - b).norm() / b.norm().clamp_min(1e-20)).item() for a, b in zip(acc, results['graph']))
print(f'2 accumulated half-weight micro-steps vs one step: whole-gradient rel-L2 {worst:.2e}; graphs {sorted(step.graphs)}')
assert worst < (5e-2 if DT == torch.bfloat16 else 1e-4)
# storage that moves in the middle of an accumulation cycle must be refused (a re-capture would zero the accumulated buckets)
from lintransunet_amd import optim
opt = optim.FusedAdamW(red)
step(x, lab, micro=0)
opt2 = optim.FusedAdamW(red)          # re-homes the parameters again
try:
    step(x, lab, micro=1)
    raise SystemExit('re-capture in the middle of an accumulation cycle was not refused')
except RuntimeError as e:
    print('refused as expected:', str(e)[:80])
# repeated build / capture / replay cycles, back to back
for c in range(CYCLES):
    m, red = build()
    step = train.GraphedStep(m, x, lab, w, red, overlap='graph')
    for _ in range(3):
        step(x, lab)
    torch.cuda.synchronize()
print('cycles', CYCLES)
comm.close()
print('ok')

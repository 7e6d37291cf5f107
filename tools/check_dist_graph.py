"""Rehearsal of bench.py's multi-GPU code path on ONE GPU with a live 1-rank RCCL process group (what every rank does at N > 1):
  1. GraphedStep(overlap='graph'): the bucket all-reduces are CAPTURED inside the step graph (side branches on the process
     group's stream, forked where a bucket's last gradient is produced);
  2. GraphedStep(overlap='after'): the collectives are issued after the replay (`reduce_all`);
  3. the eager hook path.
The three must leave identical gradients in the flat buckets (dropout off, same batch); the captured graph must contain the RCCL
kernels (checked by counting graph nodes through the debug dump when available, and by the timing of a replay without a
following reduce_all).  Also rehearses accumulation over 2 micro-steps (reduce on the last only).
run: python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 tools/check_dist_graph.py"""
import os, sys, time, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd import train
from lintransunet_amd.model import get_model_dict
from oracle import seedgen, step as O_step

torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
# fp32 storage: run-to-run differences are 1e-6 (bf16 storage turns the last-bit noise of atomic sums into 1e-2 gradient differences)
DT = torch.bfloat16 if 'bf16' in sys.argv else torch.float32
CYCLES = int(os.environ.get('CYCLES', '1'))
dist.init_process_group('nccl', device_id=dev)


def build():
    torch.manual_seed(5)
    m = get_model_dict('MaskTransUnet')([8, 8, 8, 16, 32], [20, 12, 9, 10, 6], [False, True, True, True, True], 1, 2,
                                        dropout=0.0, act_dtype=DT).to(dev).train()
    train.broadcast_parameters(m)
    red = train.GradReducer(m, bucket_mb=0.25, unused=train.UNUSED_PARAMETERS)
    red.world = 2           # force the collective path
    red.avg = True          # ... with RCCL's in-collective average, as at N > 1 (1 rank: the average of one)
    return m, red


x = seedgen.seeded_volume((2, 1, 32, 32, 32), 1).to(dev)
lab = seedgen.seeded_label((2, 1, 32, 32, 32), 2).to(dev)
w = O_step.dynamic_weights(0)
calls = {'n': 0}
orig = dist.all_reduce


import threading
seen = set()


def counting(*a, **k):
    calls['n'] += 1
    key = (threading.current_thread().name, torch.cuda.is_current_stream_capturing(), torch.cuda.current_stream().cuda_stream)
    if key not in seen:
        seen.add(key)
        print('all_reduce from', key, flush=True)
    return orig(*a, **k)


dist.all_reduce = counting
results = {}
for mode in ('graph', 'after'):
    m, red = build()
    calls['n'] = 0
    step = train.GraphedStep(m, x, lab, w, red, overlap=mode)
    captured_calls = calls['n']
    calls['n'] = 0
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        totals, _ = step(x, lab)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5 * 1e3
    results[mode] = [f.clone() for f in red.flat]
    print(f'{mode:6s}: {len(red.flat)} buckets, all_reduce calls while building (warm-up + capture) {captured_calls}, '
          f'per replay {calls["n"] / 5:.1f}, {dt:.2f} ms/step, loss {sum(t.item() for t in totals):.6f}')
    if mode == 'graph':
        assert calls['n'] == 0, 'captured collectives must not be re-issued from the host'
        assert captured_calls >= 3 * len(red.flat)
    else:
        assert calls['n'] == 5 * len(red.flat)
m, red = build()
red.zero_grad()
train.train_step(m, x, lab, w, reducer=red)
torch.cuda.synchronize()
results['eager'] = [f.clone() for f in red.flat]
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
for mode in ('after', 'eager'):
    worst = max(((a - b).norm() / b.norm().clamp_min(1e-20)).item() for a, b in zip(results['graph'], results[mode]))
    print(f'gradients graph vs {mode}: worst bucket rel-L2 {worst:.2e}')
    assert worst < (5e-2 if DT == torch.bfloat16 else 1e-4)
# accumulation: 2 micro-steps, collectives only inside the last one's graph
m, red = build()
calls['n'] = 0
step = train.GraphedStep(m, x, lab, w, red, step_times=2, overlap='graph')
for j in range(2):
    step(x, lab, micro=j)
torch.cuda.synchronize()
acc = [f.clone() for f in red.flat]
worst = max(((a - b).norm() / b.norm().clamp_min(1e-20)).item() for a, b in zip(acc, results['graph']))
print(f'2 accumulated half-weight micro-steps vs one step: worst bucket rel-L2 {worst:.2e}; graphs {sorted(step.graphs)}')
assert worst < (5e-2 if DT == torch.bfloat16 else 1e-4)
# repeated build / capture / replay cycles: the watchdog must survive every capture
for c in range(CYCLES - 1):
    m, red = build()
    step = train.GraphedStep(m, x, lab, w, red, overlap='graph')
    for _ in range(3):
        step(x, lab)
    torch.cuda.synchronize()
    time.sleep(0.15)
print('cycles', CYCLES)
dist.barrier()
print('ok')
dist.destroy_process_group()

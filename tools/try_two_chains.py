"""Experiment: the per-GPU batch of two patches as TWO independent one-patch steps on two stream pairs (every op of the network is
per-sample), offset against each other so that one chain's latency-bound phases (the d = 256 levels) meet the other's
machine-filling ones.  Step time is ~4.3 ms + 4.2 ms x patches (9.45 / 12.7 / 21.1 ms at 1 / 2 / 4 patches): two interleaved
one-patch chains could approach 2 x 4.2 ms + little.  Measures the period per pair of patches: sequential, concurrent from the same
start, concurrent with an offset, free-running (no sync between iterations) and with a sync every iteration (what a real optimizer step
would force)."""
import copy, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lintransunet_amd.model import get_model_dict
from lintransunet_amd import train, ops

dev = torch.device('cuda:0')
torch.manual_seed(1234)
def make():
    return get_model_dict('MaskTransUnet')([16, 32, 64, 128, 256], [100, 65, 40, 25, 10], [False, True, True, True, True], 1, 2,
                                           dropout=0.3, act_dtype=torch.bfloat16).to(dev).train()
m1 = make()
m2 = make()
m2.load_state_dict(m1.state_dict())
weights = train.get_dynamic_weight(1)[0]
x, lab = bench.synthetic_batch(2, (128,) * 3, 100, dev)
sA = torch.cuda.Stream()
sB = ops.concurrent_stream(dev, [torch.cuda.current_stream(), sA])
steps = []
for m, s, i in ((m1, sA, 0), (m2, sB, 1)):
    red = train.GradReducer(m, unused=train.UNUSED_PARAMETERS)
    with torch.cuda.stream(s):
        red.zero_grad()
        train.train_step(m, x[i:i + 1], lab[i:i + 1], weights, reducer=red)
        torch.cuda.synchronize()
        red.rebucket()
        steps.append(train.GraphedStep(m, x[i:i + 1], lab[i:i + 1], weights, red, overlap='segments'))
g1, g2 = steps
g2.wq_stream = ops.concurrent_stream(dev, [sA, sB, g1.wq_stream])
torch.cuda.synchronize()

def run(n, mode, offset_cycles=0, sync_each=False):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if offset_cycles and not sync_each:
        with torch.cuda.stream(sB):
            torch.cuda._sleep(offset_cycles)
    for _ in range(n):
        if mode == 'seq':
            with torch.cuda.stream(sA):
                g1(); g2()
        else:
            with torch.cuda.stream(sA):
                g1()
            with torch.cuda.stream(sB):
                if offset_cycles and sync_each:
                    torch.cuda._sleep(offset_cycles)
                g2()
        if sync_each:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for _ in range(3):
    run(3, 'seq')
print('two one-patch steps, sequential on one stream: %.3f ms per pair' % run(20, 'seq'))
print('concurrent, same start, sync every pair:        %.3f ms per pair' % run(20, 'par', 0, True))
for us in (1000, 2000, 3000, 4000, 5000):
    cyc = int(us * 2400 * 0.042 / 0.042)        # _sleep counts cycles of the 100 MHz-ish clock domain? calibrated below
    print('concurrent, offset ~%d "us" of spin, sync every pair: %.3f ms per pair' % (us, run(20, 'par', us * 100, True)))
print('concurrent, free-running (no sync), no offset:  %.3f ms per pair' % run(40, 'par', 0, False))
for us in (2000, 4000):
    print('concurrent, free-running, initial offset %d: %.3f ms per pair' % (us, run(40, 'par', us * 100, False)))

"""Is the GPU idle between two replays of the step graph?  (GPU box)
  loop      : K x step() back to back, one synchronize at the end            -> ms per step as bench.py measures it
  device    : one replay between two events, host far ahead                  -> the graph's own duration on the device
  host idle : host time of graph.replay() with the GPU idle
  host busy : host time of graph.replay() issued while the previous replay of the same graph is still running
  ping-pong : two captures of the same step sharing the memory pool, replayed alternately (LTU_TWO_GRAPHS)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lintransunet_amd.model import get_model_dict
from lintransunet_amd import train

dev = torch.device('cuda:0')
torch.manual_seed(1234)
model = get_model_dict('MaskTransUnet')([16, 32, 64, 128, 256], [100, 65, 40, 25, 10], [False, True, True, True, True], 1, 2,
                                        dropout=0.3, act_dtype=torch.bfloat16).to(dev).train()
red = train.GradReducer(model, bucket_mb=32.0, unused=train.UNUSED_PARAMETERS)
weights = train.get_dynamic_weight(1)[0]
x, lab = bench.synthetic_batch(2, (128,) * 3, 100, dev)
for _ in range(2):
    red.zero_grad()
    train.train_step(model, x, lab, weights, reducer=red)
red.rebucket()
step = train.GraphedStep(model, x, lab, weights, red)
for _ in range(3):
    step(x, lab)
torch.cuda.synchronize()
K = 20


def loop(fn, k=K):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


print(f'loop, step(x, lab)          : {loop(lambda: step(x, lab)):.3f} ms per step')
print(f'loop, step() without copies : {loop(lambda: step()):.3f} ms per step')
g = step.graphs[(True, True)][0][0][0]
print(f'loop, bare graph.replay()   : {loop(g.replay):.3f} ms per step')
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
print(f'device, one replay between events: {e0.elapsed_time(e1):.3f} ms')
evs = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
torch.cuda.synchronize()
for i in range(5):
    evs[i].record(); g.replay()
evs[5].record(); torch.cuda.synchronize()
print('device, 5 replays with an event between each: ' + ' '.join(f'{evs[i].elapsed_time(evs[i + 1]):.3f}' for i in range(5)))
torch.cuda.synchronize()
t0 = time.perf_counter(); g.replay(); t1 = time.perf_counter(); g.replay(); t2 = time.perf_counter(); g.replay(); t3 = time.perf_counter()
torch.cuda.synchronize()
print(f'host time of replay(): GPU idle {1e3 * (t1 - t0):.3f} ms, previous replay in flight {1e3 * (t2 - t1):.3f} ms, again {1e3 * (t3 - t2):.3f} ms')
t0 = time.perf_counter(); sig = step._signature(); t1 = time.perf_counter()
print(f'host time of _signature(): {1e3 * (t1 - t0):.3f} ms')

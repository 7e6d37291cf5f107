"""How long does the HOST take to issue one replayed step (the loop of hipGraphLaunch calls over the linear segments and the side
batches), against what the step takes on the GPU?  If the two are close the step is launch-bound on the host and gaps open on
the main queue wherever the GPU catches up (seen in the kernel trace as idle stretches of the main queue behind a side batch)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lintransunet_amd.model import get_model_dict
from lintransunet_amd import train

dev = torch.device('cuda:0')
torch.manual_seed(1234)
model = get_model_dict('MaskTransUnet')([16, 32, 64, 128, 256], [100, 65, 40, 25, 10], [False, True, True, True, True], 1, 2,
                                        dropout=0.3, act_dtype=torch.bfloat16).to(dev).train()
weights = train.get_dynamic_weight(1)[0]
x, lab = bench.synthetic_batch(2, (128,) * 3, 100, dev)
red = train.GradReducer(model, unused=train.UNUSED_PARAMETERS)
red.zero_grad()
train.train_step(model, x, lab, weights, reducer=red)
torch.cuda.synchronize()
red.rebucket()
g = train.GraphedStep(model, x, lab, weights, red, overlap=os.environ.get('MODE', 'segments'))
for _ in range(5):
    g()
torch.cuda.synchronize()
segs = g.graphs[(True, True)][0]
print('segments:', len(segs), 'main', sum(1 for s in segs if s[0] is not None and s[1] != 'side'), 'side', sum(1 for s in segs if s[1] == 'side'),
      'join', sum(1 for s in segs if s[1] == 'join'))
for reps in (1, 10):
    host, total = [], []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            g()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host.append((t1 - t0) / reps * 1e3)
        total.append((t2 - t0) / reps * 1e3)
    print('%2d steps back to back: host issue %.3f ms per step, until the GPU is done %.3f ms per step' % (reps, min(host), min(total)))
# per-segment host time of one step
import collections
torch.cuda.synchronize()
main, side = torch.cuda.current_stream(dev), g.wq_stream
times = []
for graph, kind, bi in segs:
    t0 = time.perf_counter()
    if kind == 'join':
        main.wait_stream(side)
    elif graph is not None:
        if kind == 'side':
            side.wait_stream(main)
            with torch.cuda.stream(side):
                graph.replay()
        else:
            graph.replay()
    times.append(((time.perf_counter() - t0) * 1e6,))
main.wait_stream(side)
torch.cuda.synchronize()
print('host us per segment:', ' '.join('%s%.0f' % ('S' if s[1] == 'side' else 'J' if s[1] == 'join' else 'm', t[0]) for s, t in zip(segs, times)))


# per-main-segment GPU time (events at the end of every main segment), with and without the side batches in between
def run(with_side):
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True)]
    evs[0].record(main)
    for graph, kind, bi in segs:
        if kind == 'join':
            main.wait_stream(side)
        elif graph is not None:
            if kind == 'side':
                if with_side:
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        graph.replay()
            else:
                graph.replay()
                e = torch.cuda.Event(enable_timing=True)
                e.record(main)
                evs.append(e)
    main.wait_stream(side)
    torch.cuda.synchronize()
    return [evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(len(evs) - 1)]


for _ in range(2):
    a, b = run(True), run(False)
a = [min(x) for x in zip(*[run(True) for _ in range(5)])]
b = [min(x) for x in zip(*[run(False) for _ in range(5)])]
print('main segments, us (with side batches | main chain alone | difference):')
for i, (x, y) in enumerate(zip(a, b)):
    print('  segment %2d: %7.0f | %7.0f | %+6.0f' % (i, x, y, x - y))
print('  total      : %7.0f | %7.0f | %+6.0f' % (sum(a), sum(b), sum(a) - sum(b)))

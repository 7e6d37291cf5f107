"""Does the latency-bound part of the step overlap with itself?  Two independent B = 1 training steps (two model copies, two
captured GraphedSteps) replayed on two streams at once, against one B = 1 step alone and the benchmarked B = 2 step.
usage: python tools/two_chains.py [steps]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd import train, data
from lintransunet_amd.model import get_model_dict

dev = torch.device('cuda:0')
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
weights = train.get_dynamic_weight(1)[0]


def make(batch, seed):
    torch.manual_seed(1234)
    m = get_model_dict('MaskTransUnet')([16, 32, 64, 128, 256], [100, 65, 40, 25, 10], [False, True, True, True, True], 1, 2,
                                        dropout=0.3, act_dtype=torch.bfloat16).to(dev).train()
    red = train.GradReducer(m, unused=train.UNUSED_PARAMETERS)
    x, y = data.synthetic_patches(batch, (128,) * 3, seed, dev, n_classes=2)
    return train.GraphedStep(m, x, y, weights, red)


def timed(fn, n=steps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


from lintransunet_amd import ops
sa = torch.cuda.Stream(dev)
sb = ops.concurrent_stream(dev, [sa])
probe, extra = ops.concurrent_stream, [sb]
ops.concurrent_stream = lambda d, avoid: probe(d, list(avoid) + extra)      # four streams, four hardware queues
with torch.cuda.stream(sa):
    a = make(1, 100)
extra[:] = [sa] + ([a.wq_stream] if a.wq_stream is not None else [])
with torch.cuda.stream(sb):
    b = make(1, 101)
ops.concurrent_stream = probe
torch.cuda.synchronize()
print('side streams concurrent:', getattr(a.wq_stream, 'ltu_concurrent', None), getattr(b.wq_stream, 'ltu_concurrent', None), flush=True)


def one():
    with torch.cuda.stream(sa):
        a()


def both():
    with torch.cuda.stream(sa):
        a()
    with torch.cuda.stream(sb):
        b()


print(f'B=1 alone            : {timed(one):7.3f} ms', flush=True)
print(f'two B=1 chains at once: {timed(both):7.3f} ms  (per pair of patches)', flush=True)
del a, b
c = make(2, 100)
print(f'B=2 one chain         : {timed(lambda: c()):7.3f} ms', flush=True)

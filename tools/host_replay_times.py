"""Where the main chain waits at the hand-over of a weight-gradient batch, by experiment on the real step: the replay loop of
train.GraphedStep.__call__ with an event at the end of every main segment, under variants of HOW the side batch is handed over.
Prints the step time and the GPU time of the main segments that follow the three large batches.  usage: host_replay_times.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd.model import get_model_dict
from lintransunet_amd import train
import bench
dev = torch.device('cuda', 0)
torch.manual_seed(1234)
model = get_model_dict('MaskTransUnet')([16, 32, 64, 128, 256], [100, 65, 40, 25, 10], [False, True, True, True, True], 1, 2, dropout=0.3,
                                        act_dtype=torch.bfloat16).to(dev).train()
reducer = train.GradReducer(model, bucket_mb=32.0, unused=train.UNUSED_PARAMETERS)
batches = [bench.synthetic_batch(2, (128,) * 3, 100 + i, dev, 2) for i in range(2)]
w = train.get_dynamic_weight(1)[0]
for i in range(2):
    reducer.zero_grad(); train.train_step(model, *batches[i], w, reducer=reducer)
reducer.rebucket()
g = train.GraphedStep(model, batches[0][0], batches[0][1], w, reducer)
segs = g.graphs[(True, True)][0]
main, side = torch.cuda.current_stream(dev), g.wq_stream
side2 = torch.cuda.Stream()
dummy = torch.zeros(64, device=dev)
# the first large weight-gradient batch of backward = the side graph with the most kernels; a stand-in graph of one tiny kernel
side_idx = [i for i, (gr, kind, bi) in enumerate(segs) if kind == 'side' and gr is not None]
mains = [i for i, (gr, kind, bi) in enumerate(segs) if kind == 'main' and gr is not None]
NTH = int(os.environ.get('BIG_AFTER_MAIN', '6'))       # the side graph right behind the NTH main segment (6: the long one that ends where backward leaves ROI bridge 1)
BIG = min(i for i in side_idx if i > mains[NTH - 1])
tiny = torch.cuda.CUDAGraph()
with torch.cuda.stream(side2):
    dummy.add_(1.0)
    torch.cuda.synchronize()
    tiny.capture_begin(); dummy.add_(1.0); tiny.capture_end()
med = torch.cuda.CUDAGraph()
bigbuf = torch.zeros(64 << 20, device=dev)
with torch.cuda.stream(side2):
    bigbuf.add_(1.0)
    torch.cuda.synchronize()
    med.capture_begin()
    for _ in range(10):
        bigbuf.add_(1.0)
    med.capture_end()


def replay(variant, marks=None):
    pending = None
    k = 0
    for si, (gr, kind, bi) in enumerate(segs):
        if si == BIG and variant == 'skip_big':
            continue
        if si == BIG and variant == 'tiny_big':
            gr = tiny
        if si == BIG and variant == 'stream_big':
            gr = med
        if kind == 'join':
            main.wait_stream(side)
            if variant == 'two_side_streams':
                main.wait_stream(side2)
        elif gr is None:
            continue
        elif kind == 'side':
            s = side2 if (variant == 'two_side_streams' and k % 2) else side
            k += 1
            if variant == 'side_after_next_main':
                ev = torch.cuda.Event(); ev.record(main)
                pending = (gr, ev, s)
                continue
            if not (si == BIG and variant == 'nowait_big'):
                s.wait_stream(main)
            if variant == 'dummy_kernel':
                dummy.add_(1.0)
            with torch.cuda.stream(s):
                gr.replay()
        else:
            gr.replay()
            if pending is not None:
                pg, ev, s = pending
                pending = None
                s.wait_event(ev)
                with torch.cuda.stream(s):
                    pg.replay()
            if marks is not None:
                e = torch.cuda.Event(enable_timing=True); e.record(main); marks.append(e)
    if pending is not None:
        pg, ev, s = pending
        s.wait_event(ev)
        with torch.cuda.stream(s):
            pg.replay()
    main.wait_stream(side)
    if variant == 'two_side_streams':
        main.wait_stream(side2)


for variant in (sys.argv[1:] or ['baseline', 'side_after_next_main', 'dummy_kernel', 'two_side_streams', 'baseline']):
    for _ in range(3):
        replay(variant)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        replay(variant)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    marks = []
    e0 = torch.cuda.Event(enable_timing=True); e0.record(main)
    replay(variant, marks)
    torch.cuda.synchronize()
    ts = [e0.elapsed_time(m) for m in marks]
    d = [b - a for a, b in zip([0.0] + ts, ts)]
    print(f'{variant:22s} {ms:7.3f} ms/step; main segments (ms): ' + ' '.join(f'{x:.2f}' for x in d), flush=True)

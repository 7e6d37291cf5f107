"""Which kernels of a weight-gradient batch starve the main chain behind the hand-over?  The C-ABI calls of the captured step are
recorded (family_timer), the first large batch of weight-gradient calls of backward (ROI bridge 1's) and the main-chain calls that
follow it are cut out, and the main piece is replayed from a graph of its own alone and beside sub-sets of the batch (another graph,
on the step's side stream, behind one event).  GPU time of the main piece by an event pair on the main stream.
usage: bench_handover_real.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd.model import get_model_dict
from lintransunet_amd import train, _lib
from lintransunet_amd.family_timer import FamilyTimer
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
ft = FamilyTimer()
ft.attach()
g = train.GraphedStep(model, batches[0][0], batches[0][1], w, reducer)
ft.detach()
calls = [(n, a) for n, a, _ in ft.calls]
WG = ('ltu_conv3d_wgrad', 'ltu_conv3d_pair_wgrad', 'ltu_upconv_wgrad', 'ltu_linear_wgrad_group', 'ltu_linear_wgrad', 'ltu_reduce_batch')
is_wg = [n in WG or (n == 'ltu_dwconv_bwd' and not a[4]) for n, a in calls]
# the first run of >= 12 consecutive weight-gradient calls that contains grouped projection gradients = bridge 1's batch
i = 0
start = end = None
while i < len(calls):
    if is_wg[i]:
        j = i
        while j < len(calls) and is_wg[j]:
            j += 1
        if j - i >= 12 and any(calls[k][0] == 'ltu_linear_wgrad_group' for k in range(i, j)):
            start, end = i, j
            break
        i = j
    else:
        i += 1
assert start is not None
batch = calls[start:end]
nxt = []
k = end
while k < len(calls) and len(nxt) < 60 and not is_wg[k]:
    nxt.append(calls[k]); k += 1
print(f'{len(calls)} recorded calls; batch = calls [{start}, {end}) : ' + ', '.join(f'{n[4:]} x {sum(1 for m, _ in batch if m == n)}' for n in dict.fromkeys(m for m, _ in batch)))
print(f'main piece behind it: {len(nxt)} calls: ' + ' '.join(n[4:] for n, _ in nxt[:14]) + ' ...')
orig = _lib.call


def graph_of(lst, stream):
    def issue():
        st = torch.cuda.current_stream().cuda_stream
        for name, a in lst:
            orig(name, *a[:-1], st)
    with torch.cuda.stream(stream):
        issue()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        gr.capture_begin(); issue(); gr.capture_end()
    return gr


main, side = torch.cuda.current_stream(dev), g.wq_stream
cap = torch.cuda.Stream()
gh, gm = graph_of(nxt[:1], cap), graph_of(nxt[1:], cap)          # the main piece: its first launch, the rest
fam = {'convs': [c for c in batch if c[0] in ('ltu_conv3d_wgrad', 'ltu_conv3d_pair_wgrad')],
       'unembed': [c for c in batch if c[0] == 'ltu_upconv_wgrad'],
       'group': [c for c in batch if c[0] == 'ltu_linear_wgrad_group'],
       'small': [c for c in batch if c[0] in ('ltu_reduce_batch', 'ltu_linear_wgrad', 'ltu_dwconv_bwd')]}
subsets = {'nothing': [], 'whole batch': batch}
for tag in ('convs', 'unembed', 'group', 'small', 'convs+unembed', 'unembed+convs', 'convs+group', 'group+convs', 'unembed+group',
            'group+unembed+convs+small', 'unembed+group+small+convs'):
    subsets[tag] = [c for f in tag.split('+') for c in fam[f]]
sel = os.environ.get('HANDOVER_TAGS')
print('GPU time on the main stream from the hand-over to the end of the main piece\'s FIRST launch | to its end; the side graph; us')
for tag, sub in subsets.items():
    if sel and tag not in sel.split(','):
        continue
    gs = graph_of(sub, cap) if sub else None
    th, tot, tside = 0.0, 0.0, 0.0
    for r in range(8):
        torch.cuda.synchronize()
        e0, eh, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        if gs is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                s0.record(side); gs.replay(); s1.record(side)
        gh.replay(); eh.record(main); gm.replay(); e1.record(main)
        torch.cuda.synchronize()
        if r >= 2:
            th += e0.elapsed_time(eh)
            tot += e0.elapsed_time(e1)
            tside += s0.elapsed_time(s1) if gs is not None else 0.0
    print(f'  beside {tag:28s} ({len(sub):3d} calls): first launch done {th / 6 * 1e3:8.1f}   main piece {tot / 6 * 1e3:8.1f}   side graph {tside / 6 * 1e3:8.1f}', flush=True)
reducer.zero_grad()

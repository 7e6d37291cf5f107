"""When does each gradient bucket become complete during the step?  (GPU box)  Buckets in registration order vs gradient-ready order
(GradReducer.rebucket); times are GPU timestamps (events recorded where the bucket's all-reduce would be issued) relative to the
start of the step, as a fraction of the step."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lintransunet_amd.model import get_model_dict
from lintransunet_amd import train

dev = torch.device('cuda:0')
torch.manual_seed(1234)
model = get_model_dict('MaskTransUnet')([16, 32, 64, 128, 256], [100, 65, 40, 25, 10], [False, True, True, True, True], 1, 2,
                                        dropout=0.3, act_dtype=torch.bfloat16).to(dev).train()
red = train.GradReducer(model, bucket_mb=float(os.environ.get('BUCKET_MB', '32')), unused=train.UNUSED_PARAMETERS, force_collectives=True)
weights = train.get_dynamic_weight(1)[0]
x, lab = bench.synthetic_batch(2, (128,) * 3, 100, dev)


class _Done:
    def wait(self):
        pass


def timeline(tag):
    marks = []
    red._all_reduce = lambda flat, also=None: (marks.append((len(marks), flat.numel(), torch.cuda.Event(enable_timing=True))), marks[-1][2].record(), _Done())[2]
    for _ in range(2):
        marks.clear()
        red.zero_grad()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        train.train_step(model, x, lab, weights, reducer=red)
        e1.record()
        torch.cuda.synchronize()
    total = e0.elapsed_time(e1)
    print(f'{tag}: step {total:.2f} ms; bucket complete at ' + ', '.join(f'{e0.elapsed_time(ev) / total * 100:.0f} % ({n * 4 / 1e6:.0f} MB)' for _, n, ev in marks))


timeline('registration order')
red.rebucket()
timeline('gradient-ready order')

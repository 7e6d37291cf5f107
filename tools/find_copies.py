"""Which host-side ops of one training step end in device-to-device copies or ATen kernels (torch.profiler, GPU box).
    python tools/find_copies.py"""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd import train
from lintransunet_amd.model import get_model_dict
import bench

dev = torch.device('cuda:0')
torch.manual_seed(0)
model = get_model_dict('MaskTransUnet')([16, 32, 64, 128, 256], [100, 65, 40, 25, 10], [False, True, True, True, True], 1, 2,
                                        dropout=0.3, act_dtype=torch.bfloat16).to(dev).train()
x, lab = bench.synthetic_batch(2, (128,) * 3, 5, dev)
weights = train.get_dynamic_weight(1)[0]
reducer = train.GradReducer(model, unused=train.UNUSED_PARAMETERS)
for _ in range(2):
    reducer.zero_grad()
    train.train_step(model, x, lab, weights, reducer=reducer)
torch.cuda.synchronize()
reducer.zero_grad()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    train.train_step(model, x, lab, weights, reducer=reducer)
    torch.cuda.synchronize()
ev = prof.events()
cnt = collections.Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CUDA or 'Memcpy' in e.name or 'Memset' in e.name:
        n = e.name
        if n.startswith('void ') and 'at::' not in n:
            continue
        if 'at::' in n or 'Memcpy' in n or 'Memset' in n or 'rocclr' in n:
            cnt[n[:110]] += 1
for k, v in cnt.most_common(30):
    print(v, k)
print('--- aten ops with shapes (copy_/clone/contiguous/add/zero_)')
ka = prof.key_averages(group_by_input_shape=True)
for r in ka:
    if r.key in ('aten::copy_', 'aten::clone', 'aten::contiguous', 'aten::add_', 'aten::add', 'aten::zero_', 'aten::fill_', 'aten::zeros', 'aten::to', 'aten::_to_copy', 'aten::cat', 'aten::mul', 'aten::sum'):
        print(r.count, r.key, str(r.input_shapes)[:160])

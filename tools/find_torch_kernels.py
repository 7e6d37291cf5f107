"""List the torch (non-libltu) device ops of one eager training step with their input shapes (diagnostic)."""
import collections, os, sys
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lintransunet_amd.model import get_model_dict
from lintransunet_amd import train


def main():
    dev = torch.device('cuda:0')
    torch.manual_seed(1234)
    model = get_model_dict('MaskTransUnet')([16, 32, 64, 128, 256], [100, 65, 40, 25, 10], [False, True, True, True, True],
                                            1, 2, dropout=0.3, act_dtype=torch.bfloat16).to(dev).train()
    reducer = train.GradReducer(model, bucket_mb=16.0, unused=train.UNUSED_PARAMETERS)
    weights = train.get_dynamic_weight(1)[0]
    x, lab = bench.synthetic_batch(2, (128,) * 3, 100, dev)

    def step():
        reducer.zero_grad()
        return train.train_step(model, x, lab, weights, reducer=reducer)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    cnt = collections.Counter()
    tim = collections.Counter()
    for e in prof.events():
        if e.name.startswith('aten::') and e.device_time_total > 0 and not any(c.name.startswith('aten::') for c in e.cpu_children):
            k = (e.name, str(e.input_shapes))
            cnt[k] += 1
            tim[k] += e.device_time_total
    for k, c in sorted(cnt.items(), key=lambda kv: -tim[kv[0]]):
        print(f'{c:4d} {tim[k]:9.1f} us  {k[0]} {k[1]}')


if __name__ == '__main__':
    main()

"""Rehearsal of bench.py's multi-GPU code path on ONE GPU: a 1-rank RCCL process group is alive while the step graph is
captured and replayed, and the bucket all-reduces run after each replay (what every rank does at N > 1).
run: python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 tools/check_dist_graph.py"""
import os, sys, time, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd import train
from lintransunet_amd.model import get_model_dict
from oracle import seedgen, step as O_step

torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
dist.init_process_group('nccl', device_id=dev)
model = get_model_dict('MaskTransUnet')([8, 8, 8, 16, 32], [20, 12, 9, 10, 6], [False, True, True, True, True], 1, 2,
                                        dropout=0.3, act_dtype=torch.bfloat16).to(dev).train()
train.broadcast_parameters(model)
reducer = train.GradReducer(model, unused=train.UNUSED_PARAMETERS)
reducer.world = 2           # force the collective path
reducer.avg = True          # ... with RCCL's in-collective average, as at N > 1
x = seedgen.seeded_volume((2, 1, 32, 32, 32), 1).to(dev)
lab = seedgen.seeded_label((2, 1, 32, 32, 32), 2).to(dev)
step = train.GraphedStep(model, x, lab, O_step.dynamic_weights(0), reducer)
dist.barrier()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    totals, _ = step(x, lab)
torch.cuda.synchronize()
dist.barrier()
print('ok: 5 replays + all-reduces in', round((time.perf_counter() - t0) * 1e3, 1), 'ms; loss', sum(t.item() for t in totals))
dist.destroy_process_group()

"""Sliding-window inference throughput at the reference configuration (inference_embed_attn.py: 512x512xdepth windows,
sw_batch_size 4, overlap 0.6) on a synthetic 512x512xD scan.  usage: bench_infer.py [D] [dtype]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd import infer
from lintransunet_amd.model import get_model_dict

D = int(sys.argv[1]) if len(sys.argv) > 1 else 96
act = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == 'f32') else torch.bfloat16
torch.manual_seed(0)
model = get_model_dict('MaskTransUnet')([16, 32, 64, 128, 256], [100, 65, 40, 25, 10], [False, True, True, True, True], 1, 2,
                                        act_dtype=act).cuda()
x = torch.randn(1, 1, 512, 512, D, device='cuda')
mask = (torch.rand(1, 1, 512, 512, D, device='cuda') > 0.97)
img = (512, 512, D)
nwin = len(infer.patch_starts(img, (512, 512, 32), infer.scan_interval(img, (512, 512, 32), 0.6)))
use_graph = os.environ.get('INFER_GRAPH', '1') != '0'
pred = infer.GraphedPredictor(model.eval(), 4, (512, 512, 32), x.device) if use_graph else False
out = infer.infer_volume(model, x, graph=pred)           # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    out = infer.infer_volume(model, x, graph=pred)
    vals = infer.evaluate(out, mask)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f'512x512x{D} scan, {nwin} windows of 512x512x32, {act}, {"graph replay" if use_graph else "eager"}: {dt * 1e3:.1f} ms per scan = {nwin / dt:.1f} windows/s; '
      + ', '.join(f'{k} {v.item():.4f}' for k, v in vals.items()), flush=True)

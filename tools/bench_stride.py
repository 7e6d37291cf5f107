"""Does reading one third of every 768-byte row (q out of the interleaved q|k|v buffer) cost HBM bandwidth against a contiguous read?
torch's strided copy kernel as the probe; graph-replayed."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_nt import timed
M, d = 114816, 128
nb = 6
qkvs = [torch.randn(M, 3 * d, device='cuda').bfloat16() for _ in range(nb)]
qs = [torch.randn(M, d, device='cuda').bfloat16() for _ in range(nb)]
out = torch.empty(M, d, device='cuda', dtype=torch.bfloat16)
out2 = torch.empty(M, 2 * d, device='cuda', dtype=torch.bfloat16)
cnt = [0]
def strided():
    i = cnt[0] % nb; cnt[0] += 1
    out.copy_(qkvs[i][:, :d])
def strided_kv():
    i = cnt[0] % nb; cnt[0] += 1
    out2.copy_(qkvs[i][:, d:])
def contig():
    i = cnt[0] % nb; cnt[0] += 1
    out.copy_(qs[i])
for name, fn, byts in (('contiguous [M][d] copy', contig, 2 * M * d * 2), ('q third of [M][3d]', strided, 2 * M * d * 2), ('k|v two thirds of [M][3d]', strided_kv, 4 * M * d * 2)):
    t = timed(fn)
    print(f'{name:28s} {t:6.1f} us  {byts / t * 1e-6:5.2f} TB/s', flush=True)

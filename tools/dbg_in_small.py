import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd import _lib
DEV = 'cuda'
B, S, C = 3, 77, 8
g = torch.Generator().manual_seed(43)
x = (torch.randn(B, S, C, generator=g) * 2 + 0.7).to(DEV)
res = torch.randn(B, S, C, generator=g).to(DEV)
ws = torch.empty(_lib.load().ltu_norm_ws_floats(), device=DEV)
st = torch.cuda.current_stream().cuda_stream
for trial in range(3):
    s0, s1 = torch.zeros(B, C, 3, device=DEV), torch.zeros(B, C, 3, device=DEV)
    y = torch.empty_like(x)
    _lib.call('ltu_instnorm_stats', x.data_ptr(), s0.data_ptr(), ws.data_ptr(), B, S, C, 0, st)
    _lib.call('ltu_instnorm_fwd', x.data_ptr(), s1.data_ptr(), ws.data_ptr(), res.data_ptr(), y.data_ptr(), B, S, C, 1, 0.01, 0.0, 0, 0, 0, st)
    torch.cuda.synchronize()
    d = (s0 - s1).abs()
    print(trial, d.max().item(), (d > 0).nonzero()[:8].tolist())
    xs = x - x[:, :1]
    want = torch.stack((x[:, 0], xs.double().sum(1).float(), (xs.double() ** 2).sum(1).float()), -1)
    print('  vs fp64: stats', (s0 - want).abs().max().item(), 'fwd', (s1 - want).abs().max().item())

"""Graph-replayed micro-benchmark: the row-block chain kernel (ltu_layer_tail_fwd) against the five op-by-op launches it replaces."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd import _lib, ops
from lintransunet_amd.ops import _p, _s
from bench_nt import timed
import numpy as np


def bf(*shape):
    return (torch.randn(*shape, device='cuda') * 0.5).bfloat16()


def frag(w):
    N, K = w.shape
    out = torch.empty(N * K, device='cuda', dtype=torch.bfloat16)
    rec = np.zeros(1, dtype=ops.WPREP_DTYPE)
    rec[0] = (w.data_ptr(), out.data_ptr(), 8, N, K, 0, 0, 0)
    table = torch.from_numpy(rec.view(np.uint8).copy()).cuda()
    _lib.call('ltu_weight_prep', table.data_ptr(), 1, 1, _s())
    torch.cuda.synchronize()
    return out


def run(M, d):
    a, x = bf(M, d), bf(M, d)
    wo, w1, w2 = (torch.randn(d, d, device='cuda') / d ** 0.5, torch.randn(2 * d, d, device='cuda') / d ** 0.5,
                  torch.randn(d, 2 * d, device='cuda') / (2 * d) ** 0.5)
    bo, b1, b2 = torch.zeros(d, device='cuda'), torch.zeros(2 * d, device='cuda'), torch.zeros(d, device='cuda')
    g, be = torch.ones(d, device='cuda'), torch.zeros(d, device='cuda')
    fo, f1, f2 = frag(wo), frag(w1), frag(w2)
    z1, t1, z2, y = (torch.empty(M, d, device='cuda', dtype=torch.bfloat16) for _ in range(4))
    u, h = (torch.empty(M, 2 * d, device='cuda', dtype=torch.bfloat16) for _ in range(2))
    s1, s2 = torch.empty(M, 2, device='cuda'), torch.empty(M, 2, device='cuda')
    tail = lambda: _lib.call('ltu_layer_tail_fwd', _p(a), _p(x), _p(fo), _p(f1), _p(f2), _p(bo), _p(b1), _p(b2), _p(g), _p(be), _p(g),
                             _p(be), _p(z1), _p(t1), _p(u), _p(h), _p(z2), _p(y), _p(s1), _p(s2), M, d, 1e-6, 0.3, 11, 12, 13, 0, 1, _s())
    wob, w1b, w2b = wo.bfloat16(), w1.bfloat16(), w2.bfloat16()
    o, f = torch.empty(M, d, device='cuda', dtype=torch.bfloat16), torch.empty(M, d, device='cuda', dtype=torch.bfloat16)
    pa = ops._ptr_array

    def steps():
        _lib.call('ltu_linear_fwd', _p(a), d, pa([wob]), 1, pa([bo]), _p(o), d, M, d, d, 0, 1, _s())
        _lib.call('ltu_layernorm_fwd', _p(x), _p(o), _p(g), _p(be), _p(t1), _p(s1), M, d, 1e-6, 0.3, 11, 0, 1, _s())
        _lib.call('ltu_linear_gelu_fwd', _p(t1), d, _p(w1b), _p(b1), _p(u), _p(h), M, 2 * d, d, 0.3, 12, 0, 1, _s())
        _lib.call('ltu_linear_fwd', _p(h), 2 * d, pa([w2b]), 1, pa([b2]), _p(f), d, M, d, 2 * d, 0, 1, _s())
        _lib.call('ltu_layernorm_fwd', _p(t1), _p(f), _p(g), _p(be), _p(y), _p(s2), M, d, 1e-6, 0.3, 13, 0, 1, _s())
    tt, ts = timed(tail), timed(steps)
    print(f'M={M:7d} d={d}: chain {tt:7.1f} us   five launches {ts:7.1f} us', flush=True)


for M, d in ((1024, 256), (8640, 256), (21504, 256), (28704, 128), (114816, 128)):
    run(M, d)

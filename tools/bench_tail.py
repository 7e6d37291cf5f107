"""Graph-replayed micro-benchmark: the row-block chain kernel (ltu_layer_tail_fwd) against the five op-by-op launches it replaces."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd import _lib, ops
from lintransunet_amd.ops import _p, _s
from bench_nt import timed
import numpy as np
P = float(os.environ.get("TAIL_P", "0.3"))          # dropout probability of the chain kernels (0 = no mask hashing)


def bf(*shape):
    return (torch.randn(*shape, device='cuda') * 0.5).bfloat16()


def frag(w, kind=8):
    N, K = w.shape
    out = torch.empty(N * K, device='cuda', dtype=torch.bfloat16)
    rec = np.zeros(1, dtype=ops.WPREP_DTYPE)
    rec[0] = (w.data_ptr(), out.data_ptr(), kind, N, K, 0, 0, 0)
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
                             _p(be), _p(z1), _p(t1), _p(u), _p(h), _p(z2), _p(y), _p(s1), _p(s2), M, d, 1e-6, P, 11, 12, 13, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, _s())
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
    # backward: chain vs LayerNorm bwd, dgrad, GELU bwd, dgrad, LayerNorm bwd, dgrad
    fot, f1t, f2t = frag(wo, 9), frag(w1, 9), frag(w2, 9)
    gy = bf(M, d)
    dr2, dr1, dz1, da, dz2, dt1 = (torch.empty(M, d, device='cuda', dtype=torch.bfloat16) for _ in range(6))
    du, dh = (torch.empty(M, 2 * d, device='cuda', dtype=torch.bfloat16) for _ in range(2))
    nblk = _lib.load().ltu_layer_tail_blocks(M)
    lnws = torch.empty(2, nblk, 2 * d, device='cuda')
    dg, db = torch.zeros(d, device='cuda'), torch.zeros(d, device='cuda')
    ws = torch.empty(_lib.load().ltu_norm_ws_floats(), device='cuda')
    btail = lambda: _lib.call('ltu_layer_tail_bwd', _p(gy), 0, _p(z2), _p(z1), _p(u), _p(s2), _p(s1), _p(g), _p(g), _p(f2t), _p(f1t), _p(fot),
                              _p(dr2), _p(du), _p(dr1), _p(dz1), _p(da), _p(lnws[0]), _p(lnws[1]), lnws[0].numel(), M, d, P, 11, 12, 13, 0, 1, 1, _s())
    wot, w1t, w2t = wo.t().contiguous().bfloat16(), w1.t().contiguous().bfloat16(), w2.t().contiguous().bfloat16()

    def bsteps():
        _lib.call('ltu_layernorm_bwd', _p(gy), 0, _p(z2), _p(s2), _p(g), _p(dz2), _p(dr2), _p(dg), _p(db), _p(ws), ws.numel(), 0, M, d, 0.3, 13, 0, 1, _s())
        _lib.call('ltu_linear_fwd', _p(dr2), d, pa([w2t]), 1, pa([None]), _p(dh), 2 * d, M, 2 * d, d, 0, 1, _s())
        _lib.call('ltu_gelu_dropout_bwd', _p(dh), _p(u), _p(du), M * 2 * d, 0.3, 12, 0, 1, _s())
        _lib.call('ltu_linear_fwd', _p(du), 2 * d, pa([w1t]), 1, pa([None]), _p(dt1), d, M, d, 2 * d, 0, 1, _s())
        _lib.call('ltu_layernorm_bwd', _p(dt1), _p(dz2), _p(z1), _p(s1), _p(g), _p(dz1), _p(dr1), _p(dg), _p(db), _p(ws), ws.numel(), 0, M, d, 0.3, 11, 0, 1, _s())
        _lib.call('ltu_linear_fwd', _p(dr1), d, pa([wot]), 1, pa([None]), _p(da), d, M, d, d, 0, 1, _s())
    bt, bs = timed(btail), timed(bsteps)
    print(f'M={M:7d} d={d}: forward chain {tt:7.1f} us vs five launches {ts:7.1f} us;  backward chain {bt:7.1f} us vs six launches {bs:7.1f} us', flush=True)


import sys
SHAPES = ((1024, 256), (8640, 256), (21504, 256), (28704, 128), (114816, 128))
if len(sys.argv) > 1 and sys.argv[1] == 'scale':
    SHAPES = tuple((m, d) for d in (128, 256) for m in (64, 32 * 256, 32 * 512, 32 * 1024, 32 * 2048))
for M, d in SHAPES:
    run(M, d)

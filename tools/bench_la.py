"""Graph-replayed micro-benchmark of the linear-attention core through the C-ABI (forward and backward separately), with a sweep of
the geometry knobs.  usage: bench_la.py [knob=value ...]   e.g.  bench_la.py LTU_LA_TOKB_BLOCKS=1024"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd import _lib
from lintransunet_amd.ops import _p, _s
from bench_nt import timed

SHAPES = ((2, 57408, 128), (2, 10752, 256), (2, 4320, 256), (2, 512, 256))


def run(B, N, d):
    H = d // 32
    nb = 3
    qkvs = [(torch.randn(B * N, 3 * d, device='cuda') * 0.5).bfloat16() for _ in range(nb)]
    gos = [(torch.randn(B * N, d, device='cuda') * 0.5).bfloat16() for _ in range(nb)]
    nsplit = _lib.load().ltu_linattn_splits(B, N)
    out = torch.empty(B * N, d, device='cuda', dtype=torch.bfloat16)
    cx = torch.empty(B * H, 32, 32, device='cuda')
    cs = torch.empty(B * H, 64, device='cuda')
    qstat = torch.empty(B * N, H, 2, device='cuda')
    ws = torch.empty(_lib.load().ltu_linattn_ws_floats(B, N, d), device='cuda')
    dqkv = torch.empty_like(qkvs[0])
    dctx = torch.empty_like(cx)
    tvec = torch.empty(B * H, 32, device='cuda')
    cnt = [0]

    def fwd():
        i = cnt[0] % nb; cnt[0] += 1
        _lib.call('ltu_linattn_fwd', _p(qkvs[i]), _p(out), _p(cx), _p(cs), _p(qstat), _p(ws), ws.numel(), B, N, d, 1, _s())

    def bwd():
        i = cnt[0] % nb; cnt[0] += 1
        _lib.call('ltu_linattn_bwd', _p(qkvs[i]), _p(gos[i]), _p(cx), _p(cs), _p(qstat), _p(dqkv), _p(dctx), _p(tvec), _p(ws), ws.numel(), B, N, d, 1, _s())
    tf = timed(fwd)
    tb = timed(bwd)
    fb, bb = 4 * B * N * d * 2, 8 * B * N * d * 2
    print(f'B={B} N={N:6d} d={d}: fwd {tf:6.1f} us ({fb / tf * 1e-6:5.2f} TB/s)  bwd {tb:6.1f} us ({bb / tb * 1e-6:5.2f} TB/s)  splits {nsplit}', flush=True)


for kv in sys.argv[1:]:
    k, v = kv.split('=')
    _lib.config_set(k, int(v))
print(' '.join(sys.argv[1:]) or 'defaults')
for s in SHAPES:
    run(*s)

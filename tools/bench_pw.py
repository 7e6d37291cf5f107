"""Graph-replayed micro-benchmark of the streaming kernels (LayerNorm, GELU, InstanceNorm) at the bench's sizes, bf16."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd import _lib
from lintransunet_amd.ops import _p, _s
from bench_nt import timed

PDROP = float(os.environ.get("PW_P", "0.3"))        # dropout probability of the InstanceNorm passes (0 = no mask hashing)
WS = torch.empty(1 << 20, device='cuda') if os.environ.get('NO_WS') is None else None

def bf(*shape):
    return torch.randn(*shape, device='cuda').bfloat16()

def ln(M, d):
    x, r, y, g = bf(M, d), bf(M, d), bf(M, d), bf(M, d)
    dz, dr = bf(M, d), bf(M, d)
    gamma, beta = torch.ones(d, device='cuda'), torch.zeros(d, device='cuda')
    dg, db = torch.zeros(d, device='cuda'), torch.zeros(d, device='cuda')
    stat = torch.empty(M, 2, device='cuda')
    f = lambda: _lib.call('ltu_layernorm_fwd', _p(x), _p(r), _p(gamma), _p(beta), _p(y), _p(stat), M, d, 1e-5, 0.3, 1, 0, 1, _s())
    b = lambda: _lib.call('ltu_layernorm_bwd', _p(g), 0, _p(r), _p(stat), _p(gamma), _p(dz), _p(dr), _p(dg), _p(db), _p(WS), (WS.numel() if WS is not None else 0), 0, M, d, 0.3, 1, 0, 1, _s())
    tf, tb = timed(f), timed(b)
    mb = M * d * 2 / 1e6
    print(f'LN   M={M:7d} d={d:4d}: fwd {tf:6.1f} us ({4 * mb / tf:.1f} TB/s)  bwd {tb:6.1f} us ({4 * mb / tb:.1f} TB/s)', flush=True)

def gelu(n):
    u, h, dh, du = bf(n), bf(n), bf(n), bf(n)
    f = lambda: _lib.call('ltu_gelu_dropout_fwd', _p(u), _p(h), n, 0.3, 1, 0, 1, _s())
    b = lambda: _lib.call('ltu_gelu_dropout_bwd', _p(dh), _p(u), _p(du), n, 0.3, 1, 0, 1, _s())
    tf, tb = timed(f), timed(b)
    mb = n * 2 / 1e6
    print(f'GELU n={n:10d}: fwd {tf:6.1f} us ({2 * mb / tf:.1f} TB/s)  bwd {tb:6.1f} us ({3 * mb / tb:.1f} TB/s)', flush=True)

def inorm(B, S, C):
    x, y, dy, dx = bf(B, S, C), bf(B, S, C), bf(B, S, C), bf(B, S, C)
    sums = torch.zeros(B, C, 3, device='cuda'); bs = torch.zeros(B, C, 2, device='cuda')
    st = lambda: _lib.call('ltu_instnorm_stats', _p(x), _p(sums), _p(WS), (WS.numel() if WS is not None else 0), B, S, C, 1, _s())
    ap = lambda: _lib.call('ltu_instnorm_apply', _p(x), _p(sums), 0, _p(y), B, S, C, 1, 0.01, PDROP, 1, 0, 1, _s())
    bw = lambda: _lib.call('ltu_instnorm_bwd', _p(dy), 0, 0, _p(x), _p(sums), _p(bs), _p(WS), (WS.numel() if WS is not None else 0), _p(dx), B, S, C, 1, 0.01, PDROP, 1, 0, 1, _s())
    fw = lambda: _lib.call('ltu_instnorm_fwd', _p(x), _p(sums), _p(WS), (WS.numel() if WS is not None else 0), 0, _p(y), B, S, C, 1, 0.01, PDROP, 1, 0, 1, _s())
    t1, t2, t3, t4 = timed(st), timed(ap), timed(bw), timed(fw)
    mb = B * S * C * 2 / 1e6
    print(f'IN   B={B} S={S:8d} C={C:4d} ({mb:.0f} MB): stats {t1:6.1f} us ({mb / t1:.1f} TB/s)  apply {t2:6.1f} us ({2 * mb / t2:.1f} TB/s)  '
          f'fwd(stats+apply, one call) {t4:6.1f} us ({3 * mb / t4:.1f} TB/s)  bwd(stats+apply) {t3:6.1f} us ({5 * mb / t3:.1f} TB/s)', flush=True)

def gate(B, S, C):
    """attention gate (Unet_3Dblock.py:217-221): forward | backward (reduce + fold + apply)"""
    u1, u2, skip, out, dout, dskip, du1, du2 = (bf(B, S, C) for _ in range(8))
    s1, s2 = torch.zeros(B, C, 3, device='cuda'), torch.zeros(B, C, 3, device='cuda')
    s1[..., 2] = S; s2[..., 2] = S
    pw, pb = torch.randn(C, device='cuda'), torch.zeros(1, device='cuda')
    a, ds = torch.empty(B * S, device='cuda'), torch.empty(B * S, device='cuda')
    dpw, dpb = torch.zeros(C, device='cuda'), torch.zeros(1, device='cuda')
    b1, b2 = torch.zeros(B, C, 2, device='cuda'), torch.zeros(B, C, 2, device='cuda')
    f = lambda: _lib.call('ltu_gate_fwd', _p(u1), _p(u2), _p(s1), _p(s2), _p(pw), _p(pb), _p(skip), _p(a), _p(out), B, S, C, 1, _s())
    bw = lambda: _lib.call('ltu_gate_bwd', _p(dout), _p(u1), _p(u2), _p(s1), _p(s2), _p(pw), _p(skip), _p(a), _p(dskip), _p(ds), _p(dpw),
                           _p(dpb), _p(b1), _p(b2), _p(WS), (WS.numel() if WS is not None else 0), _p(du1), _p(du2), B, S, C, 1, _s())
    tf, tb = timed(f), timed(bw)
    mb = B * S * C * 2 / 1e6
    print(f'gate B={B} S={S:8d} C={C:4d} ({mb:.0f} MB): fwd {tf:6.1f} us ({4 * mb / tf:.1f} TB/s)  bwd {tb:6.1f} us ({9 * mb / tb:.1f} TB/s)', flush=True)


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if which in ('all', 'ln'):
        for M, d in [(114816, 128), (21504, 256), (8640, 256), (1024, 256)]:
            ln(M, d)
    if which in ('all', 'gelu'):
        for n in [114816 * 256, 21504 * 512, 8640 * 512]:
            gelu(n)
    if which in ('all', 'in'):
        for B, S, C in [(2, 524288, 16), (2, 131072, 32), (2, 16384, 64), (2, 2048, 128)]:
            inorm(B, S, C)
    if which in ('all', 'gate'):
        for B, S, C in [(2, 524288, 16), (2, 131072, 32), (2, 16384, 64), (2, 2048, 128)]:
            gate(B, S, C)

"""Micro-benchmark of the bf16 dense projection kernels (GPU box).  usage: bench_gemm.py [M K N]..."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd import ops

def run(M, K, N, nw=1):
    x = torch.randn(M, K, device='cuda').bfloat16().requires_grad_(True)
    ws = [torch.randn(N // nw, K, device='cuda').mul_(0.05).requires_grad_(True) for _ in range(nw)]
    bs = [torch.zeros(N // nw, device='cuda').requires_grad_(True) for _ in range(nw)]
    go = torch.randn(M, N, device='cuda').bfloat16()
    res = {}
    for name in ('fwd', 'fwd+bwd'):
        def it():
            y = ops.linear(x, ws, bs)
            if name != 'fwd':
                y.backward(go)
        for _ in range(5):
            it()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            it()
        e1.record(); e1.synchronize()
        res[name] = e0.elapsed_time(e1) / 20 * 1e3
    byts = (M * K + M * N) * 2
    print(f'M={M} K={K} N={N}: fwd(incl. weight cast) {res["fwd"]:.1f} us  ({byts / res["fwd"] / 1e6:.2f} TB/s, {2 * M * K * N / res["fwd"] / 1e6:.0f} TF)  fwd+bwd {res["fwd+bwd"]:.1f} us')

shapes = [(114816, 128, 256), (114816, 128, 384), (114816, 256, 128), (114816, 128, 128), (21504, 256, 512), (21504, 512, 256), (1024, 256, 512)]
print('variant', os.environ.get('LTU_NT_VARIANT', '0'))
for s in shapes:
    run(*s)

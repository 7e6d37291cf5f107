"""Micro-benchmark of the linear-attention core (run under rocprofv3 --kernel-trace --stats on the GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd import ops
B, N, d = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dt = torch.bfloat16 if len(sys.argv) < 5 or sys.argv[4] == 'bf16' else torch.float32
qkv = torch.randn(B * N, 3 * d, device='cuda').to(dt).requires_grad_(True)
go = torch.randn(B * N, d, device='cuda').to(dt)
for _ in range(10):
    out = ops.linear_attention(qkv, B, N, d)
    out.backward(go)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    out = ops.linear_attention(qkv, B, N, d)
    out.backward(go)
e1.record(); e1.synchronize()
print(f'B={B} N={N} d={d} {dt}: fwd+bwd {e0.elapsed_time(e1) / 20 * 1e3:.1f} us')

"""hipExtStreamCreateWithCUMask: does a CU-masked stream confine a kernel, and how does the mask map onto the 256 CUs?
Times a streaming copy (HBM-bound) and a spin kernel on masked streams; then a small kernel on the default stream while a long
kernel occupies the masked stream (the contention case of the weight-gradient side stream)."""
import ctypes, os, sys, time
import torch
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so'))
dev = torch.device('cuda', 0)
torch.cuda.init(); torch.zeros(1, device=dev)

def masked(words):
    st = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), len(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)

a = torch.randn(64 << 20, device=dev)      # 256 MB
b = torch.empty_like(a)
def t_copy(s):
    with torch.cuda.stream(s):
        b.copy_(a); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5): b.copy_(a)
        e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) / 5
print('default stream copy 256 MB: %.3f ms' % t_copy(torch.cuda.current_stream()))
for name, words in [('all 256', [0xFFFFFFFF] * 8), ('low half of every word', [0x0000FFFF] * 8), ('first 4 words', [0xFFFFFFFF] * 4 + [0] * 4),
                    ('quarter of every word', [0x000000FF] * 8), ('first 2 words', [0xFFFFFFFF] * 2 + [0] * 6), ('even bits', [0x55555555] * 8)]:
    s = masked(words)
    print(f'{name:26s}: copy {t_copy(s):.3f} ms')
# contention: long copy on the masked stream, small kernels on the default stream
x = torch.randn(1 << 16, device=dev)
def small_latency(s_side):
    torch.cuda.synchronize()
    with torch.cuda.stream(s_side):
        for _ in range(20): b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): x.mul_(1.0001)
    e1.record(); e1.synchronize()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 200 * 1e3
print('200 small kernels alone: %.2f us each' % small_latency(torch.cuda.current_stream()))
print('... beside a copy on an unmasked side stream: %.2f us each' % small_latency(torch.cuda.Stream()))
for name, words in [('low half of every word', [0x0000FFFF] * 8), ('quarter of every word', [0x000000FF] * 8), ('first 4 words', [0xFFFFFFFF] * 4 + [0] * 4)]:
    print(f'... beside a copy on a side stream masked to {name}: %.2f us each' % small_latency(masked(words)))

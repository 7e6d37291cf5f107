"""Which torch streams really run beside the current stream?  HIP maps its streams onto a few hardware queues (GPU_MAX_HW_QUEUES,
default 4); two streams on one queue serialise.  For each of N streams from torch's pool: time a spin kernel on the current stream
plus the same kernel on the candidate (1x = concurrent, 2x = one queue)."""
import sys, time
import torch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
pre = int(sys.argv[2]) if len(sys.argv) > 2 else 0       # streams created (and used) before the candidates, like a communicator's
main = torch.cuda.current_stream()
keep = []
for _ in range(pre):
    s = torch.cuda.Stream(); keep.append(s)
    with torch.cuda.stream(s):
        torch.cuda._sleep(1000)
torch.cuda._sleep(100000)
torch.cuda.synchronize()
cyc = 3_000_000


def both(b):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    torch.cuda._sleep(cyc)
    if b is not None:
        with torch.cuda.stream(b):
            torch.cuda._sleep(cyc)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
base = min(both(None) for _ in range(3))
out = []
for i in range(N):
    s = torch.cuda.Stream()
    keep.append(s)
    t = min(both(s) for _ in range(2))
    out.append(round(t / base, 2))
print('one kernel', round(base, 3), 'ms; ratio per candidate stream:', out)

"""Is a captured hipMemsetAsync node ordered like a kernel node when a HIP graph is replayed?  (Round 2 moved ltu_roi_plan's
histogram clear out of the library after replays >= 2 saw stale histograms; ADVICE asked for the cause.)
Graph: memset(buf, 0) -> buf += 1 -> acc += buf, all on the capturing stream.  After R replays acc must be exactly R everywhere and
buf exactly 1; a memset that is not ordered before the increment shows up as buf == 2, 3, ... or acc != R."""
import ctypes, os, sys, torch
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so'))
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
dev = torch.device('cuda', 0)
for n in (1 << 10, 1 << 16, 1 << 22):
    for pre in (0, 8):                   # kernels in front of the memset inside the graph (the step graph has hundreds)
        buf = torch.zeros(n, device=dev)
        acc = torch.zeros(n, device=dev)
        junk = torch.zeros(1 << 20, device=dev)

        def body():
            for _ in range(pre):
                junk.mul_(1.0001)
            rc = hip.hipMemsetAsync(buf.data_ptr(), 0, n * 4, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
            buf.add_(1.0)
            acc.add_(buf)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            body()
        torch.cuda.synchronize()
        acc.zero_()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode='thread_local'):
            body()
        R = 2000
        for _ in range(R):
            g.replay()
        torch.cuda.synchronize()
        bad_buf = int((buf != 1).sum().item())
        bad_acc = int((acc != R).sum().item())
        print(f'n = {n:8d}, {pre} kernels in front: buf != 1 at {bad_buf} elements (max {buf.max().item():.0f}), '
              f'acc != {R} at {bad_acc} elements (range {acc.min().item():.0f} .. {acc.max().item():.0f})', flush=True)

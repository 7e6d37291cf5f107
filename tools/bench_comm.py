"""Per-collective cost of the direct RCCL path on ONE GPU (1-rank communicator): back-to-back all-reduces of one bucket size on
the communicator's stream, eagerly and replayed from a captured graph, with HIP events on that stream.
    python tools/bench_comm.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd import comm as C

dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
comm = C.RcclComm(dev)
for mb in (0.25, 1, 4, 16, 64):
    n = int(mb * 1024 * 1024 / 4)
    bufs = [torch.randn(n, device=dev) for _ in range(8)]
    for b in bufs:
        comm.allreduce_avg(b).wait()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        for b in bufs:
            comm.allreduce_avg(b).wait()
    e1.record()
    torch.cuda.synchronize()
    eager = e0.elapsed_time(e1) / 32 * 1e3
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        for b in bufs:
            comm.allreduce_avg(b).wait()
    g.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(4):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f'{mb:6.2f} MB bucket: eager {eager:7.1f} us per all-reduce, graph replay {e0.elapsed_time(e1) / 32 * 1e3:7.1f} us', flush=True)
comm.close()

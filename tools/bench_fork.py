"""What does a fork cost inside a replayed HIP graph?  Main chain of N dependent kernels on the capturing stream; K side branches
(one kernel each on a second stream, forked after main kernel p_k, all joined at the end).  Reports replay time against the
unforked chain (+ the side kernels appended to the chain) for several K, side-kernel sizes and chain-kernel sizes."""
import sys, torch
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)


def build(n_main, main_elems, forks, side_elems, same_side_stream=True, join='end'):
    x = torch.ones(main_elems, device=dev)
    sides = [torch.ones(side_elems, device=dev) for _ in forks]
    s_main = torch.cuda.Stream()
    s_side = [torch.cuda.Stream() for _ in range(1 if same_side_stream else max(1, len(forks)))]

    def body():
        fi = 0
        for i in range(n_main):
            x.mul_(1.0001)
            while fi < len(forks) and forks[fi] == i:
                ss = s_side[0 if same_side_stream else fi]
                ss.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(ss):
                    sides[fi].mul_(1.0001)
                if join == 'next':
                    pass
                fi += 1
        for ss in s_side:
            torch.cuda.current_stream().wait_stream(ss)
    with torch.cuda.stream(s_main):
        body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        body()
    return g


def timed(g, reps=20):
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


N = 400
for main_elems in (1 << 16, 1 << 22):
    base = timed(build(N, main_elems, [], 1))
    print(f'main chain {N} kernels x {main_elems} elements: {base:.0f} us ({base / N:.2f} us per kernel)')
    for side_elems in (1 << 12, 1 << 22, 1 << 25):
        for K in (1, 2, 3, 4, 8, 16, 64):
            forks = [int((k + 0.5) * N / K) for k in range(K)]
            t1 = timed(build(N, main_elems, forks, side_elems, True))
            t2 = timed(build(N, main_elems, forks, side_elems, False))
            print(f'   side {side_elems:9d} elems, K = {K:3d} forks: one side stream {t1 - base:+8.0f} us ({(t1 - base) / K:+6.1f} per fork), '
                  f'K side streams {t2 - base:+8.0f} us ({(t2 - base) / K:+6.1f} per fork)', flush=True)

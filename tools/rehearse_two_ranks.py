"""Two data-parallel ranks end to end on ONE GPU (RCCL refuses two ranks on one device, tools/try_two_ranks_one_gpu.py, so the
gradient exchange of this rehearsal is a TEST communicator that stages the buckets through the host and gloo; everything else is the
product path: rendezvous, parameter broadcast, per-rank patches, HIP kernels, gradient hooks in ready order, re-bucketing,
GraphedStep with the collectives after the replay and as linear segments with the collectives between them, accumulation).  Checks the property the data-parallel design rests on
(SURVEY 8e): every op of the network is per-sample, so mean over ranks of the per-rank gradients == gradient of the full batch.

    python tools/rehearse_two_ranks.py          (parent: spawns rank 0 / 1 on device 0, then compares with a 1-process full-batch run)
    python tools/rehearse_two_ranks.py --rccl   (needs two GPUs: rank r on device r, the exchange through the PRODUCT communicator -
                                                 RcclComm over a gloo control plane - plus a known-answer all-reduce: ranks
                                                 contribute 1 and 3, every rank must read 2; tests/test_gpu_comm.py runs it
                                                 wherever two devices are visible)
"""
import os, socket, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'gpurun_out')
SMALL = dict(num_layers=[8, 8, 8, 16, 32], roi_size_list=[20, 12, 9, 10, 6])


def setup(batch, seed0):
    import torch
    from lintransunet_amd.model import get_model_dict
    from oracle import net as O_net, seedgen
    cfg = O_net.NetConfig(**SMALL)
    m = get_model_dict('MaskTransUnet')(cfg.num_layers, cfg.roi_size_list, cfg.is_roi_list, 1, 2, dropout=0.0, act_dtype=torch.bfloat16)
    m.load_state_dict(seedgen.seeded_params(O_net.param_shapes(cfg), seed0), strict=True)
    return m.to('cuda').train()


def patches(indices):
    import torch
    from oracle import seedgen
    xs = [seedgen.seeded_volume((1, 1, 32, 32, 32), 500 + i) for i in indices]
    ls = [seedgen.seeded_label((1, 1, 32, 32, 32), 600 + i) for i in indices]
    return torch.cat(xs).cuda(), torch.cat(ls).cuda()


def whole(m):
    import torch
    return torch.cat([p.grad.flatten().float() for p in m.parameters() if p.grad is not None]).cpu()


def rank_main():
    import torch, torch.distributed as dist
    from lintransunet_amd import train, comm as C
    from oracle import step as O_step
    rank = int(os.environ['RANK'])
    rccl = os.environ.get('LTU_REHEARSE_RCCL') == '1'
    torch.cuda.set_device(rank if rccl else 0)
    dist.init_process_group('gloo')
    gloo = C.GlooComm()

    if rccl:
        comm = C.RcclComm(torch.device('cuda', rank), control=gloo)       # the product path: direct RCCL calls behind the C-ABI
        # known answer first: a wrong reduction op, dtype enum or count would show here, not as "gradients look plausible"
        for n in (1, 1000, (1 << 22) + 3):
            t = torch.full((n,), 1.0 + 2.0 * rank, device='cuda')
            comm.allreduce_avg(t).wait()
            torch.cuda.synchronize()
            assert torch.equal(t, torch.full_like(t, 2.0)), (n, t[:4].tolist())
        b = torch.full((1000,), float(rank + 5), device='cuda')
        comm.broadcast(b, 0)
        torch.cuda.synchronize()
        assert torch.equal(b, torch.full_like(b, 5.0))
    else:
        comm = C.HostStagedComm(gloo)       # test-only: GPU bucket -> host -> gloo -> GPU (the product path uses RcclComm)
    m = setup(2, 100 + rank)                # different initial weights per rank: the broadcast must fix that
    train.broadcast_parameters(m, comm)
    red = train.GradReducer(m, bucket_mb=0.25, unused=train.UNUSED_PARAMETERS, comm=comm)
    assert red.world == 2
    x, lab = patches([2 * rank, 2 * rank + 1])          # this rank's two patches of the global batch of four
    w = O_step.dynamic_weights(0)
    red.zero_grad()
    train.train_step(m, x, lab, w, reducer=red)         # eager, collectives from the gradient hooks
    torch.cuda.synchronize()
    g_hooks = whole(m)
    names = {id(p): n for n, p in m.named_parameters()}
    per_param = {names[id(p)]: (bi, p.grad.detach().float().cpu().clone()) for bi, b in enumerate(red.buckets) for p in b}
    red.rebucket()
    step = train.GraphedStep(m, x, lab, w, red, overlap='after')       # replay + collectives after it
    step(x, lab)
    torch.cuda.synchronize()
    g_graph = whole(m)
    seg = train.GraphedStep(m, x, lab, w, red, overlap='segments')     # linear segments, collectives between them (the default)
    assert len(seg.graphs[(True, True)][0]) >= 3
    for _ in range(2):
        seg(x, lab)
    torch.cuda.synchronize()
    g_seg = whole(m)
    # accumulation: two micro-steps of one patch each (losses / 2), reduced on the last only
    step2 = train.GraphedStep(m, x[:1], lab[:1], w, red, step_times=2, overlap='segments')
    for j in range(2):
        step2(x[j:j + 1], lab[j:j + 1], micro=j)
    torch.cuda.synchronize()
    g_acc = whole(m)
    torch.save({'hooks': g_hooks, 'graph': g_graph, 'seg': g_seg, 'acc': g_acc, 'per_param': per_param, 'w0': next(m.parameters()).detach().float().cpu()},
               os.path.join(OUT, f'two_ranks_{rank}.pt'))
    tmax = comm.max_float(float(rank))
    assert tmax == 1.0
    comm.barrier()
    comm.close()
    dist.destroy_process_group()


def parent():
    import torch
    os.makedirs(OUT, exist_ok=True)
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    rccl = '--rccl' in sys.argv
    if rccl and torch.cuda.device_count() < 2:
        print('--rccl needs two visible GPUs')
        sys.exit(2)
    env = dict(os.environ, WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), LTU_REHEARSE_RCCL='1' if rccl else '0')
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    procs = [subprocess.Popen([sys.executable, __file__, 'rank'], env=dict(env, RANK=str(r))) for r in range(2)]
    rcs = [p.wait(timeout=600) for p in procs]
    assert rcs == [0, 0], rcs
    from lintransunet_amd import train
    from oracle import step as O_step
    r0, r1 = (torch.load(os.path.join(OUT, f'two_ranks_{r}.pt')) for r in range(2))
    assert torch.equal(r0['w0'], r1['w0'])                                  # broadcast made the ranks identical
    for k in ('hooks', 'graph', 'seg', 'acc'):
        if not torch.equal(r0[k], r1[k]):
            print(f'{k}: ranks differ, rel-L2 {((r0[k] - r1[k]).norm() / r0[k].norm()).item():.2e}')
            if k == 'hooks':
                for n, (bi, g0) in r0['per_param'].items():
                    g1 = r1['per_param'][n][1]
                    if not torch.equal(g0, g1):
                        print(f'   bucket {bi} {n}: rel diff {((g0 - g1).norm() / g0.norm().clamp_min(1e-30)).item():.2e}')
    for k in ('hooks', 'graph', 'seg', 'acc'):
        assert torch.equal(r0[k], r1[k]), k                                  # an all-reduced gradient is the same on every rank
    m = setup(4, 100)                                                         # rank 0's weights
    red = train.GradReducer(m, bucket_mb=0.25, unused=train.UNUSED_PARAMETERS)
    x, lab = patches([0, 1, 2, 3])
    red.zero_grad()
    train.train_step(m, x, lab, O_step.dynamic_weights(0), reducer=red)
    torch.cuda.synchronize()
    ref = whole(m)
    for k in ('hooks', 'graph', 'seg', 'acc'):
        err = ((r0[k] - ref).norm() / ref.norm()).item()
        print(f'mean over 2 ranks ({k:5s}) vs full batch of 4 on one process: whole-gradient rel-L2 {err:.2e}')
        if err > 1e-4:                      # which tensors carry the difference
            off, rows = 0, []
            for n, p in m.named_parameters():
                if p.grad is None:
                    continue
                a, b = r0[k][off:off + p.numel()], ref[off:off + p.numel()]
                off += p.numel()
                rows.append((((a - b).norm() / b.norm().clamp_min(1e-12)).item(), (a - b).norm().item(), n))
            rows.sort(reverse=True)
            for e, ab, n in rows[:12]:
                print(f'      {n}: rel {e:.2e} abs {ab:.2e}')
            rows.sort(key=lambda r: -r[1])
            print('      largest absolute:', [(n, f'{ab:.2e}', f'{e:.2e}') for e, ab, n in rows[:6]])
        assert err <= 2e-2, (k, err)           # bf16 storage: batch 2 vs batch 4 launches differ in reduction splits only
    print('ok')


if __name__ == '__main__':
    rank_main() if len(sys.argv) > 1 and sys.argv[1] == 'rank' else parent()

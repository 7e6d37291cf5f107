"""Can two RCCL ranks share ONE GPU on this box?  (NCCL refuses duplicate devices; RCCL builds differ.)  Spawns two processes on
device 0 with a gloo control plane and tries a 2-rank communicator + one all-reduce.  Exit code 0 = it worked."""
import os, sys, socket, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if 'RANK' not in os.environ:
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    procs = [subprocess.Popen([sys.executable, __file__], env=dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                                                                  MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0'))
             for r in range(2)]
    rcs = []
    for p in procs:
        try:
            rcs.append(p.wait(timeout=120))
        except subprocess.TimeoutExpired:
            p.kill(); rcs.append(-9)
    print('exit codes', rcs)
    sys.exit(0 if rcs == [0, 0] else 1)
import torch, torch.distributed as dist
from lintransunet_amd import comm as C
rank = int(os.environ['RANK'])
torch.cuda.set_device(0)
dist.init_process_group('gloo')
try:
    comm = C.RcclComm(torch.device('cuda', 0), control=C.GlooComm())
    x = torch.full((1 << 20,), float(rank + 1), device='cuda')
    comm.allreduce_avg(x).wait()
    torch.cuda.synchronize()
    print(f'rank {rank}: all-reduce mean = {x[0].item()} (expected 1.5)', flush=True)
    ok = abs(x[0].item() - 1.5) < 1e-6
    comm.close()
except Exception as e:
    print(f'rank {rank}: FAILED {type(e).__name__}: {e}', flush=True)
    ok = False
dist.destroy_process_group()
sys.exit(0 if ok else 1)

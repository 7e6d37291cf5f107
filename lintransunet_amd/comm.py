"""Communicators of the data-parallel step (one process per GPU; replaces nn.DataParallel, train3D.py:119).

`train.GradReducer` talks to one of these, never to torch.distributed directly:

    comm.world, comm.rank
    comm.allreduce_avg(flat) -> handle      flat (a gradient bucket, fp32) <- mean over ranks, asynchronously; handle.wait()
                                             makes the caller's current stream (GPU) / the caller (CPU) wait for it
    comm.broadcast(tensor, src)             initial parameter sync
    comm.barrier(), comm.max_float(x)       host-side control (bench.py's bracket and max-over-ranks timing)

* `RcclComm`  - the GPU data path: direct RCCL calls behind the C-ABI (csrc/comm.hip: ltu_comm_init / ltu_comm_allreduce_avg).  The
  collective is enqueued on a communication stream of this object; fork (comm stream waits for the compute stream) and join
  (handle.wait) are plain stream dependencies, so inside a HIP-graph capture a bucket's all-reduce becomes a side branch of the
  step graph.  There is no ProcessGroupNCCL and therefore no watchdog thread that could poll an event of a capturing stream.
  The 128-byte unique id travels over the host control group.
* `GlooComm`  - the same interface over a torch.distributed gloo group on CPU tensors: the host control plane of `RcclComm`, and
  the CPU stand-in of the two entry points in the world-size-2 tests (tests/test_host_cpu.py).
* `LocalComm` - world size 1: nothing to exchange.
"""
import contextlib
import ctypes
import os
import sys

import torch
import torch.distributed as dist

from . import _lib


class _Done:
    def wait(self):
        pass


class LocalComm:
    world, rank = 1, 0

    def allreduce_avg(self, flat, also=None):
        return _Done()

    def broadcast(self, t, src=0):
        pass

    def barrier(self):
        pass

    def max_float(self, x):
        return float(x)

    def close(self):
        pass


class _GlooHandle:
    def __init__(self, work, flat, world):
        self.work, self.flat, self.world = work, flat, world

    def wait(self):
        self.work.wait()
        self.flat.div_(self.world)          # gloo has no averaging reduction


class GlooComm:
    """torch.distributed (gloo) on CPU tensors.  `group=None` = the default process group, which must be gloo."""

    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError('GlooComm needs torch.distributed.init_process_group("gloo") first')
        if dist.get_backend(group) != 'gloo':
            raise RuntimeError('GlooComm is the CPU control / test communicator: the group must use the gloo backend')
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)

    def allreduce_avg(self, flat, also=None):
        if flat.is_cuda:
            raise _lib.LtuError('GlooComm reduces CPU tensors only; GPU gradients go through RcclComm (no silent staging copies)')
        return _GlooHandle(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True), flat, self.world)

    def broadcast(self, t, src=0):
        dist.broadcast(t, src, group=self.group)

    def barrier(self):
        dist.barrier(group=self.group)

    def max_float(self, x):
        t = torch.tensor([float(x)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return t.item()

    def close(self):
        pass


class HostStagedComm:
    """TEST communicator: a GPU bucket travels host -> gloo -> host -> GPU.  It exists for rehearsals of the multi-rank code paths
    with several processes on ONE GPU, where RCCL refuses a second rank per device (tools/rehearse_two_ranks.py,
    `bench.py --test-comm staged`); it synchronises the stream and is never selected by the product path."""

    def __init__(self, control):
        self.control = control
        self.world, self.rank = control.world, control.rank
        self.calls = 0

    def allreduce_avg(self, flat, also=None):
        torch.cuda.current_stream().synchronize()
        if also is not None:
            also.synchronize()
        h = flat.cpu()
        self.control.allreduce_avg(h).wait()
        flat.copy_(h)
        self.calls += 1
        return _Done()

    def broadcast(self, t, src=0):
        h = t.cpu()
        self.control.broadcast(h, src)
        t.copy_(h)

    def barrier(self):
        self.control.barrier()

    def max_float(self, x):
        return self.control.max_float(x)

    def close(self):
        pass


class _StreamHandle:
    def __init__(self, stream):
        self.stream = stream

    def wait(self):
        torch.cuda.current_stream().wait_stream(self.stream)


@contextlib.contextmanager
def _stdout_to_stderr():
    """RCCL prints a version banner to stdout (file descriptor 1) when rank 0 creates its first communicator; a benchmark's
    stdout carries exactly one JSON line, so the banner is sent to stderr instead"""
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def _librccl_path():
    return os.environ.get('LTU_LIBRCCL') or os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')


class RcclComm:
    """One RCCL communicator per process, driven through the C-ABI.  `control`: the host-side communicator that carries the unique
    id, barriers and scalar reductions (a GlooComm at world > 1; None builds a 1-rank communicator, used to rehearse the captured
    path on one GPU)."""

    def __init__(self, device, control=None):
        self.control = control or LocalComm()
        self.world, self.rank = self.control.world, self.control.rank
        self.device = torch.device(device)
        _lib.call('ltu_comm_load', _librccl_path().encode())
        uid = torch.zeros(128, dtype=torch.uint8)
        if self.rank == 0:
            _lib.call('ltu_comm_unique_id', uid.data_ptr())
        self.control.broadcast(uid, 0)
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device), _stdout_to_stderr():
            _lib.call('ltu_comm_init', ctypes.byref(h), uid.data_ptr(), self.world, self.rank)
            from . import ops                # a stream on a hardware queue other than the compute stream's (ops.concurrent_stream)
            self.stream = ops.concurrent_stream(self.device, [torch.cuda.current_stream(self.device)])
        self.handle = h
        self.calls = 0          # collectives enqueued (eagerly or into a capture): the tests and tools count these

    def allreduce_avg(self, flat, also=None):
        """also: a second stream that contributed to the bucket (the weight-gradient queue's side stream)"""
        if not flat.is_cuda or flat.dtype != torch.float32 or not flat.is_contiguous():
            raise _lib.LtuError('RcclComm.allreduce_avg takes a contiguous fp32 CUDA tensor')
        self.stream.wait_stream(torch.cuda.current_stream())         # fork: everything issued so far produces this bucket
        if also is not None:
            self.stream.wait_stream(also)
        _lib.call('ltu_comm_allreduce_avg', self.handle, flat.data_ptr(), flat.numel(), self.stream.cuda_stream)
        self.calls += 1
        return _StreamHandle(self.stream)

    def broadcast(self, t, src=0):
        if not t.is_cuda or not t.is_contiguous():
            raise _lib.LtuError('RcclComm.broadcast takes a contiguous CUDA tensor')
        self.stream.wait_stream(torch.cuda.current_stream())
        _lib.call('ltu_comm_broadcast', self.handle, t.data_ptr(), t.numel() * t.element_size(), src, self.stream.cuda_stream)
        torch.cuda.current_stream().wait_stream(self.stream)

    def barrier(self):
        self.control.barrier()

    def max_float(self, x):
        return self.control.max_float(x)

    def close(self):
        if self.handle is not None and self.handle.value:
            torch.cuda.synchronize(self.device)
            _lib.call('ltu_comm_destroy', self.handle)
            self.handle = None

"""Per-family device time of one training step, measured live (used by bench.py for the `roofline.families` table).

Every kernel of the step is launched through `_lib.call(name, *args)`.  Between `attach()` and `detach()` the calls that go into
the captured step graph (train.GraphedStep) are logged; the graph's private memory pool keeps every recorded device pointer valid.
`measure()` then re-issues the calls of one family - same order, same arguments, the launching stream substituted - inside a
captured HIP graph of its own and brackets its replays with one HIP event pair on the launching stream: the family's kernels back
to back, nothing else.  (Inside the step's pool the allocator recycles buffers between ops, so a family replayed alone computes on
stale activations; grids and loop bounds come from host arguments, and device-side index data - the ROI sampling plans - is
produced and consumed inside one family, so only the results depend on the stale values, not the work or the addresses.)

Families follow SURVEY.md section 8d's work table (per 128^3 patch, forward; fwd + bwd = 3x): their algorithmic bytes / flops are
the survey's, so the fractions are comparable with the whole-step roofline.
"""
import ctypes

import torch

from lintransunet_amd import _lib

# C-ABI name prefix -> family
FAMILY_OF = [
    ('ltu_layer_tail', 'transformer'), ('ltu_linear', 'transformer'), ('ltu_layernorm', 'transformer'), ('ltu_gelu', 'transformer'),
    ('ltu_reduce_batch', 'transformer'), ('ltu_linattn', 'transformer'), ('ltu_gate', 'transformer'),
    ('ltu_conv3d', 'conv3'), ('ltu_upconv', 'conv3'), ('ltu_sumpool2', 'conv3'),
    ('ltu_instnorm', 'instnorm'),
    ('ltu_roi_resample', 'resample'), ('ltu_trilinear', 'resample'), ('ltu_roi_plan', 'resample'),
    ('ltu_dwconv', 'dwconv'),
]
NO_LAUNCH = ('ltu_roi_plan_size', 'ltu_config_set', 'ltu_version', 'ltu_comm_')
# host pointers passed as integers: (argument index of the array, index of its length, element type) - copied when recorded
HOST_ARRAYS = {'ltu_linear_wgrad_group': (0, 1, _lib.WgradJob), 'ltu_reduce_batch': (0, 1, _lib.ReduceJob)}
# host OUTPUT structs (a deferred-fold descriptor the call fills): replaced by a scratch struct at replay
HOST_OUT = {'ltu_linear_wgrad': 12, 'ltu_layernorm_bwd': 11}

# SURVEY 8d, forward, per 128^3 patch: (GB, GFLOP); 96^3 in brackets there.  `transformer` = Linear + LayerNorm + linear-attention
# core + 1x1x1 convs (the chain kernels fuse them); `conv3` = 3x3x3 convs + the nearest upsampling fused into the un-embedding.
WORK_128 = {'transformer': (2.567 + 0.726 + 0.726 + 0.171, 251.1 + 11.6 + 2.4), 'conv3': (0.686 + 0.202, 394.1),
            'instnorm': (0.484, 0.0), 'resample': (0.122 + 0.068, 0.0), 'dwconv': (0.045, 0.6)}
WORK_96 = {'transformer': (1.922 + 0.541 + 0.541 + 0.072, 186.9 + 8.7 + 1.0), 'conv3': (0.396 + 0.151, 254.4),
           'instnorm': (0.251, 0.0), 'resample': (0.084 + 0.029, 0.0), 'dwconv': (0.034, 0.5)}
TITLES = {'transformer': 'transformer layers: projections, LayerNorm, GELU, linear attention, attention gates (1x1x1 convs)',
          'conv3': '3x3x3 convolutions incl. strided / sub-pixel embedding convs, forward + both gradients',
          'instnorm': 'InstanceNorm + LeakyReLU + residual + dropout, forward + backward',
          'resample': 'ROI warp / un-warp (grid_sample) and trilinear upsampling with adjoints',
          'dwconv': 'positional depthwise 3x3x3 conv',
          'other': 'window embedding, softmax heads, losses, label pyramid, weight preparation, fills'}


def family_of(name):
    for prefix, fam in FAMILY_OF:
        if name.startswith(prefix):
            return fam
    return 'other'


# kernel name (as rocprofv3 prints it) -> family, for the IN-STEP times of the committed kernel statistics (profiles/rNN_bench_kernel_stats.csv):
# what each family's kernels take inside the replayed step, beside the other stream's kernels
KERNEL_FAMILY = [
    (r'tail_fwd|tail_bwd|linear_ring|wgrad_group_ring|wgrad_fat_group|wgroup_fold|wgrad_ring_bf16|linattn|layernorm|gelu_drop|reduce_batch|'
     r'gate_|pw_small', 'transformer'),
    (r'conv3_|conv_class|conv_halo_fold|igemm_|wgrad_tn|wgrad_reduce|upconv|updgrad|sdgrad|sumpool', 'conv3'),
    (r'instnorm|reduce_parts', 'instnorm'),
    (r'plan_gather|plan_scatter|roi_|trilinear|tri_', 'resample'),
    (r'dwconv', 'dwconv'),
]


def in_step_ms(csv_path, steps=8):
    """{family: ms per step} summed from a kernel-statistics file written by tools/profile_bench.sh (exact per step: the last `steps`
    complete replays); kernel durations there include what the concurrency of the two streams costs them"""
    import csv
    import re
    out = {}
    with open(csv_path) as f:
        for r in csv.DictReader(f):
            name = re.sub(r'\(.*', '', r['Name'])
            fam = next((fm for pat, fm in KERNEL_FAMILY if re.search(pat, name)), 'other')
            out[fam] = out.get(fam, 0.0) + int(r['TotalDurationNs']) / steps / 1e6
    return out


class FamilyTimer:
    def __init__(self):
        self.calls = []          # (name, args list, kept-alive objects)
        self._orig = None
        self._scratch_job = _lib.ReduceJob()

    def attach(self):
        """log every C-ABI call issued while the launching stream is CAPTURING (i.e. the calls train.GraphedStep puts into the step
        graph, on whichever thread - autograd runs backward on its own).  The captured graph's private memory pool then owns
        every recorded device pointer for as long as the GraphedStep lives: no other allocation can land there, and
        torch.cuda.graph's empty_cache() cannot unmap it (an eager step's freed blocks it would)."""
        self._orig = _lib.call
        calls = self.calls

        def logging_call(name, *args):
            if not name.startswith(NO_LAUNCH) and torch.cuda.is_current_stream_capturing():
                keep, a = [], list(args)
                if name in HOST_ARRAYS:
                    ip, il, typ = HOST_ARRAYS[name]
                    n = int(a[il])
                    buf = (typ * n)()
                    ctypes.memmove(buf, int(a[ip]), ctypes.sizeof(typ) * n)
                    keep.append(buf)
                    a[ip] = ctypes.addressof(buf)
                if name in HOST_OUT and a[HOST_OUT[name]]:
                    a[HOST_OUT[name]] = ctypes.addressof(self._scratch_job)
                calls.append((name, a, keep))
            return self._orig(name, *args)
        _lib.call = logging_call

    def detach(self):
        if self._orig is not None:
            _lib.call = self._orig
        return len(self.calls)

    def measure(self, reps=5):
        """{family: (ms per step, launches)} - each family's calls replayed back to back from a captured graph"""
        out = {}
        fams = {}
        for name, a, keep in self.calls:
            fams.setdefault(family_of(name), []).append((name, a))
        import os, sys
        only = os.environ.get('LTU_FAMILY_ONLY')
        for fam, lst in fams.items():
            if only and fam != only:
                continue
            print(f'[family_timer] replaying {fam}: {len(lst)} calls', file=sys.stderr, flush=True)
            def issue():
                st = torch.cuda.current_stream().cuda_stream
                for name, a in lst:
                    self._orig(name, *a[:-1], st)
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                issue()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode='thread_local'):
                issue()
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                g.replay()
            e1.record()
            e1.synchronize()
            out[fam] = (e0.elapsed_time(e1) / reps, len(lst))
            del g
        return out

    def group_wgrad(self, reps=5):
        """the largest single kernel of the step, the grouped projection weight gradient (csrc/gemm_ring.hip), alone: (ms per step of its
        recorded launches incl. their folds - at the width the step launches them with -, launches, algorithmic operand bytes =
        sum over jobs of (M K + M N) 2 B)"""
        calls = [(n, a) for n, a, _ in self.calls if n == 'ltu_linear_wgrad_group']
        if not calls:
            return None
        byts = 0
        for _, a in calls:
            jobs = ctypes.cast(int(a[0]), ctypes.POINTER(_lib.WgradJob))
            for i in range(int(a[1])):
                byts += 2 * jobs[i].M * (jobs[i].N + jobs[i].K)
        saved, self.calls = self.calls, [(n, a, None) for n, a in calls]
        try:
            FAMILY_OF.insert(0, ('ltu_linear_wgrad_group', '_group'))
            ms = self.measure(reps)['_group'][0]
        finally:
            FAMILY_OF.pop(0)
            self.calls = saved
        return ms, len(calls), byts

    def table(self, measured, size, batch, hbm_peak_gbs=8000.0, mfma_peak_tflops=2500.0, in_step=None):
        """the `roofline.families` list of bench.py: per family time (replayed ALONE at the capture's launch geometry, and - in_step, from
        the committed kernel statistics of the same bench command - INSIDE the step), algorithmic work of SURVEY 8d (fwd + bwd = 3 x
        forward, x batch), bound, achieved rate and fraction of the peak (of the alone time)"""
        work = {128: WORK_128, 96: WORK_96}.get(size)
        rows = []
        for fam, (ms, n) in sorted(measured.items(), key=lambda kv: -kv[1][0]):
            row = {'family': fam, 'what': TITLES[fam], 'ms_per_step': ms, 'ms_alone': ms,
                   'ms_in_step': (in_step or {}).get(fam), 'launches': n}
            if work and fam in work:
                gb, gf = (3 * batch * v for v in work[fam])
                t_hbm, t_mfma = gb / hbm_peak_gbs * 1e3, gf / (mfma_peak_tflops * 1e3) * 1e3          # ms
                if t_mfma > t_hbm:
                    row.update(bound='mfma', algorithmic_gflop=gf, achieved=gf / ms, peak=mfma_peak_tflops, unit='TFLOP/s',
                               frac=gf / ms / mfma_peak_tflops)
                else:
                    row.update(bound='hbm', algorithmic_gb=gb, achieved=gb / ms * 1e3, peak=hbm_peak_gbs, unit='GB/s',
                               frac=gb / ms * 1e3 / hbm_peak_gbs)
            rows.append(row)
        return rows

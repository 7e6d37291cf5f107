#!/bin/bash
# regenerates the files kept under profiles/ for the current round (run from the repo root on the GPU box; outputs land in
# gpurun_out/ and are copied into profiles/ by hand afterwards):  bash tools/refresh_profiles.sh r02
set -e
R=${1:-r05}
ROOT=$(pwd)
python3 bench.py > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err
bash tools/profile_bench.sh
{
  echo "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-families --steps 10 --warmup 3   (MI355X).  EXACT per-step figures: tools/prof_summary.py --last-steps 8 counts only the kernels of the last 8 complete replays of the step (of the timed region), nothing of the eager / warm-up steps or of the chain-kernel timing graph.  The step is replayed as linear graph segments on the main stream and batches of weight gradients on a side stream: kernel durations include what the concurrency costs them, and their sum exceeds the step time."
  cat gpurun_out/prof_bench_stats.txt
} > gpurun_out/${R}_bench_kernel_stats.txt
cp gpurun_out/prof_bench_stats.csv gpurun_out/${R}_bench_kernel_stats.csv
DB=$(find /tmp/prof_bench -name '*.db' | head -1)
{
  echo "one replayed step of the same trace (tools/trace_step.py): launches, busy time vs span, per-kernel sums"
  python3 tools/trace_step.py "$DB"
  echo; echo "the same step by hardware queue (tools/trace_overlap.py): the side stream's share and what runs concurrently"
  python3 tools/trace_overlap.py "$DB"
} > gpurun_out/${R}_step_trace.txt
bash tools/pmc_step.sh ${R}
bash tools/pmc_valu.sh
cp gpurun_out/pmc_valu.txt gpurun_out/${R}_pmc_valu.txt
{
  echo "== bench_tail.py (transformer layer chain kernels vs the op-by-op launches they replace)"; (cd tools && python3 bench_tail.py 2>/dev/null)
  echo; echo "== bench_nt.py (projections: forward kernel | weight gradient incl. second stage)"; python3 tools/bench_nt.py 2>/dev/null
  echo; echo "== bench_la.py (linear-attention core, forward | backward)"; python3 tools/bench_la.py 2>/dev/null
  echo; echo "== bench_pw.py (LayerNorm, GELU, InstanceNorm)"; python3 tools/bench_pw.py 2>/dev/null
  echo; echo "== bench_conv.py (3x3x3 convs, stride 1)"; python3 tools/bench_conv.py 2>/dev/null
  echo; echo "== the same with the first-generation halo weight gradient (LTU_WHALO_RING=0)"; LTU_WHALO_RING=0 python3 tools/bench_conv.py 2>/dev/null
  echo; echo "== bench_class.py (sub-pixel un-embedding forward) at the three bridge shapes + one brick"; python3 tools/bench_class.py 2 39 23 64 128 32  2 24 14 32 256 64  2 15 9 32 256 128  2 8 8 8 256 256 2>/dev/null
  echo; echo "== the same with the first-generation class kernel (LTU_NO_UPRING=1)"; LTU_NO_UPRING=1 python3 tools/bench_class.py 2 39 23 64 128 32  2 24 14 32 256 64  2 15 9 32 256 128  2 8 8 8 256 256 2>/dev/null
  echo; echo "== bench_sdgrad.py (data gradient of the stride-2 convs)"; python3 tools/bench_sdgrad.py 2>/dev/null
  echo; echo "== the same with the first-generation class kernel (LTU_NO_SDGRAD_RING=1)"; LTU_NO_SDGRAD_RING=1 python3 tools/bench_sdgrad.py 2>/dev/null
  echo; echo "== bench_updgrad.py (data gradient of the un-embedding)"; python3 tools/bench_updgrad.py 2>/dev/null
  echo; echo "== the same through the 64-tap gather implicit GEMM (LTU_NO_UPDGRAD_RING=1)"; LTU_NO_UPDGRAD_RING=1 python3 tools/bench_updgrad.py 2>/dev/null
  echo; echo "== bench_conv.py at the 32x32x128 level: generic convs in the ring style"; python3 tools/bench_conv.py 2 32 32 128 64 0 64  2 32 32 128 32 32 32  2 16 16 64 128 0 160 2>/dev/null
  echo; echo "== the same with the first-generation halo kernel (LTU_NO_CONV_RING=1)"; LTU_NO_CONV_RING=1 python3 tools/bench_conv.py 2 32 32 128 64 0 64  2 32 32 128 32 32 32  2 16 16 64 128 0 160 2>/dev/null
  echo; echo "== bench_upwgrad.py (weight gradient of the un-embedding)"; python3 tools/bench_upwgrad.py 2>/dev/null
  echo; echo "== the same with the first-generation kernel (LTU_UPW_RING=0)"; LTU_UPW_RING=0 python3 tools/bench_upwgrad.py 2>/dev/null
  echo; echo "== bench_dwconv.py (positional depthwise conv)"; python3 tools/bench_dwconv.py 2>/dev/null
  echo; echo "== bucket_timeline.py"; python3 tools/bucket_timeline.py 2>/dev/null
} > gpurun_out/${R}_microbench.txt

#!/bin/bash
# regenerates the files kept under profiles/ for the current round (run from the repo root on the GPU box; outputs land in
# gpurun_out/ and are copied into profiles/ by hand afterwards):  bash tools/refresh_profiles.sh r02
set -e
R=${1:-r03}
ROOT=$(pwd)
python3 bench.py > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err
bash tools/profile_bench.sh
{
  echo "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 3   (MI355X; the trace holds 2 eager steps, 2 capture warm-ups, 13 replays of the step graph, 6 replays of the 64-launch chain-kernel timing graph and 6 replays of each family graph of the roofline table; per-step figures divide by 18, so everything reads high by the replayed families - the one-step trace ${R}_step_trace.txt is exact; summary of the rocpd database by tools/prof_summary.py)"
  cat gpurun_out/prof_bench_stats.txt
} > gpurun_out/${R}_bench_kernel_stats.txt
cp gpurun_out/prof_bench_stats.csv gpurun_out/${R}_bench_kernel_stats.csv
DB=$(find /tmp/prof_bench -name '*.db' | head -1)
{
  echo "one replayed step of the same trace (tools/trace_step.py): launches, busy time vs span, per-kernel sums"
  python3 tools/trace_step.py "$DB"
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
  echo; echo "== bench_class.py (sub-pixel un-embedding forward)"; python3 tools/bench_class.py 2>/dev/null
  echo; echo "== bench_dwconv.py (positional depthwise conv)"; python3 tools/bench_dwconv.py 2>/dev/null
  echo; echo "== bucket_timeline.py"; python3 tools/bucket_timeline.py 2>/dev/null
} > gpurun_out/${R}_microbench.txt

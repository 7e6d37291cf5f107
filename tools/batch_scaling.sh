#!/bin/bash
# per-kernel time per step at B = 1, 2, 4 patches per GPU (rocprofv3 kernel trace of bench.py): which kernels of the step do not
# scale with the batch, i.e. are latency-bound at the benchmarked B = 2.  Run from the repo root on the GPU box.
set -e
ROOT=$(pwd)
export TMPDIR=/tmp
for B in 1 2 4; do
  rm -rf /tmp/prof_b$B
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_b$B -- python3 "$ROOT/bench.py" --batch $B --no-cpu-baseline --no-families --steps 10 --warmup 3 > "$ROOT/gpurun_out/prof_b$B.log" 2>&1
  DB=$(find /tmp/prof_b$B -name '*.db' | head -1)
  python3 "$ROOT/tools/prof_summary.py" "$DB" 8 --last-steps 8 --csv > "$ROOT/gpurun_out/batch_b$B.csv"
  tail -1 "$ROOT/gpurun_out/prof_b$B.log" | cut -c1-160
done
